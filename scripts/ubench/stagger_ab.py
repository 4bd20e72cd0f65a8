"""Whole-step A/B of the residual GEMM's start offset (unopose_gemm_fold_stagger; 1/8 ticks of 100 MHz per K-tile): one process per measurement.
usage: python scripts/ubench/stagger_ab.py [values ...]
Round 6, whole step at 518 x 518 (three runs each): 0 -> 1075.4, 420 -> 1078.0, 840 (default) -> 1075.6, 1680 -> 1070.1 pairs/s: flat -- the offset that buys
5-9 % of the isolated proj launch is invisible inside the pipelined step."""
import io, json, os, subprocess, sys
from contextlib import redirect_stdout

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
if sys.argv[1] == "--one":
    import bench
    from unopose_amd._lib import lib

    val = sys.argv[2]
    lib().unopose_gemm_fold_stagger(int(val))  # (returns the previous value, not a status: not through `call`)
    sys.argv = ["bench.py", "--no-cpu-baseline", "--no-fp32", "--no-extra", "--no-roofline", "--steps", "40", "--warmup", "6"]
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.main()
    d = json.loads(buf.getvalue().strip().splitlines()[-1])
    print(f"stagger {val}: {d['value']:.1f} pairs/s  {d['ms_per_step']:.3f} ms/step", flush=True)
else:
    vals = sys.argv[1:] or ["0", "840"]
    for rep in range(3):
        for v in vals:
            subprocess.run([sys.executable, os.path.abspath(__file__), "--one", v], stderr=subprocess.DEVNULL)
