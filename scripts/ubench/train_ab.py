"""Same-box A/B of the training step (bench.py --train shape) for one `unopose_amd.ops` attribute: one process per measurement, alternating.
usage: python scripts/ubench/train_ab.py TRAIN_OWN_GEMM_MIN_FLOP 2e10 1e10 [--reps 3]"""
import io, json, os, subprocess, sys
from contextlib import redirect_stdout
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
if sys.argv[1] == "--one":
    import bench
    from unopose_amd import ops
    name, val = sys.argv[2], sys.argv[3]
    assert hasattr(ops, name), name
    setattr(ops, name, {"True": True, "False": False}.get(val, None) if val in ("True", "False") else float(val))
    sys.argv = ["bench.py", "--train", "--steps", "5", "--warmup", "2"]
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.main()
    line = json.loads(buf.getvalue().strip().splitlines()[-1])
    print(f"{name}={val}: {line['ms_per_step']:.2f} ms/step  loss {line['loss_first_last'][1]:.4f}", flush=True)
else:
    name, vals, reps = sys.argv[1], [a for a in sys.argv[2:] if not a.startswith("--")], 3
    for rep in range(reps):
        for v in vals:
            subprocess.run([sys.executable, os.path.abspath(__file__), "--one", name, v], stderr=subprocess.DEVNULL)
