"""Same-box A/B of whole bench steps for one boolean `unopose_amd.ops` switch: ONE PROCESS per measurement (an in-process loop drifts: the
same setting went 31.2 -> 35.4 ms over four repetitions of bench.main() in one interpreter, round 6), alternating False / True, the
driver's own step counts.
usage: python scripts/ubench/bench_ab.py GEO_TABLE [--reps 4] [--img 518 ...]   (prints value / ms_per_step per run)"""
import io, json, os, subprocess, sys
from contextlib import redirect_stdout

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)

if sys.argv[1] == "--one":
    import bench
    from unopose_amd import ops

    name, val = sys.argv[2], sys.argv[3] == "True"
    if name.startswith("model."):  # a module-level flag of unopose_amd.model.unopose (e.g. model.COARSE_SLOT)
        import unopose_amd.model.unopose as mu
        assert hasattr(mu, name[6:]), name
        setattr(mu, name[6:], val)
    else:
        assert hasattr(ops, name), name
        setattr(ops, name, val)
    sys.argv = ["bench.py", "--no-cpu-baseline", "--no-fp32", "--no-extra", "--no-roofline", "--steps", "20", "--warmup", "5"] + sys.argv[4:]
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.main()
    line = json.loads(buf.getvalue().strip().splitlines()[-1])
    print(f"{name}={val}: {line['value']:.1f} {line['unit']}  {line['ms_per_step']:.3f} ms/step  (HIP-event median {line['step_ms_hip_events']['median']:.3f})", flush=True)
else:
    name, rest, reps = sys.argv[1], sys.argv[2:], 4
    if "--reps" in rest:
        i = rest.index("--reps")
        reps = int(rest[i + 1])
        del rest[i:i + 2]
    for rep in range(reps):
        for val in (False, True):
            subprocess.run([sys.executable, os.path.abspath(__file__), "--one", name, str(val)] + rest, stderr=subprocess.DEVNULL)
