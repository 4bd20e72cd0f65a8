"""Same-box A/B of whole bench steps: runs bench.py's main() in this process twice per setting of one `unopose_amd.ops` switch.
usage: python scripts/ubench/bench_ab.py GEO_TABLE [--img 518 ...]   (prints value / ms_per_step per run, alternating False / True)"""
import io, json, os, sys
from contextlib import redirect_stdout

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import bench
from unopose_amd import ops

name = sys.argv[1]
rest = sys.argv[2:]
for rep in range(2):
    for val in (False, True):
        setattr(ops, name, val)
        sys.argv = ["bench.py", "--no-cpu-baseline", "--no-fp32", "--no-extra", "--no-roofline", "--steps", "30"] + rest
        buf = io.StringIO()
        with redirect_stdout(buf):
            bench.main()
        line = json.loads(buf.getvalue().strip().splitlines()[-1])
        print(f"{name}={val}: {line['value']:.1f} {line['unit']}  {line['ms_per_step']:.3f} ms/step", flush=True)
