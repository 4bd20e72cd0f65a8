"""One bench run with a module attribute / environment knob set: python scripts/ubench/knob_ab.py <img> name=value ...
names: model.GEOM_UNDER_VIT=2, env.GPU_MAX_HW_QUEUES=24, ops.SOMETHING=False, pipeline.WHOLE_INTERNAL_OVERLAP=False"""
import io, json, os, sys
from contextlib import redirect_stdout
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
img, knobs = sys.argv[1], sys.argv[2:]
for k in knobs:
    n, v = k.split("=")
    if n.startswith("env."):
        os.environ[n[4:]] = v
import bench  # noqa: E402
import unopose_amd.model.unopose as mu  # noqa: E402
import unopose_amd.pipeline as pl  # noqa: E402
from unopose_amd import ops  # noqa: E402
for k in knobs:
    n, v = k.split("=")
    val = {"True": True, "False": False}.get(v, int(v) if v.lstrip("-").isdigit() else v)
    if n.startswith("model."):
        setattr(mu, n[6:], val)
    elif n.startswith("pipeline."):
        setattr(pl, n[9:], val)
    elif n.startswith("ops."):
        setattr(ops, n[4:], val)
sys.argv = ["bench.py", "--img", img, "--no-cpu-baseline", "--no-roofline", "--no-fp32", "--no-extra", "--steps", "40", "--warmup", "6"]
buf = io.StringIO()
with redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print(f"{img} {' '.join(knobs) or '(defaults)'}: {d['value']:.1f} pairs/s  {d['ms_per_step']:.3f} ms/step", flush=True)
