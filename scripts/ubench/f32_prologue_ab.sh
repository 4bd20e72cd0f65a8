#!/bin/bash
# fp32 step (the reference's default precision) with the ViT's fused fp32 front end (modules.F32_PROLOGUE) on / off: one process per run
R=${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2 3; do
  for v in False True; do
    python3 - <<P
import io, json, sys
from contextlib import redirect_stdout
sys.path.insert(0, "$R")
import bench
import unopose_amd.model.modules as mm
mm.F32_PROLOGUE = $v
sys.argv = ["bench.py", "--dtype", "fp32", "--no-cpu-baseline", "--no-extra", "--no-roofline", "--steps", "12", "--warmup", "3"]
buf = io.StringIO()
with redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print("F32_PROLOGUE=$v", round(d["value"], 1), "pairs/s", round(d["ms_per_step"], 3), "ms/step  rot err", d["sanity"]["median_rot_err_vs_gt"])
P
  done
done
