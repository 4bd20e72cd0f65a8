"""torch.profiler view of one training step at BASELINE configs[3]'s shape: aten ops by device time with input shapes (where the
torch glue of the step goes: copies, adds, library GEMMs).  Run on the GPU box."""
import os, sys
import torch
from torch.profiler import ProfilerActivity, profile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

sys.argv = ["bench.py", "--train", "--steps", "1", "--warmup", "2"]
args = bench.parse()
import io
from contextlib import redirect_stdout
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    with redirect_stdout(io.StringIO()):
        bench.main()
rows = prof.key_averages(group_by_input_shape=True)
rows = sorted(rows, key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in rows)
print(f"total device time {tot / 1e3:.1f} ms over 3 steps")
for e in rows[:int(os.environ.get("TOP", "60"))]:
    if os.environ.get("ATEN_ONLY") and not (e.key.startswith("aten::") or e.key.startswith("_") or "Backward" in e.key):
        continue
    print(f"{e.self_device_time_total / 3e3:8.3f} ms/step  n/step {e.count / 3:6.1f}  {e.key[:50]:50s} {str(e.input_shapes)[:150]}")
