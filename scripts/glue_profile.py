"""Where the remaining torch glue (casts, cats, copies, clamps, fills) of one forward comes from:
torch.profiler with Python stacks, device time grouped by (op, innermost unopose_amd frame)."""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unopose_amd.model import UNOPose, default_model_cfg  # noqa: E402
from unopose_amd.synthetic import make_batch, trained_like_  # noqa: E402

img = int(sys.argv[1]) if len(sys.argv) > 1 else 224
torch.set_grad_enabled(False)
dev = torch.device("cuda")
model = trained_like_(UNOPose(default_model_cfg(feature_extraction=dict(img_size=img)))).to(dev).eval()
batch, _, _ = make_batch(32, 2048, 5000, img, seed=1, device=dev)
batch["coarse_rand"] = torch.rand(32, 18000, device=dev)
for _ in range(2):
    with torch.autocast("cuda", dtype=torch.bfloat16):
        model(dict(batch))
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    with torch.autocast("cuda", dtype=torch.bfloat16):
        model(dict(batch))
    torch.cuda.synchronize()
GLUE = ("aten::_to_copy", "aten::cat", "aten::copy_", "aten::clamp", "aten::clamp_min", "aten::relu", "aten::fill_",
        "aten::zeros", "aten::gelu", "aten::add", "aten::mul", "aten::div", "aten::sub", "aten::index", "aten::gather",
        "aten::sigmoid", "aten::stack", "aten::where", "aten::norm", "aten::linalg_vector_norm", "aten::mean", "aten::sum",
        "aten::max", "aten::topk", "aten::argmax", "aten::index_select", "aten::eq", "aten::gt", "aten::lt")
agg = collections.defaultdict(lambda: [0.0, 0])
for ev in prof.key_averages(group_by_stack_n=12):
    if ev.key in GLUE and ev.self_device_time_total > 0:
        frame = next((f for f in ev.stack if "unopose_amd" in f), ev.stack[0] if ev.stack else "?")
        frame = frame.split("unopose_amd/")[-1]
        a = agg[(ev.key, frame)]
        a[0] += ev.self_device_time_total
        a[1] += ev.count
rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
tot = sum(v[0] for _, v in rows)
print(f"glue device time {tot/1e3:.2f} ms")
for (name, frame), (t, n) in rows[:60]:
    print(f"{t/1e3:7.3f} ms {n:4d}x {name:22s} {frame}")
