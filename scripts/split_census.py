"""Which fp32 activations are split (ops.split_f32) in one fp32 forward at bench size, who asks, and how often the SAME tensor is split
again.   python scripts/split_census.py"""
import collections, os, sys, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unopose_amd import ops
from unopose_amd.model import UNOPose, default_model_cfg
from unopose_amd.synthetic import make_batch, trained_like_
torch.set_grad_enabled(False)
dev = torch.device("cuda")
model = trained_like_(UNOPose(default_model_cfg(feature_extraction=dict(img_size=518)))).to(dev).eval()
batch, _, _ = make_batch(32, 2048, 5000, 518, seed=100, device=dev)
batch["coarse_rand"] = torch.rand(32, 18000, device=dev)
model(dict(batch)); torch.cuda.synchronize()
orig = ops.split_f32
seen, sites = collections.Counter(), collections.Counter()
def spy(x2):
    key = (x2.data_ptr(), tuple(x2.shape), x2._version)
    seen[key] += 1
    fr = [f for f in traceback.extract_stack()[:-1] if "unopose_amd" in f.filename][-3:]
    sites[(tuple(x2.shape), " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(fr)))] += 1
    return orig(x2)
ops.split_f32 = spy
model(dict(batch)); torch.cuda.synchronize()
print("split launches per forward:", sum(seen.values()), " distinct (pointer, shape, version):", len(seen), " repeated:", sum(v - 1 for v in seen.values()))
for (shape, site), n in sorted(sites.items(), key=lambda kv: -kv[1] * kv[0][0][0] * kv[0][0][1]):
    print(f"{n:4d} x {str(shape):16s} {shape[0] * shape[1] * 4 / 1e6:8.1f} MB  {site}")
