import os, sys, traceback, torch
sys.path.insert(0, os.getcwd())
import torch.nn.functional as F
from unopose_amd.model import UNOPose, default_model_cfg
from unopose_amd.synthetic import trained_like_, make_batch
torch.set_grad_enabled(False)
model = trained_like_(UNOPose(default_model_cfg(feature_extraction=dict(img_size=518)))).cuda().eval()
ep, _, _ = make_batch(32, 2048, 5000, 518, device="cuda")
ep["coarse_rand"] = torch.rand(32, 18000, device="cuda")
with torch.autocast("cuda", dtype=torch.bfloat16):
    model(dict(ep))  # warm caches
seen = {}
def wrap(mod, name):
    orig = getattr(mod, name)
    def f(*a, **k):
        ts = [t for t in a if torch.is_tensor(t)]
        if ts and ts[0].is_cuda and (any(t.dtype == torch.bfloat16 for t in ts) or torch.is_autocast_enabled()):
            st = [l for l in traceback.format_stack(limit=8) if "unopose_amd" in l]
            key = (name, tuple(tuple(t.shape) for t in ts[:3]), tuple(str(t.dtype) for t in ts[:3]), st[-1].strip().splitlines()[0] if st else "?")
            seen[key] = seen.get(key, 0) + 1
        return orig(*a, **k)
    setattr(mod, name, f)
for mod, name in ((torch, "matmul"), (torch, "bmm"), (torch, "einsum"), (F, "linear"), (torch, "_addmm_activation"), (torch, "addmm"), (torch, "baddbmm"), (torch, "mm")):
    wrap(mod, name)
orig_call = torch.nn.Linear.forward
def lf(self, x):
    if x.is_cuda:
        st = [l for l in traceback.format_stack(limit=10) if "unopose_amd" in l]
        key = ("nn.Linear.forward", tuple(x.shape), str(x.dtype), st[-1].strip().splitlines()[0] if st else "?")
        seen[key] = seen.get(key, 0) + 1
    return orig_call(self, x)
torch.nn.Linear.forward = lf
with torch.autocast("cuda", dtype=torch.bfloat16):
    model(dict(ep))
for k, v in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(v, k)
