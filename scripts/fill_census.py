import os, sys, traceback, collections
import torch
sys.path.insert(0, os.getcwd())
from unopose_amd.model import UNOPose, default_model_cfg
from unopose_amd.synthetic import make_batch, trained_like_
from torch.utils._python_dispatch import TorchDispatchMode
torch.set_grad_enabled(False)
dev = torch.device("cuda")
model = trained_like_(UNOPose(default_model_cfg(feature_extraction=dict(img_size=518)))).to(dev).eval()
batch, _, _ = make_batch(32, 2048, 5000, 518, seed=100, device=dev)
batch["coarse_rand"] = torch.rand(32, 18000, device=dev)
with torch.autocast("cuda", dtype=torch.bfloat16): model(dict(batch))
sites = collections.Counter()
class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if any(k in name for k in ("fill", "zero", "full", "ones", "arange")):
            t = out if torch.is_tensor(out) else (args[0] if args and torch.is_tensor(args[0]) else None)
            if t is not None and t.is_cuda and t.numel() * t.element_size() > (1 << 20):
                fr = [f for f in traceback.extract_stack() if "unopose_amd" in f.filename][-2:]
                sites[(name, str(t.dtype), tuple(t.shape), " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(fr)))] += 1
        return out
with Spy(), torch.autocast("cuda", dtype=torch.bfloat16):
    model(dict(batch))
for k, v in sites.items(): print(v, k)
