import os, sys, torch
sys.path.insert(0, os.getcwd())
from unopose_amd.model import UNOPose, default_model_cfg
from unopose_amd.synthetic import trained_like_, make_batch
from unopose_amd import ops
torch.set_grad_enabled(False)
model = trained_like_(UNOPose(default_model_cfg())).cuda().eval()
ep, _, _ = make_batch(3, S=224, seed=53, device="cuda")
ep["coarse_rand"] = torch.rand(3, 18000, generator=torch.Generator().manual_seed(3)).cuda()
cap = {}
orig = model._forward_from
def wrapped(pre, *a, **k):
    cap.clear(); cap.update({kk: v for kk, v in pre.items() if torch.is_tensor(v)})
    return orig(pre, *a, **k)
model._forward_from = wrapped
og = model._geo
def wgeo(*a, **k):
    g = og(*a, **k); cap["geo_main"] = g; return g
model._geo = wgeo
def run():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        o = model(dict(ep))
    torch.cuda.synchronize()
    return {k: v.float().clone() for k, v in cap.items()}
base = run(); bad = {}
for it in range(40):
    r = run()
    for k in base:
        d = (r[k] - base[k]).abs().max().item()
        if d > 0: bad[k] = max(bad.get(k, 0), d)
print({k: f"{v:.2e}" for k, v in bad.items()}, "keys:", list(base))
# lrf_global alone, repeatedly, with and without a concurrent stream
pts = ep["pts"]
b0 = ops.lrf_global(pts, False).clone()
w = 0
for _ in range(50): w = max(w, (ops.lrf_global(pts, False) - b0).abs().max().item())
print("lrf_global alone:", w)
