import os, sys, torch
sys.path.insert(0, os.getcwd())
from unopose_amd.model import UNOPose, default_model_cfg
from unopose_amd.synthetic import trained_like_, make_batch
from unopose_amd import ops
torch.set_grad_enabled(False)
model = trained_like_(UNOPose(default_model_cfg())).cuda().eval()
ep, _, _ = make_batch(3, S=224, seed=53, device="cuda")
ep["coarse_rand"] = torch.rand(3, 18000, generator=torch.Generator().manual_seed(3)).cuda()
cap = []
orig = ops.lrf_global
def wl(pts, u=False):
    o = orig(pts, u); cap.append(o); return o
ops.lrf_global = wl
def run():
    cap.clear()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        model(dict(ep))
    torch.cuda.synchronize()
    return [o.clone() for o in cap]
base = run(); bad = 0
for it in range(60):
    r = run()
    bad += any((a - b).abs().max().item() > 0 for a, b in zip(r, base))
print(os.environ.get("TAG"), "bad forwards:", bad, "/ 60")
