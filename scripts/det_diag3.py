import os, sys, torch
sys.path.insert(0, os.getcwd())
from unopose_amd.model import UNOPose, default_model_cfg
from unopose_amd.synthetic import trained_like_, make_batch
from unopose_amd import ops
import unopose_amd.model.unopose as U
torch.set_grad_enabled(False)
model = trained_like_(UNOPose(default_model_cfg())).cuda().eval()
ep, _, _ = make_batch(3, S=224, seed=53, device="cuda")
ep["coarse_rand"] = torch.rand(3, 18000, generator=torch.Generator().manual_seed(3)).cuda()
def run():
    model.taps = {}; model.coarse_point_matching.taps = {}
    with torch.autocast("cuda", dtype=torch.bfloat16):
        o = model(dict(ep))
    torch.cuda.synchronize()
    r = {k: v.float().clone() for k, v in model.taps.items() if torch.is_tensor(v)}
    r.update({"coarse_" + k: v.float().clone() for k, v in model.coarse_point_matching.taps.items() if torch.is_tensor(v)})
    model.taps = None; model.coarse_point_matching.taps = None
    return r
for label, setup in (("default", lambda: None), ("GEOM_UNDER_VIT=0", lambda: setattr(U, "GEOM_UNDER_VIT", 0)),
                     ("sparse upproj off", lambda: setattr(ops, "USE_SPARSE_UPPROJ", False))):
    setup()
    base = run(); bad = {}
    for it in range(30):
        r = run()
        for k in base:
            d = (r[k] - base[k]).abs().max().item()
            if d > 0: bad[k] = max(bad.get(k, 0), d)
    print(label, {k: f"{v:.2e}" for k, v in bad.items()})
