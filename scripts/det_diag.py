import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from unopose_amd.model import UNOPose, default_model_cfg
from unopose_amd.synthetic import trained_like_, make_batch
from unopose_amd import ops
torch.set_grad_enabled(False)
model = trained_like_(UNOPose(default_model_cfg())).cuda().eval()
ep, _, _ = make_batch(3, S=224, seed=53, device="cuda")
ep["coarse_rand"] = torch.rand(3, 18000, generator=torch.Generator().manual_seed(3)).cuda()
def run(taps):
    fm = model.fine_point_matching
    fm.taps = {} if taps else None
    cm = model.coarse_point_matching
    cm.taps = {} if taps else None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        o = model(dict(ep))
    r = {k: o[k].float().clone() for k in ("init_R", "pred_R", "pred_t")}
    if taps:
        r.update({"fine_" + k: v.float().clone() for k, v in fm.taps.items() if torch.is_tensor(v)})
        r.update({"coarse_" + k: v.float().clone() for k, v in cm.taps.items() if torch.is_tensor(v)})
    fm.taps = None; cm.taps = None
    return r
for taps in (True, False):
    base = run(taps)
    bad = {}
    for it in range(30):
        r = run(taps)
        for k in base:
            d = (r[k] - base[k]).abs().max().item()
            if d > 0: bad[k] = max(bad.get(k, 0), d); 
    print("taps" if taps else "fused", {k: f"{v:.2e}" for k, v in bad.items()})
