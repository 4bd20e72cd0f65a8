import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from unopose_amd.model import UNOPose, default_model_cfg
from unopose_amd.synthetic import trained_like_, make_batch
from unopose_amd.pipeline import PipelinedForward
torch.set_grad_enabled(False)
KEYS = ("init_R", "init_t", "init_pose_score", "pred_R", "pred_t", "pred_pose_score")
model = trained_like_(UNOPose(default_model_cfg())).cuda().eval()
eps = []
for i in range(6):
    ep, _, _ = make_batch(3, S=224, seed=50 + i, device="cuda")
    ep["coarse_rand"] = torch.rand(3, 18000, generator=torch.Generator().manual_seed(i)).cuda()
    eps.append(ep)
def run(depth):
    p = PipelinedForward(model, depth=depth)
    ts = [p.submit(dict(ep)) for ep in eps]
    outs = [{k: t.result()[k].clone() for k in KEYS} for t in ts]
    p.drain(); torch.cuda.synchronize()
    return outs
a = run(1); b = run(1); c = run(2); d = run(2)
def cmp(x, y, name):
    for i in range(6):
        ds = {k: (x[i][k] - y[i][k]).abs().max().item() for k in KEYS}
        if any(v > 0 for v in ds.values()):
            print(name, "batch", i, {k: f"{v:.2e}" for k, v in ds.items() if v > 0})
cmp(a, b, "seq vs seq"); cmp(a, c, "seq vs pipe"); cmp(c, d, "pipe vs pipe"); print("done")
