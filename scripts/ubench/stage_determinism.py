import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from unopose_amd.model import UNOPose, default_model_cfg
from unopose_amd.pipeline import PipelinedForward
from unopose_amd.synthetic import trained_like_, make_batch
torch.set_grad_enabled(False)
img = int(sys.argv[1]) if len(sys.argv) > 1 else 518
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
model = trained_like_(UNOPose(default_model_cfg(feature_extraction=dict(img_size=img)))).cuda().eval()
eps = []
for i in range(3):
    ep, _, _ = make_batch(B, S=img, seed=50 + i, device="cuda"); ep["coarse_rand"] = torch.rand(B, 18000, generator=torch.Generator().manual_seed(i)).cuda(); eps.append(ep)
K = ("init_R", "init_t", "pred_R", "pred_t", "pred_pose_score")
def run_seq():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        return [{k: model(dict(e))[k].clone() for k in K} for e in eps]
a = run_seq(); b = run_seq()
for i in range(3):
    print("seq vs seq batch", i, {k: (a[i][k] - b[i][k]).abs().max().item() for k in K})
pf = PipelinedForward(model, depth=2, stages="auto")
outs = [pf.submit(dict(e)) for e in eps]
for i, t in enumerate(outs):
    o = t.result()
    print("stage vs seq batch", i, {k: (o[k] - a[i][k]).abs().max().item() for k in K})
pf.close()
c = run_seq()
for i in range(3):
    print("seq(after) vs seq batch", i, {k: (a[i][k] - c[i][k]).abs().max().item() for k in K})
