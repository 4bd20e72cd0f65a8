"""Time of the geometric embedding's forward + backward under autograd at the training step's shape (8 clouds x 197 tokens, twice per step):
own kernels (ops.TRAIN_OWN_GEO) vs the op-by-op composite."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unopose_amd import ops
from unopose_amd.model import UNOPose, default_model_cfg
m = UNOPose(default_model_cfg()).geo_embedding.cuda()
pts = torch.randn(8, 197, 3).cuda(); pts = pts / pts.norm(dim=2).max()
dE = torch.randn(8, 197, 197, 256).cuda()
def step():
    for p in m.parameters(): p.grad = None
    with ops.differentiable():
        out = ops.geo_embedding(pts, m)
    out.backward(dE)
for own in (True, False, True, False):
    ops.TRAIN_OWN_GEO = own
    for _ in range(3): step()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): step()
    e.record(); torch.cuda.synchronize()
    print(f"TRAIN_OWN_GEO={own}: {s.elapsed_time(e) / 10:.3f} ms per forward + backward (8 x 197 x 197 pairs)")
