"""Cost of the geometric embedding under autograd (training path: op-by-op torch composite), forward + backward, 16 clouds x 197 points."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unopose_amd import ops
from unopose_amd.model import UNOPose, default_model_cfg
torch.manual_seed(0)
m = UNOPose(default_model_cfg()).cuda().train().geo_embedding
pts = torch.cat([torch.ones(16, 1, 3), torch.rand(16, 196, 3) * 1.2 - 0.6], 1).cuda()
g = torch.randn(16, 197, 197, 256, device="cuda")
def step():
    for p in m.parameters(): p.grad = None
    with ops.differentiable():
        e = ops.geo_embedding(pts, m)
    e.backward(g)
for _ in range(3): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): step()
e1.record(); torch.cuda.synchronize()
print(f"geo embedding fwd + bwd under autograd: {e0.elapsed_time(e1) / 10:.2f} ms per call (the training step makes two calls of 8 clouds... or one of 16)")
