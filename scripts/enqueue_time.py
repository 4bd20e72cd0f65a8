"""How long does the HOST need to enqueue one forward (launch-bound check)?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unopose_amd.model import UNOPose, default_model_cfg
from unopose_amd.synthetic import make_batch, trained_like_
torch.set_grad_enabled(False)
dev = torch.device("cuda")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
S0 = int(sys.argv[2]) if len(sys.argv) > 2 else 224
model = trained_like_(UNOPose(default_model_cfg(feature_extraction=dict(img_size=S0)))).to(dev).eval()
S = int(sys.argv[2]) if len(sys.argv) > 2 else 224
ep, _, _ = make_batch(B, S=S, device=dev)
ep["coarse_rand"] = torch.rand(B, 18000, device=dev)
def step():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        return model(dict(ep))
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"B={B}: host enqueue {1e3*(t1-t0)/5:.2f} ms/step, wall {1e3*(t2-t0)/5:.2f} ms/step")
