import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unopose_amd import ops
from unopose_amd.model import UNOPose, default_model_cfg
torch.set_grad_enabled(False)
m = UNOPose(default_model_cfg()).cuda().eval()
att = m.coarse_point_matching.transformers[0].layers[0].attention.attention
B, n = 32, 197
E = torch.randn(B, n, n, 256, device="cuda").bfloat16()
x = torch.randn(B, n, 256, device="cuda")
def run():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        return ops.token_attention(x, x, att, 4, E)
for _ in range(3): run()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(10): run()
    torch.cuda.synchronize()
for e in prof.key_averages():
    if "token_attn" in e.key:
        us = e.device_time_total / e.count
        print(e.key[:40], f"{us:.1f} us  {E.numel()*2/us/1e6:.2f} TB/s")
