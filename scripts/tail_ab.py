import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unopose_amd import ops
from unopose_amd.model import UNOPose, default_model_cfg
torch.set_grad_enabled(False)
m = UNOPose(default_model_cfg()).cuda().eval()
layer = m.coarse_point_matching.transformers[0].layers[1]
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/it*1e3
for rows in (197*32, 2048*32):
    h = torch.randn(1, rows, 256, device="cuda").bfloat16(); x = torch.randn(1, rows, 256, device="cuda").bfloat16()
    def comp():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            a = layer.attention
            r = ops.add_layernorm(a.linear(h), x, a.norm)
            return layer.output(r)
    print(rows, "tail us", t(lambda: ops.transformer_tail(h, x, layer)), "composite us", t(comp))
