"""Isolated time of the linear attention's key / value state: one launch (unopose_linear_attention_kv_state) vs the 7-launch form, B = 64, J = 196."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unopose_amd import ops
from unopose_amd.model.unopose import UNOPose
from unopose_amd.model.config import default_model_cfg

torch.manual_seed(0)
m = UNOPose(default_model_cfg()).cuda().eval()
att = m.fine_point_matching.transformers[0].dense_layer.attention.attention
xq = torch.randn(64, 2048, 256, device="cuda").bfloat16()
xkv = torch.randn(64, 196, 256, device="cuda").bfloat16()
for flag in (False, True, False, True):
    ops.USE_LA_KV_STATE = flag
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        for _ in range(5):
            ops.focused_linear_attention(xq, xkv, att, 4, 3)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            ops.focused_linear_attention(xq, xkv, att, 4, 3)
        e1.record()
        torch.cuda.synchronize()
    print(f"USE_LA_KV_STATE={flag}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per focused_linear_attention call")
