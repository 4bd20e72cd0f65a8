"""ops.vit_attention at the bench shape under the kernel-selection switches of csrc/vit_attn.hip (read once per process):
   for m in "" UNOPOSE_VIT_SPLIT=0 UNOPOSE_VIT_DMA=0; do env $m python scripts/ubench/vit_attn_modes.py; done"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unopose_amd import ops
torch.set_grad_enabled(False)
qkv = torch.randn(64, 1374, 2304, device="cuda").bfloat16()
def t(f, n=20):
    f(); torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
ts = sorted(t(lambda: ops.vit_attention(qkv, 12)) for _ in range(5))
print({k: v for k, v in os.environ.items() if k.startswith("UNOPOSE_VIT")}, "min %.1f us  median %.1f us" % (ts[0], ts[2]))
