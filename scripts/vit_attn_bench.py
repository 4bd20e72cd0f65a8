import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unopose_amd import ops
torch.set_grad_enabled(False)
for T in ([int(a) for a in sys.argv[1:]] or [261, 1374]):
    qkv = torch.randn(64, T, 2304, device="cuda").bfloat16()
    for _ in range(3): ops.vit_attention(qkv, 12)
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): ops.vit_attention(qkv, 12)
    e.record(); torch.cuda.synchronize(); us = s.elapsed_time(e)/20*1e3
    fl = 64*12*4*T*T*64
    print(f"T={T}: {us:.0f} us, {fl/us/1e6:.0f} TFLOP/s")
