"""fc1 + GELU -> fc2 of a ViT-B block at M = 87 936 with the hidden activation dense (rows 6144 B apart) vs padded by 64 elements."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unopose_amd import ops
M = 64 * 1374
fc1, fc2 = torch.nn.Linear(768, 3072).cuda(), torch.nn.Linear(3072, 768).cuda()
x = torch.randn(M, 768, device="cuda").bfloat16()
def t(pad):
    ops.MLP_PAD_HIDDEN = pad
    with torch.autocast("cuda", dtype=torch.bfloat16):
        for _ in range(3): ops.mlp(x, fc1, fc2)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(5):
            s.record()
            for _ in range(5): ops.mlp(x, fc1, fc2)
            e.record(); torch.cuda.synchronize(); best = min(best, s.elapsed_time(e) / 5)
    return best * 1e3
for r in range(3):
    print(f"mlp (fc1+GELU, fc2) dense hidden {t(False):7.1f} us   padded hidden {t(True):7.1f} us", flush=True)
