"""How much of a ViT linear's launch is the last, partly filled round of 256 x 256 tiles?  Times csrc/gemm.hip on the four shapes at row counts
around whole rounds of the 256 CUs (tiles = ceil(M / 256) * N / 256)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unopose_amd import ops
torch.set_grad_enabled(False)
g = torch.Generator(device="cuda").manual_seed(0)

def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): f()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n * 1e3)
    return best

for name, K, N, gelu in (("qkv", 768, 2304, False), ("proj", 768, 768, False), ("fc1+gelu", 768, 3072, True), ("fc2", 3072, 768, False)):
    tn = N // 256
    for M in (87680, (1024 // tn) * 256 if tn == 3 else (3072 // tn) * 256 if tn == 9 else (4096 // tn) * 256, 87040, 88064):
        a = torch.randn(M, K, device="cuda", generator=g).bfloat16()
        w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
        b = torch.randn(N, device="cuda", generator=g)
        us = timeit(lambda: ops.linear_bf16_hip(a, w, b, gelu))
        tiles = -(-M // 256) * tn
        print(f"{name:9s} M={M:6d} tiles={tiles:5d} = {tiles / 256:6.3f} rounds: {us:7.1f} us  {2 * M * K * N / us / 1e6:7.1f} TFLOP/s  {us / (tiles / 256):6.1f} us per round", flush=True)
