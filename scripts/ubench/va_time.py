"""Time ops.vit_attention at the bench's shape with the library UNOPOSE_LIB points at (same-box A/B with scripts/build_variant.py):
64 crops x 1370 tokens x 12 heads; prints us per launch and a checksum (variants must agree bit for bit)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unopose_amd import ops
torch.manual_seed(0)
for B, T in ((64, 1370), (64, 1374), (8, 1025)):
    qkv = torch.randn(B, T, 2304, device="cuda").bfloat16()
    for _ in range(3):
        out = ops.vit_attention(qkv, 12)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            out = ops.vit_attention(qkv, 12)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 100)
    print(f"{os.path.basename(os.environ.get('UNOPOSE_LIB', 'product')):32s} B={B} T={T}: {best:7.1f} us  checksum {out.float().abs().sum().item():.6e}", flush=True)
