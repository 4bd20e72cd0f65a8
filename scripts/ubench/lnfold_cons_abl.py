"""Timing of the LayerNorm-fold consumer GEMM (EPI 6 / 7) for the library in $UNOPOSE_LIB against EPI 0 / 1 (ablation builds: -DGEMM_LNF_ABL)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
from unopose_amd._lib import call, ptr, stream_ptr
torch.set_grad_enabled(False)
dev = torch.device("cuda"); M = 64 * 1374; C = 768
def timeit(f, n=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
xb = torch.randn(M, C, device=dev).bfloat16(); Mp = (M + 255) // 256 * 256
stats = torch.rand(Mp, 3, 2, device=dev) * 256 + 300
out = {}
for name, N, gelu in (("qkv", 2304, 0), ("fc1", 3072, 1)):
    W = (torch.randn(N, C, device=dev) / C ** 0.5).bfloat16(); d = torch.randn(N, device=dev); c = torch.randn(N, device=dev)
    o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    f0 = lambda: call("unopose_linear_bf16", ptr(xb), ptr(W), ptr(d), ptr(o), M, N, C, gelu, stream_ptr())
    f6 = lambda: call("unopose_linear_bf16_lnfold", ptr(xb), ptr(W), ptr(d), ptr(c), ptr(stats), 3, 1e-6, ptr(o), M, N, C, gelu, stream_ptr())
    t0, t6 = [], []
    for r in range(7): t0.append(timeit(f0)); t6.append(timeit(f6))
    out[name] = (sorted(t0)[3], sorted(t6)[3])
print(os.path.basename(os.environ.get("UNOPOSE_LIB", "product")), "  ".join(f"{k}: EPI0/1 {a:.1f} us, EPI6/7 {b:.1f} us (+{b - a:.1f})" for k, (a, b) in out.items()))
