"""fp32-class GEMM (csrc/gemm_f32.hip) vs the library SGEMM on the four ViT linear shapes at M = 64 x 1374 (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from unopose_amd import ops
torch.set_grad_enabled(False)
M = 64 * 1374
def timeit(f, n=5):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for nm, K, N, gelu in (("qkv", 768, 2304, False), ("proj", 768, 768, False), ("fc1+gelu", 768, 3072, True), ("fc2", 3072, 768, False)):
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
    ws = ops.split_f32(w); as_ = ops.split_f32(a)
    t_split = timeit(lambda: ops.split_f32(a))
    t_gemm = timeit(lambda: ops.linear_f32x3(as_, ws, b, M, N, K, gelu=gelu))
    t_lib = timeit(lambda: F.gelu(F.linear(a, w, b)) if gelu else F.linear(a, w, b))
    fl = 2.0 * M * K * N
    print(f"{nm:9s} K={K:4d} N={N:4d}: split {t_split:7.1f} us, f32x3 GEMM {t_gemm:7.1f} us ({fl / t_gemm / 1e6:6.1f} TF fp32-equivalent, "
          f"{3 * fl / t_gemm / 1e6:6.1f} TF of bf16 MFMA)  library {t_lib:7.1f} us ({fl / t_lib / 1e6:6.1f} TF)", flush=True)
