"""bf16 GEMM rate of the four ViT-B linears at the bench token count, per BLAS backend."""
import sys, torch
import torch.nn.functional as F
torch.set_grad_enabled(False)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 64 * 1374
for lib in ("cublaslt", "cublas"):
    torch.backends.cuda.preferred_blas_library(lib)
    for name, K, N in (("qkv", 768, 2304), ("proj", 768, 768), ("fc1", 768, 3072), ("fc2", 3072, 768)):
        x = torch.randn(M, K, device="cuda").bfloat16(); w = torch.randn(N, K, device="cuda").bfloat16(); b = torch.randn(N, device="cuda").bfloat16()
        for _ in range(3): F.linear(x, w, b)
        torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): F.linear(x, w, b)
        e.record(); torch.cuda.synchronize(); us = s.elapsed_time(e) / 10 * 1e3
        print(f"{lib:9s} {name:5s} M={M} K={K} N={N}: {us:7.1f} us  {2*M*K*N/us/1e6:7.1f} TFLOP/s")
