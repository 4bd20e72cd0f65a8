"""csrc/gemm.hip vs the library on the ViT linear shapes: correctness against an fp32 reference of the same op and
interleaved timing (HIP events).  python scripts/gemm_ab.py [rows]"""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unopose_amd import ops

torch.set_grad_enabled(False)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 64 * 1374
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)

def timeit(f, n=20):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

# correctness on ragged M, every epilogue
for (m, K, N, gelu) in ((1000, 128, 256, False), (4096 + 77, 768, 768, True), (8192, 3072, 768, False), (5000, 768, 2304, True)):
    a = torch.randn(m, K, device=dev, generator=g).bfloat16()
    w = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).bfloat16()
    b = torch.randn(N, device=dev, generator=g)
    ref = a.float() @ w.float().t() + b
    if gelu: ref = F.gelu(ref)
    out = ops.linear_bf16_hip(a, w, b, gelu)
    err = (out.float() - ref).abs()
    tol = (ref.abs() * 2 ** -8 + 2e-3)
    print(f"check M={m} K={K} N={N} gelu={gelu}: max err {err.max().item():.3e}, violations {(err > tol).sum().item()}")
    assert (err <= tol).all()

for name, K, N, gelu in (("qkv", 768, 2304, False), ("proj", 768, 768, False), ("fc1+gelu", 768, 3072, True), ("fc2", 3072, 768, False),
                         ("upproj", 3072, 4096, False)):
    a = torch.randn(M, K, device=dev, generator=g).bfloat16()
    w = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).bfloat16()
    b = torch.randn(N, device=dev, generator=g)
    bb = b.bfloat16()
    lib = (lambda: F.gelu(F.linear(a, w, bb))) if gelu else (lambda: F.linear(a, w, bb))
    mine = lambda: ops.linear_bf16_hip(a, w, b, gelu)
    tl, tm = [], []
    for _ in range(3):
        tl.append(timeit(lib)); tm.append(timeit(mine))
    fl = 2.0 * M * K * N
    print(f"{name:9s} M={M} K={K} N={N}: library {min(tl):7.1f} us ({fl / min(tl) / 1e6:6.0f} TF)   hip {min(tm):7.1f} us ({fl / min(tm) / 1e6:6.0f} TF)")
