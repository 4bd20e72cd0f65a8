"""Is fc2 (K = 3072: operand rows 6144 B apart) slowed by L2 channel camping?  Same kernel, same M and N, K = 3072 vs K with 64 / 128
padding columns (rows 6272 / 6400 B apart): TFLOP/s on the useful 3072 columns."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unopose_amd import ops
M, N = 64 * 1374, 768
for K in (3072, 3136, 3200, 3328, 2560, 2048, 1536, 1024, 768, 832):
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16(); b = torch.randn(N, device="cuda")
    for _ in range(3):
        ops.linear_bf16_hip(a, w, b)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        s.record()
        for _ in range(5):
            ops.linear_bf16_hip(a, w, b)
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 5)
    print(f"K={K:5d}: {best*1e3:7.1f} us  {2.0*M*N*K/best/1e9:7.1f} TFLOP/s  ({best*1e3/(K/64):6.2f} us per K-step over the launch)", flush=True)
