"""fp32 path's fine similarity at bench size (32 pairs x 2049 x 2049 x 256) on csrc/bmm_f32.hip: python scripts/ubench/bmm_f32_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unopose_amd import ops
torch.set_grad_enabled(False)
a, b = torch.randn(32, 2049, 256, device="cuda"), torch.randn(32, 2049, 256, device="cuda")
def timeit(f, n=5):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
t = timeit(lambda: ops.bmm_nt_f32(a, b)); t2 = timeit(lambda: ops.bmm_nt_f32(a[:, :, :], b[:, :255]))
print(f"2049 x 2049 x 256 x 32: {t:.0f} us ({2.0 * 32 * 2049 * 2049 * 256 / t / 1e6:.0f} TF fp32);   32 x 32 form on 2049 x 255: {t2:.0f} us ({2.0 * 32 * 2049 * 255 * 256 / t2 / 1e6:.0f} TF)")
