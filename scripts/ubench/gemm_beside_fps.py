"""How much does a kernel that holds a few CUs for a long time (the 5000 -> 2048 FPS of the reference clouds: one 512-thread workgroup per
cloud, ~1.9 ms) stretch the persistent GEMM (one workgroup per CU, tiles statically dealt) that runs beside it on another stream?
    python scripts/ubench/gemm_beside_fps.py"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unopose_amd import _lib
from unopose_amd.pointnet2 import _ext
torch.set_grad_enabled(False)
M = 64 * 1374
shapes = (("qkv", 768, 2304, 0), ("proj", 768, 768, 0), ("fc1", 768, 3072, 1), ("fc2", 3072, 768, 0))
ops = []
for nm, K, N, epi in shapes:
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda"); c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ops.append((nm, a, w, b, c, N, K, epi))
tem = torch.rand(32, 5000, 3, device="cuda")
side = torch.cuda.Stream()
def block(n=3):
    st = _lib.stream_ptr()
    for _ in range(n):
        for nm, a, w, b, c, N, K, epi in ops:
            _lib.call("unopose_linear_bf16", _lib.ptr(a), _lib.ptr(w), _lib.ptr(b), _lib.ptr(c), M, N, K, epi, st)
def timed(with_fps):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if with_fps:
        with torch.cuda.stream(side):
            _ext.furthest_point_sampling(tem, 2048); _ext.furthest_point_sampling(tem, 2048)
    s.record(); block(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)
for _ in range(2): timed(False); timed(True)
a = sorted(timed(False) for _ in range(7)); b = sorted(timed(True) for _ in range(7))
print(f"12 ViT GEMMs alone: {a[3]:.3f} ms   beside two 5000->2048 FPS launches (32 workgroups, ~3.8 ms): {b[3]:.3f} ms   stretch {b[3]-a[3]:.3f} ms")
