"""hipBLASLt with a pinned (swept) solution vs torch's first-heuristic route on the ViT linear shapes."""
import ctypes, os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unopose_amd._lib import ptr, stream_ptr
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(HERE)
SO = os.path.join(HERE, "ubench", "_ltgemm.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":  # on the CPU: python scripts/lt_ab.py build
    import subprocess
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", os.path.join(HERE, "ubench", "ltgemm.hip"),
                           os.path.join(ROOT, "unopose_amd/csrc/abi.hip"), "-L/opt/rocm/lib", "-lhipblaslt", "-o", SO])
    sys.exit(0)
LT = ctypes.CDLL(SO)
LT.unopose_linear_lt.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
def call(name, *a):
    assert getattr(LT, name)(*a) == 0
torch.set_grad_enabled(False)
M = 64 * 1374
def timeit(f, n=20):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for name, K, N in (("qkv", 768, 2304), ("proj", 768, 768), ("fc2", 3072, 768), ("upproj", 3072, 4096), ("fc1", 768, 3072)):
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda").bfloat16(); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    info = (ctypes.c_float * 4)()
    f_lt = lambda: call("unopose_linear_lt", ptr(a), ptr(w), ptr(b), ptr(out), M, N, K, 48, info, stream_ptr())
    f_lt()
    ref = F.linear(a, w, b)
    err = (out.float() - ref.float()).abs().max().item()
    tl = min(timeit(lambda: F.linear(a, w, b)) for _ in range(3)); tp = min(timeit(f_lt) for _ in range(3))
    print(f"{name:7s} K={K} N={N}: torch {tl:7.1f} us | pinned {tp:7.1f} us  (picked #{int(info[0])} of {int(info[1])} timed; sweep best {info[2]:.1f}, first {info[3]:.1f}) max|diff| {err:.3e}")
