"""Same-box timing of the residual-epilogue GEMM (EPI 5: proj / fc2 of a ViT-B block at M = 64 x 1374) for builds of csrc/gemm.hip with different
-D flags, beside the plain epilogue (EPI 0) of the same build.
Build here (CPU):  python scripts/ubench/resid_epi_abl.py build base= noload=-DGEMM_EABL=4 ...
Run on the GPU:    python scripts/ubench/resid_epi_abl.py run base noload ..."""
import ctypes, os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
def so(n): return os.path.join(HERE, f"_re_{n}.so")
if sys.argv[1] == "build":
    def one(spec):
        name, _, flags = spec.partition("=")
        flags = [f for f in flags.split(",") if f]
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-fno-honor-nans", "-ffp-contract=off",
               "-I", os.path.join(ROOT, "unopose_amd/csrc"), "-I", os.path.join(ROOT, "include"), *flags, os.path.join(ROOT, "unopose_amd/csrc/gemm.hip"),
               os.path.join(ROOT, "unopose_amd/csrc/gemm_small.hip"), os.path.join(ROOT, "unopose_amd/csrc/abi.hip"), "-o", so(name)]
        subprocess.check_call(cmd)
        return name
    with ThreadPoolExecutor(4) as ex:
        print(list(ex.map(one, sys.argv[2:])))
    sys.exit(0)
import torch
torch.set_grad_enabled(False)
names = sys.argv[2:]
M = 64 * 1374
Mp = (M + 255) // 256 * 256
libs = {n: ctypes.CDLL(so(n)) for n in names}
P, L_, I = ctypes.c_void_p, ctypes.c_long, ctypes.c_int
for L in libs.values():
    L.unopose_linear_bf16.argtypes = [P] * 4 + [L_, I, I, I, P]
    L.unopose_linear_bf16_residual.argtypes = [P] * 6 + [L_, I, I, P]
def timeit(f, n=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
st = P(torch.cuda.current_stream().cuda_stream)
for nm, K, N in (("proj", 768, 768), ("fc2", 3072, 768)):
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    x = torch.randn(M, N, device="cuda"); stats = torch.empty(Mp, N // 256, 2, device="cuda")
    f0 = {n: (lambda L=L: L.unopose_linear_bf16(a.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, 0, st)) for n, L in libs.items()}
    f5 = {n: (lambda L=L: L.unopose_linear_bf16_residual(a.data_ptr(), w.data_ptr(), b.data_ptr(), x.data_ptr(), out.data_ptr(), stats.data_ptr(), M, N, K, st))
          for n, L in libs.items()}
    t0, t5 = {n: [] for n in names}, {n: [] for n in names}
    for r in range(7):
        for n in names:
            t0[n].append(timeit(f0[n])); t5[n].append(timeit(f5[n]))
    for n in names:
        print(f"{nm:5s} {n:14s} EPI 0 {sorted(t0[n])[3]:7.1f} us   EPI 5 {sorted(t5[n])[3]:7.1f} us", flush=True)
