"""Ablations of csrc/gemm.hip (one piece removed at a time) -- which phase bounds the K loop?
Build here (CPU): python scripts/ubench/gemm_abl.py build ; run on the GPU box: python scripts/ubench/gemm_abl.py"""
import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
NAMES = {10: "full (K rotation)", 12: "no MFMA", 13: "no fragment reads", 15: "DMA only, vmcnt(0) per tile... see 16",
         16: "placeholder"}
NAMES = {4410: "full", 14410: "full, A tiles nt", 24410: "full, W tiles nt", 34410: "full, A and W nt"}
def so(v): return os.path.join(HERE, f"_gemm_abl{v}.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    for v in NAMES:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-fno-honor-nans",
                               f"-DGEMM_ABL={v % 10}", f"-DGEMM_ROT={(v // 10) % 10}", f"-DGEMM_GM={max(1, (v // 100) % 10)}", f"-DGEMM_SKEW={(v // 1000) % 10}", f"-DGEMM_NT={v // 10000}", os.path.join(ROOT, "unopose_amd/csrc/gemm.hip"), os.path.join(ROOT, "unopose_amd/csrc/abi.hip"), "-o", so(v)])
    sys.exit(0)
import torch
torch.set_grad_enabled(False)
M = 64 * 1374
def timeit(f, n=20):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
libs = {v: ctypes.CDLL(so(v)) for v in NAMES}
for K, N in ((768, 2304), (768, 768), (768, 3072), (3072, 768)):
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for v, L in libs.items():
        fn = L.unopose_linear_bf16
        fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        f = lambda: fn(a.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, 0, st)
        t = min(timeit(f) for _ in range(3))
        print(f"K={K} N={N} {NAMES[v]:22s} {t:8.1f} us  ({2.0 * M * K * N / t / 1e6:6.0f} TF-equivalent)")
