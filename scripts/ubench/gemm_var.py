"""Same-box A/B of csrc/gemm.hip built with different -D flags, on the four ViT linear shapes at M = 64 x 1374.
Build here (CPU):  python scripts/ubench/gemm_var.py build name1=-DX=1,-DY=2 name2=...
Run on the GPU:    python scripts/ubench/gemm_var.py run name1 name2 ...     (interleaved rounds, min and median)"""
import ctypes, os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
def so(n): return os.path.join(HERE, f"_gv_{n}.so")
if sys.argv[1] == "build":
    def one(spec):
        name, _, flags = spec.partition("=")
        flags = [f for f in flags.split(",") if f]
        src = os.path.join(ROOT, "unopose_amd/csrc/gemm.hip")  # a flag "@path" names another source file (e.g. last round's kernel)
        for f in flags:
            if f.startswith("@"): src = os.path.join(ROOT, f[1:])
        flags = [f for f in flags if not f.startswith("@")]
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-fno-honor-nans", "-ffp-contract=off",
               "-I", os.path.join(ROOT, "unopose_amd/csrc"), *flags, src, os.path.join(ROOT, "unopose_amd/csrc/gemm_small.hip"), os.path.join(ROOT, "unopose_amd/csrc/abi.hip"),
               "-o", so(name)]
        subprocess.check_call(cmd)
        return name
    with ThreadPoolExecutor(6) as ex:
        print(list(ex.map(one, sys.argv[2:])))
    sys.exit(0)
import torch
torch.set_grad_enabled(False)
names = sys.argv[2:]
M = int(os.environ.get("GV_M", 64 * 1374))
libs = {n: ctypes.CDLL(so(n)) for n in names}
for L in libs.values():
    L.unopose_linear_bf16.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
def timeit(f, n=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for nm, K, N, epi in (("qkv", 768, 2304, 0), ("proj", 768, 768, 0), ("fc1+gelu", 768, 3072, 1), ("fc2", 3072, 768, 0)):
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ref = torch.nn.functional.linear(a[:4096].float(), w.float(), b)
    if epi == 1: ref = torch.nn.functional.gelu(ref)
    fs = {n: (lambda L=L: L.unopose_linear_bf16(a.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, epi, st)) for n, L in libs.items()}
    errs = {}
    for n, f in fs.items():
        out.zero_(); f(); torch.cuda.synchronize()
        errs[n] = (out[:4096].float() - ref).abs().max().item()
        tail = (out[-300:].float() - (torch.nn.functional.gelu if epi == 1 else (lambda x: x))(torch.nn.functional.linear(a[-300:].float(), w.float(), b))).abs().max().item()
        errs[n] = max(errs[n], tail)
    ts = {n: [] for n in names}
    for r in range(7):
        for n, f in fs.items():
            ts[n].append(timeit(f))
    for n in names:
        t = sorted(ts[n]); mn, med = t[0], t[len(t) // 2]
        print(f"{nm:9s} K={K:4d} N={N:4d} {n:18s} min {mn:7.1f} us  med {med:7.1f} us  ({2.0 * M * K * N / med / 1e6:6.0f} TF)  maxerr {errs[n]:.3g}", flush=True)
    print()
