"""Variants of the four-wave GEMM stream (scripts/ubench/gemm4w/gen4w) as stand-alone libraries, and their check / A-B on the GPU.

Build (CPU):  python scripts/ubench/gemm4w/g4w_var.py build base cap6=cap:6 n316=n3_mid:16 ...      (name=key:value,key:value generator arguments)
Run (GPU):    python scripts/ubench/gemm4w/g4w_var.py run base cap6 ...          correctness on a set of shapes, then interleaved timing rounds
              against the 8-wave kernel (the same library with the four-wave path switched off)."""
import ctypes, os, shutil, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
CSRC = os.path.join(ROOT, "unopose_amd/csrc")
def so(n): return os.path.join(HERE, f"_g4w_{n}.so")
if sys.argv[1] == "build":
    from gen4w import emit
    def one(spec):
        name, _, kv = spec.partition("=")
        kw = {}
        for item in [x for x in kv.split(",") if x]:
            k, v = item.split(":")
            kw[k] = int(v)
        d = os.path.join(HERE, f"_g4w_{name}")
        os.makedirs(os.path.join(d, "_gen"), exist_ok=True)
        emit.write_all(os.path.join(d, "_gen"), **kw)
        for f in ("gemm4w.hip", "gemm4w_clobbers.h"):
            shutil.copy(os.path.join(HERE, f), os.path.join(d, f))
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-fno-honor-nans", "-ffp-contract=off",
               "-fno-slp-vectorize", "-fno-vectorize", "-DUNOPOSE_PROBE_GEMM4W", "-I", CSRC, os.path.join(d, "gemm4w.hip"), os.path.join(CSRC, "gemm.hip"),
               os.path.join(CSRC, "gemm_small.hip"), os.path.join(CSRC, "abi.hip"), "-o", so(name)]
        subprocess.check_call(cmd)
        return name
    with ThreadPoolExecutor(4) as ex:
        print(list(ex.map(one, sys.argv[2:])))
    sys.exit(0)
import torch
torch.set_grad_enabled(False)
names = sys.argv[2:]
libs = {n: ctypes.CDLL(so(n)) for n in names}
for L in libs.values():
    L.unopose_linear_bf16.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    L.unopose_gemm4w_enable.argtypes = [ctypes.c_int]
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
act = {0: lambda x: x, 1: torch.nn.functional.gelu, 2: torch.relu}
def check(L, M, N, K, epi, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    a = torch.randn(M, K, device="cuda", generator=g).bfloat16(); w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda", generator=g)
    out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    L.unopose_gemm4w_enable(2)
    rc = L.unopose_linear_bf16(a.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, epi, st)
    torch.cuda.synchronize()
    ref = torch.nn.functional.linear(a.float(), w.float(), b)
    if epi == 1: ref = ref.bfloat16().float()
    ref = act[epi](ref)
    bad = torch.isnan(out.float()).sum().item()
    err = (out.float() - ref).abs().max().item() if not bad else float("nan")
    # against the 8-wave kernel of the same library: the same products in another summation order
    out8 = torch.empty_like(out)
    L.unopose_gemm4w_enable(0)
    L.unopose_linear_bf16(a.data_ptr(), w.data_ptr(), b.data_ptr(), out8.data_ptr(), M, N, K, epi, st)
    torch.cuda.synchronize()
    d8 = (out.float() - out8.float()).abs().max().item()
    return rc, bad, err, d8
ok = True
if not os.environ.get("G4W_SKIP_CHECK"):
    for n, L in libs.items():
        for (M, N, K, epi) in [(256, 256, 512, 0), (904, 1024, 512, 0), (2048, 3072, 768, 0), (5000, 768, 768, 0), (4096, 768, 3072, 0), (3000, 3072, 768, 1),
                               (1000, 512, 448, 2), (64 * 1374, 2304, 768, 0), (64 * 1374, 3072, 768, 1), (64 * 1374, 768, 3072, 0), (64 * 1374, 768, 768, 0),
                               (64 * 261, 768, 768, 0)][:3 if os.environ.get("G4W_QUICK") else None]:
            for rep in range(int(os.environ.get("G4W_REPS", 2)) if M > 50000 else 1):
                rc, bad, err, d8 = check(L, M, N, K, epi, seed=rep)
                flag = "" if (rc == 0 and bad == 0 and err < 0.06) else "   <-- FAIL"
                ok = ok and not flag
                print(f"check {n:10s} M={M:6d} N={N:5d} K={K:5d} epi={epi} rc={rc} nan={bad} maxerr={err:.4g} vs8wave={d8:.4g}{flag}", flush=True)
print("CHECK", "OK" if ok else "FAILED", flush=True)
if os.environ.get("G4W_QUICK"): sys.exit(0 if ok else 1)
if os.environ.get("G4W_STRESS"):
    # many launches of the four ViT shapes, synchronised every 10: the process dies on a fault, the log tells how far it came
    M = 64 * 1374
    for nm, K, N, epi in (("qkv", 768, 2304, 0), ("proj", 768, 768, 0), ("fc1+gelu", 768, 3072, 1), ("fc2", 3072, 768, 0)):
        a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
        b = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        for n, L in libs.items():
            L.unopose_gemm4w_enable(1)
            for r in range(int(os.environ["G4W_STRESS"])):
                for _ in range(10): L.unopose_linear_bf16(a.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, epi, st)
                torch.cuda.synchronize()
            print("stress", n, nm, "ok", flush=True)
    sys.exit(0)
def timeit(f, n=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
M = int(os.environ.get("GV_M", 64 * 1374))
for nm, K, N, epi in (("qkv", 768, 2304, 0), ("proj", 768, 768, 0), ("fc1+gelu", 768, 3072, 1), ("fc2", 3072, 768, 0)):
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    fs = {}
    for n, L in libs.items():
        def f(L=L, on=1):
            L.unopose_gemm4w_enable(on)
            L.unopose_linear_bf16(a.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, epi, st)
        fs[n] = f
    first = next(iter(libs.values()))
    fs["8wave"] = lambda L=first: (L.unopose_gemm4w_enable(0), L.unopose_linear_bf16(a.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, epi, st))
    ts = {n: [] for n in fs}
    for r in range(7):
        for n, f in fs.items():
            ts[n].append(timeit(f))
    for n in fs:
        t = sorted(ts[n]); mn, med = t[0], t[len(t) // 2]
        print(f"{nm:9s} K={K:4d} N={N:4d} {n:14s} min {mn:7.1f} us  med {med:7.1f} us  ({2.0 * M * K * N / med / 1e6:6.0f} TF)", flush=True)
    print()
