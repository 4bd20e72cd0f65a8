"""Same-box A/B of csrc/vit_attn.hip (or another source) built with different -D flags at the bench shape (64 images x 1374 tokens x 12 heads).
  build (CPU):  python scripts/ubench/vit_attn_var.py build name1=-DX=1,-DY=2 name2=@scripts/ubench/other.hip ...
  run (GPU):    python scripts/ubench/vit_attn_var.py run name1 name2 ...        interleaved rounds, min / median, error vs fp32 torch
  pmc target:   python scripts/ubench/vit_attn_var.py one name                   a few launches of ONE build (for rocprofv3 --pmc)"""
import ctypes, os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
F32 = bool(os.environ.get("VA_F32"))  # VA_F32=1: the fp32-class kernel (split layout in and out, csrc/vit_attn_f32s.hip)
def so(n): return os.path.join(HERE, f"_va{'f' if F32 else ''}_{n}.so")
if sys.argv[1] == "build":
    def one(spec):
        name, _, flags = spec.partition("=")
        flags = [f for f in flags.split(",") if f]
        src = os.path.join(ROOT, "unopose_amd/csrc/vit_attn_f32s.hip" if os.environ.get("VA_F32") else "unopose_amd/csrc/vit_attn.hip")
        for f in flags:
            if f.startswith("@"): src = os.path.join(ROOT, f[1:])
        flags = [f for f in flags if not f.startswith("@")]
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-fno-honor-nans", "-ffp-contract=off",
               "-fno-slp-vectorize", "-fno-vectorize", "-I", os.path.join(ROOT, "unopose_amd/csrc"), "-I", os.path.join(ROOT, "include"), *flags, src,
               os.path.join(ROOT, "unopose_amd/csrc/abi.hip"), "-o", so(name)]
        subprocess.check_call(cmd)
        return name
    with ThreadPoolExecutor(6) as ex:
        print(list(ex.map(one, sys.argv[2:])))
    sys.exit(0)
import torch
torch.set_grad_enabled(False)
names = sys.argv[2:]
B, T, H = int(os.environ.get("VA_B", 64)), int(os.environ.get("VA_T", 1374)), 12
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B, T, 3, H, 64, generator=g).cuda()
def split(x):  # (..., C) fp32 -> (..., C/32, 2, 32) bf16: per 32-channel block [hi | lo]
    hi = x.bfloat16(); lo = (x - hi.float()).bfloat16()
    return torch.stack([hi.reshape(*x.shape[:-1], -1, 32), lo.reshape(*x.shape[:-1], -1, 32)], dim=-2).contiguous()
def unsplit(s): return (s[..., 0, :].float() + s[..., 1, :].float()).reshape(*s.shape[:-3], -1)
qin = split(qkv.reshape(B, T, 3 * H * 64)) if F32 else qkv.bfloat16()
if not F32: qkv = qin
libs = {n: ctypes.CDLL(so(n)) for n in names}
entry = "unopose_vit_attention_f32_ss" if F32 else "unopose_vit_attention"
for L in libs.values():
    getattr(L, entry).argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
outs = {n: (torch.empty(B, T, H * 2, 2, 32, device="cuda", dtype=torch.bfloat16) if F32 else torch.empty(B, T, H * 64, device="cuda", dtype=torch.bfloat16)) for n in names}
fs = {n: (lambda L=L, n=n: getattr(L, entry)(qin.data_ptr(), B, T, H, outs[n].data_ptr(), st)) for n, L in libs.items()}
if sys.argv[1] == "one":
    for _ in range(4): fs[names[0]]()
    torch.cuda.synchronize()
    sys.exit(0)
q, k, v = (qkv[:2, :, i].float().permute(0, 2, 1, 3) for i in range(3))
ref = torch.nn.functional.scaled_dot_product_attention(q, k, v).permute(0, 2, 1, 3).reshape(2, T, H * 64)
def timeit(f, n=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
ts = {n: [] for n in names}
for n, f in fs.items():
    outs[n].zero_(); f(); torch.cuda.synchronize()
for r in range(7):
    for n, f in fs.items(): ts[n].append(timeit(f))
flops = 4.0 * B * H * T * T * 64
for n in names:
    t = sorted(ts[n]); err = ((unsplit(outs[n][:2]) if F32 else outs[n][:2].float()) - ref).abs().max().item()
    print(f"{n:16s} min {t[0]:7.1f} us  med {t[len(t) // 2]:7.1f} us  ({flops / t[len(t) // 2] / 1e6:5.0f} TF)  maxerr {err:.3g}", flush=True)
