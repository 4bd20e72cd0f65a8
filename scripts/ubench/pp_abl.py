"""Phase ablations of the ping-pong attention experiment (scripts/ubench/vit_attn_pp.hip): full / no MFMA (softmax phases
alone) / no softmax (matrix phases alone), plus a check of the full variant against the oracle attention core.
Build: python scripts/ubench/pp_abl.py build ; run on the GPU: python scripts/ubench/pp_abl.py"""
import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE))
NAMES = {0: "full", 1: "no MFMAs (softmax phases alone)", 2: "no softmax (matrix phases alone)"}
so = lambda v: os.path.join(HERE, f"_pp_abl{v}.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    for v in NAMES:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-fno-honor-nans", "-ffp-contract=off",
                               f"-DPP_ABL={v}", os.path.join(HERE, "vit_attn_pp.hip"), os.path.join(ROOT, "unopose_amd/csrc/abi.hip"), "-o", so(v)])
    sys.exit(0)
import torch
sys.path.insert(0, ROOT)
from oracle import unopose_ref as R
T = 1374
qkv = torch.randn(64, T, 2304, device="cuda").bfloat16(); out = torch.empty(64, T, 768, device="cuda", dtype=torch.bfloat16)
for v in NAMES:
    L = ctypes.CDLL(so(v)); fn = L._Z24unopose_vit_attention_ppPKviiiPvP12ihipStream_t
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    f = lambda: fn(qkv.data_ptr(), 64, T, 12, out.data_ptr(), st)
    for _ in range(3): f()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): f()
    e.record(); torch.cuda.synchronize()
    print(f"{NAMES[v]:40s} {s.elapsed_time(e) / 20 * 1e3:7.1f} us")
    if v == 0:
        ref = R.vit_attention_core(qkv[:2].float().cpu(), 12)
        print("   full variant vs oracle attention core: max err %.2e" % (out[:2].float().cpu() - ref).abs().max().item())
