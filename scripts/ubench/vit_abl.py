"""Ablation timing of vit_attn_kernel<2,2> (T = 1374, 64 crops x 12 heads): libraries built from
scripts/ubench/vit_attn_abl.hip with -DABL=n (results are wrong by construction; only the time matters)."""
import ctypes, os, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
T = 1374
qkv = torch.randn(64, T, 2304, device="cuda").bfloat16()
out = torch.empty(64, T, 768, device="cuda", dtype=torch.bfloat16)
names = {0: "full", 1: "no exp", 2: "no PV MFMA", 3: "no S MFMA", 4: "never move m, no fix-up", 5: "no fix-up block", 6: "no chunk staging",
         7: "no barriers", 8: "no LDS fragment reads"}
for n, name in names.items():
    lib = ctypes.CDLL(os.path.join(here, f"vit_abl{n}.so"))
    f = lib.unopose_vit_attention
    f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3): f(qkv.data_ptr(), 64, T, 12, out.data_ptr(), st)
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): f(qkv.data_ptr(), 64, T, 12, out.data_ptr(), st)
    e.record(); torch.cuda.synchronize()
    print(f"ABL {n} {name:18s} {s.elapsed_time(e)/10*1e3:7.0f} us")
