"""A/B of csrc/attn_f32.hip's ViT attention against the version at a git ref (default HEAD): bitwise comparison and time.
    python scripts/ubench/attn_f32_ab.py build [ref]     here
    python scripts/ubench/attn_f32_ab.py                 on the GPU box"""
import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "unopose_amd", "csrc")
FLAGS = ["/opt/rocm/bin/hipcc", "-O3", "-fno-slp-vectorize", "-fno-vectorize", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-ffp-contract=off",
         "-fno-honor-nans", "-I", CSRC]
if len(sys.argv) > 1 and sys.argv[1] == "build":
    ref = sys.argv[2] if len(sys.argv) > 2 else "HEAD"
    old = os.path.join(HERE, "_attn_f32_old.hip")
    open(old, "w").write(subprocess.check_output(["git", "-C", ROOT, "show", f"{ref}:unopose_amd/csrc/attn_f32.hip"], text=True))
    subprocess.check_call(FLAGS + [old, os.path.join(CSRC, "abi.hip"), "-o", os.path.join(HERE, "_attn_f32_old.so")])
    subprocess.check_call(FLAGS + [os.path.join(CSRC, "attn_f32.hip"), os.path.join(CSRC, "abi.hip"), "-o", os.path.join(HERE, "_attn_f32_new.so")])
    sys.exit(0)
import torch
libs = {n: ctypes.CDLL(os.path.join(HERE, f"_attn_f32_{n}.so")) for n in ("old", "new")}
for L in libs.values():
    L.unopose_vit_attention_f32.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for B, T in ((64, 1374), (64, 261), (3, 197)):
    qkv = torch.randn(B, T, 2304, device="cuda")
    outs = {}
    for n, L in libs.items():
        out = torch.empty(B, T, 768, device="cuda")
        f = lambda: L.unopose_vit_attention_f32(qkv.data_ptr(), B, T, 12, out.data_ptr(), st)
        assert f() == 0; torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            s.record()
            for _ in range(3): f()
            e.record(); torch.cuda.synchronize(); best = min(best, s.elapsed_time(e) / 3)
        outs[n] = (out.clone(), best)
    print(f"B={B} T={T}: old {outs['old'][1]*1e3:8.1f} us  new {outs['new'][1]*1e3:8.1f} us  bit-identical: {torch.equal(outs['old'][0], outs['new'][0])}", flush=True)
