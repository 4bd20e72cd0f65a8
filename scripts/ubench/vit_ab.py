"""Same-box A/B timing of vit_attn.hip variants: each argv path is a .hip source exporting
unopose_vit_attention; built here into scripts/ubench/_ab<i>.so (host: no GPU needed) with `build`,
timed on the GPU with `run` (T = 1374 and 261, 64 crops x 12 heads, interleaved repeats)."""
import ctypes, os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(os.path.dirname(here))
mode, srcs = sys.argv[1], sys.argv[2:]
if mode == "build":
    for i, src in enumerate(srcs):
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-honor-nans", "-fPIC", "-shared",
               f"-I{root}/include", f"-I{root}/unopose_amd/csrc", src, f"{root}/unopose_amd/csrc/abi.hip", "-o", os.path.join(here, f"_ab{i}.so")]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            sys.exit(r.stderr[-3000:])
    print("built", len(srcs))
else:
    import torch
    libs = []
    for i, src in enumerate(srcs):
        lib = ctypes.CDLL(os.path.join(here, f"_ab{i}.so"))
        lib.unopose_vit_attention.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        libs.append(lib)
    for T in (1374, 261):
        qkv = torch.randn(64, T, 2304, device="cuda").bfloat16()
        outs = [torch.empty(64, T, 768, device="cuda", dtype=torch.bfloat16) for _ in libs]
        st = torch.cuda.current_stream().cuda_stream
        res = [[] for _ in libs]
        for rep in range(4):
            for i, lib in enumerate(libs):
                for _ in range(2): lib.unopose_vit_attention(qkv.data_ptr(), 64, T, 12, outs[i].data_ptr(), st)
                torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(10): lib.unopose_vit_attention(qkv.data_ptr(), 64, T, 12, outs[i].data_ptr(), st)
                e.record(); torch.cuda.synchronize()
                res[i].append(s.elapsed_time(e) / 10 * 1e3)
        for i, src in enumerate(srcs):
            d = (outs[i].float() - outs[0].float()).abs().max().item()
            print(f"T={T} {os.path.basename(src):28s} {min(res[i]):7.1f} us (min of 4)  max|diff vs first| {d:.2e}")
