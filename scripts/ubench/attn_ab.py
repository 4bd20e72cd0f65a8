"""Same-box A/B timing of token attention (RPE) variants: `build` compiles each .hip (exporting
unopose_token_attention) into _at<i>.so; `run` times them at the in-model shape (2B = 64 clouds, 197 tokens)."""
import ctypes, os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(os.path.dirname(here))
mode, srcs = sys.argv[1], sys.argv[2:]
if mode == "build":
    for i, src in enumerate(srcs):
        flags = src.split("@")[1:]  # path@-DABL=3
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-honor-nans", "-fPIC", "-shared",
               f"-I{root}/include", f"-I{root}/unopose_amd/csrc", *flags, src.split("@")[0], f"{root}/unopose_amd/csrc/abi.hip", "-o",
               os.path.join(here, f"_at{i}.so")]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            sys.exit(r.stderr[-3000:])
    print("built", len(srcs))
else:
    import torch
    B, n, m = int(os.environ.get("AB_B", "64")), 197, 197
    g = torch.Generator().manual_seed(0)
    bf = torch.bfloat16
    q = torch.randn(B, n, 256, generator=g).cuda().to(bf); k = torch.randn(B, m, 256, generator=g).cuda().to(bf)
    qp = (torch.randn(B, n, 1024, generator=g) * 0.1).cuda().to(bf)
    E = torch.randn(B, n, m, 256, generator=g).cuda().to(bf)
    res, outs = [], []
    libs = []
    for i, src in enumerate(srcs):
        lib = ctypes.CDLL(os.path.join(here, f"_at{i}.so"))
        pad = lib.unopose_token_attention_key_pad()
        vt = torch.zeros(B, 256, pad, device="cuda", dtype=bf); vt[:, :, :m] = torch.randn(B, 256, m, generator=g).cuda().to(bf)
        f = lib.unopose_token_attention
        f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                      ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]
        libs.append((f, vt)); outs.append(torch.empty(B, n, 256, device="cuda", dtype=bf)); res.append([])
    st = torch.cuda.current_stream().cuda_stream
    for rep in range(4):
        for i, (f, vt) in enumerate(libs):
            args = (q.data_ptr(), 256, k.data_ptr(), 256, vt.data_ptr(), qp.data_ptr(), 1024, E.data_ptr(), B, n, m, 0.125, outs[i].data_ptr(), st)
            for _ in range(2): f(*args)
            torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10): f(*args)
            e.record(); torch.cuda.synchronize(); res[i].append(s.elapsed_time(e) / 10 * 1e3)
    for i, src in enumerate(srcs):
        us = min(res[i])
        d = (outs[i].float() - outs[0].float()).abs().max().item()
        print(f"{os.path.basename(src):40s} {us:7.1f} us  {E.numel()*2/us/1e6:5.2f} TB/s of E   max|diff vs first| {d:.2e}")
