"""Same-box A/B of geometric-embedding kernel variants (bf16 path, 64 clouds x 197 points)."""
import ctypes, os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(os.path.dirname(here))
sys.path.insert(0, root)
mode, srcs = sys.argv[1], sys.argv[2:]
if mode == "build":
    for i, src in enumerate(srcs):
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-honor-nans", "-fPIC", "-shared",
               f"-I{root}/include", f"-I{root}/unopose_amd/csrc", *src.split("@")[1:], src.split("@")[0], f"{root}/unopose_amd/csrc/abi.hip", "-o", os.path.join(here, f"_ge{i}.so")]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            sys.exit(r.stderr[-3000:])
    print("built", len(srcs))
else:
    import torch
    from unopose_amd import ops
    from unopose_amd.model import UNOPose, default_model_cfg
    torch.set_grad_enabled(False)
    m = UNOPose(default_model_cfg()).cuda().eval().geo_embedding
    B, n = 64, 197
    pts = torch.rand(B, n, 3, device="cuda")
    ref = ops.geo_embedding(pts, m, out_dtype=torch.bfloat16)  # fills m._hip_cache
    _, wdh, wdl, wah, wal, bias, div = m._hip_cache
    P = ctypes.c_void_p
    res, outs, fs = [], [], []
    for i, src in enumerate(srcs):
        f = ctypes.CDLL(os.path.join(here, f"_ge{i}.so")).unopose_geo_embedding
        f.argtypes = [P, ctypes.c_int, ctypes.c_int, P, P, P, P, P, P, ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_int,
                      ctypes.c_int, P, P, P]
        fs.append(f); outs.append(torch.empty(B, n, n, 256, dtype=torch.bfloat16, device="cuda")); res.append([])
    knn = torch.empty(B, n, 3, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for rep in range(4):
        for i, f in enumerate(fs):
            a = (pts.data_ptr(), B, n, wdh.data_ptr(), wdl.data_ptr(), wah.data_ptr(), wal.data_ptr(), bias.data_ptr(), div.data_ptr(),
                 float(m.sigma_d), float(m.factor_a), int(m.reduction_a == "mean"), 0, 1, knn.data_ptr(), outs[i].data_ptr(), st)
            for _ in range(2): assert f(*a) == 0
            torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5): f(*a)
            e.record(); torch.cuda.synchronize(); res[i].append(s.elapsed_time(e) / 5 * 1e3)
    for i, src in enumerate(srcs):
        d = (outs[i].float() - ref.float()).abs().max().item()
        print(f"{os.path.basename(src):24s} {min(res[i]):8.1f} us (knn + embed)   max|diff vs product kernel| {d:.2e}")
