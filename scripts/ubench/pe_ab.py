"""Same-box A/B of PE kernel variants (bf16x3 path, 64 clouds x 2048 points, S = 256 and 64)."""
import ctypes, os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(os.path.dirname(here))
sys.path.insert(0, root)
mode, srcs = sys.argv[1], sys.argv[2:]
if mode == "build":
    for i, src in enumerate(srcs):
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-fno-vectorize", "-std=c++17", "-ffp-contract=off", "-fno-honor-nans", "-fPIC", "-shared",
               f"-I{root}/include", f"-I{root}/unopose_amd/csrc", *src.split("@")[1:], src.split("@")[0], f"{root}/unopose_amd/csrc/abi.hip",
               "-o", os.path.join(here, f"_pe{i}.so")]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            sys.exit(r.stderr[-3000:])
    print("built", len(srcs))
else:
    import torch
    from unopose_amd import ops
    from unopose_amd.model import UNOPose, default_model_cfg
    from unopose_amd.synthetic import make_batch, trained_like_
    torch.set_grad_enabled(False)
    m = trained_like_(UNOPose(default_model_cfg())).cuda().eval()
    pe = m.fine_point_matching.PE
    batch, _, _ = make_batch(32, device="cuda")
    rad = torch.norm(batch["tem1_pts"] - batch["tem1_pts"].mean(1, keepdim=True), dim=2).max(1)[0]
    x = (batch["pts"] / (rad.reshape(-1, 1, 1) + 1e-6))
    x = torch.cat([x, x + 0.01], 0).contiguous()  # 64 clouds
    P = ctypes.c_void_p
    for mlp, r, S in ((pe.mlp2, pe.r2, pe.ns2), (pe.mlp1, pe.r1, pe.ns1)):
        ref = ops.pe_group_mlp_max(x, r, S, mlp, bf16x3=True)
        image = mlp._hip_cache[2]
        res, outs, fs = [], [], []
        for i, src in enumerate(srcs):
            f = ctypes.CDLL(os.path.join(here, f"_pe{i}.so")).unopose_pe_group_mlp_max_packed
            f.argtypes = [P, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int, P, P, P]
            fs.append(f); outs.append(torch.empty(64, 2048, 128, device="cuda")); res.append([])
        st = torch.cuda.current_stream().cuda_stream
        for rep in range(3):
            for i, f in enumerate(fs):
                a = (x.data_ptr(), 64, 2048, float(r), int(S), image.data_ptr(), outs[i].data_ptr(), st)
                assert f(*a) == 0
                torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(3): f(*a)
                e.record(); torch.cuda.synchronize(); res[i].append(s.elapsed_time(e) / 3 * 1e3)
        for i, src in enumerate(srcs):
            print(f"S={S:3d} {os.path.basename(src):28s} {min(res[i]):8.1f} us   max|diff vs product| {(outs[i]-ref).abs().max().item():.2e}")
