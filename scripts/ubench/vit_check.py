import ctypes, os, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, sys.argv[1]))
lib.unopose_vit_attention.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
for T in [int(a) for a in sys.argv[2:]]:
    g = torch.Generator().manual_seed(T)
    qkv = torch.randn(2, T, 2304, generator=g).cuda().bfloat16()
    out = torch.empty(2, T, 768, device="cuda", dtype=torch.bfloat16)
    lib.unopose_vit_attention(qkv.data_ptr(), 2, T, 12, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    q, k, v = qkv.float().reshape(2, T, 3, 12, 64).permute(2, 0, 3, 1, 4)
    ref = (torch.softmax(q @ k.transpose(-1, -2) / 8, -1) @ v).permute(0, 2, 1, 3).reshape(2, T, 768)
    err = (out.float() - ref).abs()
    bad = (err > 0.05).nonzero()
    print(T, "max err", err.max().item(), "n bad", len(bad), "first bad (b,t,c):", bad[:3].tolist(), "bad tokens range", (bad[:, 1].min().item(), bad[:, 1].max().item()) if len(bad) else None)
