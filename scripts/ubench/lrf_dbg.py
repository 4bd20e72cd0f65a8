"""Variants of lrf_global beside a library GEMM (scripts/ubench/lrf_dbg.hip).  build: python scripts/ubench/lrf_dbg.py build"""
import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "_lrf_dbg.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "--offload-arch=gfx950",
                           os.path.join(HERE, "lrf_dbg.hip"), os.path.join(HERE, "..", "..", "unopose_amd", "csrc", "abi.hip"), "-o", so])
    sys.exit(0)
import torch
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from unopose_amd.synthetic import make_batch
from unopose_amd import ops
lib = ctypes.CDLL(so)
P = ctypes.c_void_p
ep, _, _ = make_batch(3, S=224, seed=53, device="cuda")
side = torch.cuda.Stream()
a = torch.randn(8192, 768, device="cuda").bfloat16(); w = torch.randn(3072, 768, device="cuda").bfloat16()
pts = ep["tem1_pts"].contiguous(); B, N, _ = pts.shape
ref = ops.lrf_global(pts).clone()
import torch.nn.functional as F
BF = torch.bfloat16
qkv = torch.randn(6, 1374, 2304, device="cuda").to(BF)
bias = torch.randn(3072, device="cuda")
a_s = torch.randn(12608, 256, device="cuda").to(BF); w_s = torch.randn(256, 256, device="cuda").to(BF); w_s2 = torch.randn(512, 256, device="cuda").to(BF)
a_d = torch.randn(131136, 256, device="cuda").to(BF)
a32 = torch.randn(8192, 768, device="cuda"); w32 = torch.randn(3072, 768, device="cuda")
x32 = torch.randn(8192, 768, device="cuda"); ln = torch.nn.LayerNorm(768).cuda()
a_d32 = a_d.float(); w_s32 = w_s.float(); f197 = torch.randn(64, 197, 256, device="cuda")
kv_a = torch.randn(256, 64, 196, device="cuda"); kv_b = torch.randn(256, 196, 64, device="cuda")
loads = {
    "nothing": lambda: None,
    "own gemm (csrc/gemm.hip)": lambda: ops.linear_bf16_hip(a, w, bias, True),
    "vit_attention": lambda: ops.vit_attention(qkv, 12),
    "add_layernorm": lambda: ops.add_layernorm(x32, None, ln, BF),
    "hipBLASLt bf16 8192x768x3072": lambda: F.linear(a, w),
    "hipBLASLt bf16 12608x256x256": lambda: F.linear(a_s, w_s),
    "hipBLASLt bf16 12608x256x512": lambda: F.linear(a_s, w_s2),
    "hipBLASLt bf16 131136x256x256": lambda: F.linear(a_d, w_s),
    "hipBLASLt fp32 8192x768x3072": lambda: F.linear(a32, w32),
    "hipBLASLt fp32 131136x256x256 (PE mlp3)": lambda: F.linear(a_d32, w_s32),
    "hipBLASLt fp32 bmm 64x197x197x256": lambda: torch.bmm(f197, f197.transpose(1, 2)),
    "hipBLASLt fp32 bmm 256x64x196x64 (kv)": lambda: torch.bmm(kv_a, kv_b),
}
def runw(stream):
    out = torch.empty_like(pts)
    lib.run_variant(7, P(pts.data_ptr()), B, N, P(out.data_ptr()), P(stream.cuda_stream), None)
    return out
o0 = runw(torch.cuda.current_stream()).clone(); torch.cuda.synchronize()
for name, f in loads.items():
    nbad = 0
    for it in range(200):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            o = runw(side)
        for _ in range(4): f()
        torch.cuda.synchronize()
        nbad += (o - o0).abs().max().item() > 0
    print(f"one-wave LRF beside {name:34s}: corrupted {nbad} / 200")
