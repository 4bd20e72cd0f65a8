import os, sys, torch
sys.path.insert(0, os.getcwd())
from unopose_amd import ops
from unopose_amd.synthetic import make_batch
torch.set_grad_enabled(False)
ep, _, _ = make_batch(3, S=224, seed=53, device="cuda")
pts, tem = ep["pts"], ep["tem1_pts"]
base0, base1 = ops.lrf_global(pts).clone(), ops.lrf_global(tem).clone()
side = torch.cuda.Stream()
BF = torch.bfloat16
qkv = torch.randn(6, 261, 2304, device="cuda").to(BF)
a = torch.randn(8192, 768, device="cuda").to(BF); w = torch.randn(3072, 768, device="cuda").to(BF); bias = torch.randn(3072, device="cuda")
x32 = torch.randn(8192, 768, device="cuda"); ln = torch.nn.LayerNorm(768).cuda(); gam = torch.rand(768, device="cuda")
xyz = torch.randn(3, 5000, 3, device="cuda")
loads = {
    "vit_attention": lambda: ops.vit_attention(qkv, 12),
    "own gemm": lambda: ops.linear_bf16_hip(a, w, bias, True),
    "lib gemm": lambda: torch.nn.functional.linear(a, w),
    "add_layernorm": lambda: ops.add_layernorm(x32, None, ln, BF),
    "fps": lambda: ops.furthest_point_sample(xyz, 2048),
    "elementwise": lambda: x32 * 2 + 1,
}
for name, f in loads.items():
    bad = 0
    for it in range(100):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            o0 = ops.lrf_global(pts); o1 = ops.lrf_global(tem)
        for _ in range(4): f()
        torch.cuda.synchronize()
        bad += ((o0 - base0).abs().max().item() > 0) or ((o1 - base1).abs().max().item() > 0)
    print(name, "bad", bad, "/100")
# and the other way round: the load on the side stream BEFORE lrf (same stream), nothing on main
for name, f in loads.items():
    bad = 0
    for it in range(100):
        with torch.cuda.stream(side):
            f(); o0 = ops.lrf_global(pts); o1 = ops.lrf_global(tem)
        torch.cuda.synchronize()
        bad += ((o0 - base0).abs().max().item() > 0) or ((o1 - base1).abs().max().item() > 0)
    print("after", name, "on the same stream: bad", bad, "/100")
