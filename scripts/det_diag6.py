import os, sys, torch
sys.path.insert(0, os.getcwd())
from unopose_amd import ops
from unopose_amd.synthetic import make_batch
torch.set_grad_enabled(False)
ep, _, _ = make_batch(3, S=224, seed=53, device="cuda")
pts, tem = ep["pts"], ep["tem1_pts"]
base0, base1 = ops.lrf_global(pts).clone(), ops.lrf_global(tem).clone()
side = torch.cuda.Stream()
a = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
bad = 0
for it in range(200):
    for _ in range(3): a @ a  # heavy main-stream work
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        o0 = ops.lrf_global(pts); o1 = ops.lrf_global(tem)
    x = torch.randn(4096, 4096, device="cuda") @ torch.randn(4096, 4096, device="cuda")
    torch.cuda.synchronize()
    for name, o, b in (("pts", o0, base0), ("tem", o1, base1)):
        d = (o - b).abs()
        if d.max().item() > 0:
            bad += 1
            per_cloud = (d.amax(dim=(1, 2))).tolist()
            nbad = (d > 0).sum().item()
            print(f"it {it} {name}: max {d.max().item():.2e}, elements differing {nbad}/{d.numel()}, per cloud max {['%.1e' % v for v in per_cloud]}")
print("bad", bad)
