"""PE kernel timing on the bench's clouds + neighbour-count statistics (how many of the S list entries are real neighbours)."""
import sys, torch
sys.path.insert(0, '.')
from unopose_amd import ops
from unopose_amd.model import UNOPose, default_model_cfg
from unopose_amd.synthetic import make_batch
from unopose_amd.pointnet2 import _ext
torch.set_grad_enabled(False)
B = 32
batch, _, _ = make_batch(B, 2048, 5000, 224, seed=1, device="cuda")
model = UNOPose(default_model_cfg()).cuda().eval()
pe = model.fine_point_matching.PE
pts = batch["pts"].float()
radius = torch.norm(batch["tem1_pts"] - batch["tem1_pts"].mean(1, keepdim=True), dim=2).max(1)[0]
x = (pts / (radius.reshape(-1, 1, 1) + 1e-6)).contiguous()
for r, ns, mlp in ((pe.r1, pe.ns1, pe.mlp1), (pe.r2, pe.ns2, pe.mlp2)):
    d = torch.cdist(x, x)
    cnt = (d < r).sum(-1).float()
    print(f"radius {r} S {ns}: neighbours inside the radius: mean {cnt.mean().item():.1f}, median {cnt.median().item():.0f}, p90 {cnt.quantile(0.9).item():.0f}, max {cnt.max().item():.0f}; tiles needed {((cnt.clamp(max=ns) + 31) // 32).mean().item():.2f} of {ns // 32}")
    for bf in (True, False):
        for _ in range(2): ops.pe_group_mlp_max(x, r, ns, mlp, bf16x3=bf)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): ops.pe_group_mlp_max(x, r, ns, mlp, bf16x3=bf)
        e.record(); torch.cuda.synchronize()
        print(f"   {'bf16x3' if bf else 'fp32  '}: {s.elapsed_time(e) / 5 * 1e3:8.1f} us per {B} clouds")
