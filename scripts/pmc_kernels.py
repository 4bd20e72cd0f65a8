"""Launch the hand-written kernels that bench.py prices, a few times each, at the bench workload's
shapes -- the target of the rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE collected in separate runs)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unopose_amd import ops  # noqa: E402
from unopose_amd.model import UNOPose, default_model_cfg  # noqa: E402
from unopose_amd.pointnet2 import _ext  # noqa: E402
from unopose_amd.synthetic import make_batch, trained_like_  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.set_grad_enabled(False)
dev = torch.device("cuda")
model = trained_like_(UNOPose(default_model_cfg())).to(dev).eval()
batch, _, _ = make_batch(B, device=dev)
pts = batch["pts"].float()
radius = torch.norm(batch["tem1_pts"] - batch["tem1_pts"].mean(1, keepdim=True), dim=2).max(1)[0]
x = (pts / (radius.reshape(-1, 1, 1) + 1e-6)).contiguous()
pe = model.fine_point_matching.PE
idx = _ext.ball_query(x, x, 0.2, 256)
xt = x.transpose(1, 2).contiguous()
n = 197
gp = torch.cat([torch.ones(B, 1, 3, device=dev), x[:, :n - 1]], 1).contiguous()
for _ in range(3):
    ops.pe_group_mlp_max(x, pe.r2, pe.ns2, pe.mlp2, bf16x3=True)
    _ext.group_points(xt, idx)
    _ext.ball_query(x, x, 0.2, 256)
    E = ops.geo_embedding(gp, model.geo_embedding, out_dtype=torch.bfloat16)
    f = torch.randn(B, n, 256, device=dev)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ops.token_attention(f, f, model.coarse_point_matching.transformers[0].layers[0].attention.attention, 4, E)
# the ViT patch attention at the 518x518 token count (2B images x 1374 tokens x 12 heads)
qkv = torch.randn(2 * B, 1374, 2304, device=dev).bfloat16()
for _ in range(3):
    ops.vit_attention(qkv, 12)
# the fused fc1 + GELU GEMM at the ViT size (M = 2B x 1374 rows, 768 -> 3072)
a = torch.randn(2 * B * 1374, 768, device=dev).bfloat16()
w = (torch.randn(3072, 768, device=dev) / 768 ** 0.5).bfloat16()
bias = torch.randn(3072, device=dev)
for _ in range(3):
    ops.linear_bf16_hip(a, w, bias, True)
# the other three ViT linears on the same kernel (qkv, proj, fc2)
for (K, N) in ((768, 2304), (768, 768), (3072, 768)):
    a2 = torch.randn(2 * B * 1374, K, device=dev).bfloat16()
    w2 = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    b2 = torch.randn(N, device=dev)
    for _ in range(3):
        ops.linear_bf16_hip(a2, w2, b2)
# fine-stage assignment without the similarity matrix (B pairs, 2049 x 2049, 256-wide features)
f1 = torch.randn(B, 2049, 256, device=dev)
f2 = torch.randn(B, 2049, 256, device=dev)
sc = torch.rand(B, 4096, device=dev)
with torch.autocast("cuda", dtype=torch.bfloat16):
    for _ in range(3):
        ops.fine_pose_from_features(f1, f2, 0.1, sc, x, x)
torch.cuda.synchronize()
print("done")
