import os, sys, torch
sys.path.insert(0, os.getcwd())
from unopose_amd.model import UNOPose, default_model_cfg
from unopose_amd.synthetic import trained_like_, make_batch
from unopose_amd import ops
torch.set_grad_enabled(False)
model = trained_like_(UNOPose(default_model_cfg())).cuda().eval()
ep, _, _ = make_batch(3, S=224, seed=53, device="cuda")
net = model.feature_extraction.rgb_net
x = torch.cat([ep["rgb"], ep["tem1_rgb"]], 0)
def check(name, f, n=30):
    base = f()
    base = [t.float().clone() for t in (base if isinstance(base, (list, tuple)) else [base])]
    worst = 0.0
    for _ in range(n):
        r = f(); r = r if isinstance(r, (list, tuple)) else [r]
        for a, b in zip(r, base):
            worst = max(worst, (a.float() - b).abs().max().item())
    print(f"{name}: max run-to-run diff {worst:.2e}")
with torch.autocast("cuda", dtype=torch.bfloat16):
    check("vit taps (2B,T,3072)", lambda: net.vit(x, taps_side_by_side=True))
    acts = net.vit(x, taps_side_by_side=True)
    check("dense upproj", lambda: ops.linear(acts, net.output_upscaling))
    ch = torch.cat([ep["rgb_choose"], ep["rgb_choose"]], 0)
    plan = ops.upproj_plan(ch, 224, 224, 16, 5, 261)
    check("sparse pixel feats", lambda: ops.sparse_pixel_features(acts, net.output_upscaling, plan))
    blk = net.vit.blocks[0]
    xx = torch.randn(6, 261, 768, device="cuda")
    n1 = ops.add_layernorm(xx, None, blk.norm1, torch.bfloat16)
    check("vit block0 fused", lambda: list(blk.forward_fused(xx.clone(), n1, net.vit.blocks[1].norm1)))
    qkv = ops.linear(n1, blk.attn.qkv)
    check("qkv linear", lambda: ops.linear(n1, blk.attn.qkv))
    check("vit attention", lambda: ops.vit_attention(qkv, 12))
    check("fc1 gelu", lambda: ops.linear(n1, blk.mlp.fc1, gelu=True))
    pts = torch.cat([torch.ones(6, 1, 3, device="cuda"), torch.randn(6, 196, 3, device="cuda")], 1)
    check("geo embedding", lambda: model.geo_embedding(pts))
    check("PE", lambda: model.fine_point_matching.PE(ep["pts"] / 0.3))
with torch.autocast("cuda", dtype=torch.bfloat16):
    c = model.coarse_point_matching
    B = 3
    sf = torch.randn(2 * B, 196, 256, device="cuda")
    check("in_proj", lambda: c.in_proj(sf))
    geo = model.geo_embedding(torch.cat([pts, pts], 0)[:2 * B])
    f = torch.cat([c.bg_token.expand(2 * B, -1, -1).to(torch.bfloat16), c.in_proj(sf).to(torch.bfloat16)], 1)
    f1, f2 = f[:B].contiguous(), f[B:].contiguous()
    g1, g2 = geo[:B], geo[B:]
    blk = c.transformers[0]
    check("coarse block 0", lambda: list(blk(f1, g1, f2, g2)))
    for name, m in blk.named_children():
        print("child", name, type(m).__name__)
