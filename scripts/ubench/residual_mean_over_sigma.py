"""|row mean| / row sigma of the ViT's fp32 residual stream at the input of every LayerNorm the fold replaces (the cancellation term of
`rstd (acc - mean c)`: rounding the UN-normalised rows to bf16 costs a factor ~ (1 + |mean| / sigma) over rounding the normalised ones).
Random-init weights as the tests and the bench use them (tamed 0.1 and untamed 1.0); 8 crops of 518 x 518."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
from oracle.unopose_ref import default_cfg, random_state_dict
from unopose_amd.model import UNOPose, default_model_cfg
from unopose_amd.synthetic import trained_like_
torch.set_grad_enabled(False)
img = torch.randn(8, 3, 518, 518, generator=torch.Generator().manual_seed(3)).cuda()
def report(name, vit):
    p = vit.patch_size
    B = img.shape[0]
    gh = img.shape[2] // p
    patches = img.reshape(B, 3, gh, p, gh, p).permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gh, 3 * p * p)
    x = patches @ vit.patch_embed.proj.weight.reshape(768, -1).T + vit.patch_embed.proj.bias + vit.pos_embed
    x = torch.cat([vit.cls_token.expand(B, -1, -1), vit.reg_token.expand(B, -1, -1), x], 1)
    vals = []
    for blk in vit.blocks:
        for half in (0, 1):
            mu, sd = x.mean(-1), x.std(-1, unbiased=False)
            vals.append((mu.abs() / sd))
            if half == 0:
                a = blk.attn; n = torch.nn.functional.layer_norm(x, (768,), blk.norm1.weight, blk.norm1.bias, 1e-6)
                qkv = (n @ a.qkv.weight.T + a.qkv.bias).reshape(B, -1, 3, a.heads, 64).permute(2, 0, 3, 1, 4)
                o = torch.nn.functional.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2]).transpose(1, 2).reshape(B, -1, 768)
                x = x + blk.ls1.gamma * (o @ a.proj.weight.T + a.proj.bias)
            else:
                n = torch.nn.functional.layer_norm(x, (768,), blk.norm2.weight, blk.norm2.bias, 1e-6)
                h = torch.nn.functional.gelu(n @ blk.mlp.fc1.weight.T + blk.mlp.fc1.bias)
                x = x + blk.ls2.gamma * (h @ blk.mlp.fc2.weight.T + blk.mlp.fc2.bias)
    v = torch.stack([t.flatten() for t in vals])
    print(f"{name}: |mean|/sigma of the residual rows over the 24 LayerNorm inputs: median {v.median().item():.3f}, p99 {v.flatten().quantile(0.99).item() if v.numel() < 1.6e7 else v.flatten()[::8].quantile(0.99).item():.3f}, max {v.max().item():.3f}"
          f" (per LayerNorm input, max: " + " ".join(f"{t.max().item():.2f}" for t in vals) + ")")
for name, tame in (("oracle.random_state_dict tame=0.1 (test fixtures)", 0.1), ("oracle.random_state_dict tame=1.0 (untamed)", 1.0)):
    m = UNOPose(default_model_cfg(feature_extraction=dict(img_size=518)))
    m.load_state_dict(random_state_dict(default_cfg(), seed=0, img_size=518, tame=tame), strict=True)
    report(name, m.cuda().eval().feature_extraction.rgb_net.vit)
m = trained_like_(UNOPose(default_model_cfg(feature_extraction=dict(img_size=518)))).cuda().eval()
report("synthetic.trained_like_ (bench weights)", m.feature_extraction.rgb_net.vit)
