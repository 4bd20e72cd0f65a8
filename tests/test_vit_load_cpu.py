"""CPU: the pretrained-backbone load path (`ViT_AE.load_dinov2`, `interpolate_pos_embed`) against the
reference's own `interpolate_pos_embed` (tests/golden/interpolate_pos_embed.npz, captured from
core/unopose/utils/model_utils.py:105-134) and a synthetic checkpoint in the layout
scripts/download_and_save_dinov2_ckpt.py:21-24 writes ({"model": timm_state_dict}, pos_embed 37x37)."""
import os

import numpy as np
import torch

from unopose_amd.model import default_model_cfg
from unopose_amd.model.modules import ViT_AE, interpolate_pos_embed

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_interpolate_pos_embed_matches_reference_fixture():
    z = np.load(os.path.join(GOLD, "interpolate_pos_embed.npz"))
    out = interpolate_pos_embed(torch.from_numpy(z["src"]), 16)
    assert out.shape == z["out"].shape
    assert np.abs(out.numpy() - z["out"]).max() < 1e-6
    same = interpolate_pos_embed(torch.from_numpy(z["out"]), 16)  # already the target grid: untouched
    assert same.data_ptr() == torch.from_numpy(z["out"]).data_ptr() or torch.equal(same, torch.from_numpy(z["out"]))


def test_load_dinov2_synthetic_checkpoint(tmp_path):
    cfg = default_model_cfg().feature_extraction
    net = ViT_AE(cfg)
    g = torch.Generator().manual_seed(0)
    ck = {k: torch.randn(v.shape, generator=g) for k, v in net.vit.state_dict().items()}
    ck["pos_embed"] = torch.randn(1, 37 * 37, 768, generator=g)  # the released DINOv2 grid (518 / 14 = 37)
    ck["head.weight"] = torch.randn(5, 768, generator=g)  # a checkpoint head of another width must not block the load
    ck["head.bias"] = torch.randn(5, generator=g)
    path = str(tmp_path / "timm_vit_base_patch14_reg4_dinov2_lvd142m.pth")
    torch.save({"model": ck}, path)
    head_before = net.vit.head.weight.detach().clone()
    net.load_dinov2(path)
    sd = net.vit.state_dict()
    for k in ("cls_token", "reg_token", "patch_embed.proj.weight", "blocks.0.attn.qkv.weight", "blocks.11.mlp.fc2.bias",
              "blocks.5.ls1.gamma", "norm.weight"):
        assert torch.equal(sd[k], ck[k]), k
    assert sd["pos_embed"].shape == (1, 256, 768)
    assert torch.allclose(sd["pos_embed"], interpolate_pos_embed(ck["pos_embed"], 16))
    assert torch.equal(net.vit.head.weight, head_before)  # mismatched head skipped, as F:181-192 does
    # the cfg route (`pretrained=True, vit_ckpt=path`) goes through the same loader, at the 518 grid: no resampling
    cfg518 = default_model_cfg(feature_extraction=dict(img_size=518, pretrained=True, vit_ckpt=path)).feature_extraction
    net518 = ViT_AE(cfg518)
    assert torch.equal(net518.vit.state_dict()["pos_embed"], ck["pos_embed"])
