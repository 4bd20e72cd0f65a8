"""Building blocks of the MI355X UNOPose model.

Parameter containers mirror the reference's module tree so ``state_dict`` keys are
identical; the arithmetic is re-expressed around fused HIP
operators (``unopose_amd.ops``) where the reference uses op-by-op PyTorch.
References: T = core/unopose/model/transformer.py, F = .../oneref_feature_extraction.py,
Fi = .../oneref_predator_fine_point_matching.py, U = core/unopose/utils/model_utils.py.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


# ------------------------------------------------------------------ ViT -----
class _Attention(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.heads = heads
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, T, C = x.shape
        qkv = ops.linear(x, self.qkv)
        o = ops.vit_attention(qkv, self.heads)  # (B,T,C): softmax(q k^T / sqrt(hd)) v per head
        return ops.linear(o, self.proj)


class _LayerScale(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.gamma = nn.Parameter(torch.full((dim,), 1e-5))


class _Mlp(nn.Module):
    def __init__(self, dim, ratio=4):
        super().__init__()
        self.fc1 = nn.Linear(dim, ratio * dim)
        self.fc2 = nn.Linear(ratio * dim, dim)

    def forward(self, x):
        return ops.mlp(x, self.fc1, self.fc2)


class _Block(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attention(dim, heads)
        self.ls1 = _LayerScale(dim)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim)
        self.ls2 = _LayerScale(dim)

    def forward_fused(self, x, n1, next_norm):
        """autocast path with the residual stream read once per branch: `n1` = norm1(x) (bf16) is handed
        in by the previous block, and this block returns norm(x_out) for the next consumer (`next_norm`:
        the next block's norm1, or None at the end)."""
        y = self.attn(n1)
        n2 = ops.scale_residual_layernorm_(x, y, self.ls1.gamma, self.norm2)
        y = self.mlp(n2)
        if next_norm is None:
            ops.scale_residual_(x, y, self.ls2.gamma)
            return x, None
        return x, ops.scale_residual_layernorm_(x, y, self.ls2.gamma, next_norm)

    def forward_folded(self, x, n1, xb, stats):
        """autocast path with the residual + LayerNorm passes folded into the GEMMs (ops.linear_residual_ / ops.linear_lnfold): the block
        reads either `n1` = norm1(x) in bf16 (the first block, from the ViT prologue) or the previous block's un-normalised bf16 rows `xb`
        with their row partial sums, updates the fp32 residual stream `x` in place twice and returns (xb, stats) of its output."""
        a = self.attn
        qkv = ops.linear(n1, a.qkv) if n1 is not None else ops.linear_lnfold(xb, stats, a.qkv, self.norm1)
        xb, stats = ops.linear_residual_(x, ops.vit_attention(qkv, a.heads), a.proj, self.ls1.gamma)
        h = ops.linear_lnfold(xb, stats, self.mlp.fc1, self.norm2, gelu=True)
        return ops.linear_residual_(x, h, self.mlp.fc2, self.ls2.gamma)

    def forward_fused_f32(self, x, n1s, next_norm):
        """The same without autocast (the reference's default precision): `n1s` = norm1(x) in the split layout of csrc/gemm_f32.hip;
        qkv, fc1 and fc2 read split operands directly (fc1 hands its GELU output to fc2 in that layout too), the attention core and
        the projection run on fp32 data, and each LayerScale residual also emits the next LayerNorm in split form -- one pass over
        the residual stream per branch instead of multiply + add + LayerNorm + split."""
        B, T, C = x.shape
        rows = B * T
        a, m = self.attn, self.mlp
        cq = ops._f32x3_weights(a.qkv)
        # qkv, the attention core and the projection chained on split operands: no fp32 qkv / attention tensor exists
        qkv_s = ops.linear_f32x3(n1s, cq[1], cq[2], rows, 3 * C, C, out="split")
        cp = ops._f32x3_weights(a.proj)
        y = ops.linear_f32x3(ops.vit_attention_f32_ss(qkv_s, B, T, a.heads), cp[1], cp[2], rows, C, C).reshape(B, T, C)
        n2s = ops.scale_residual_layernorm_f32_(x, y, self.ls1.gamma, self.norm2)
        c1, c2 = ops._f32x3_weights(m.fc1), ops._f32x3_weights(m.fc2)
        hs = ops.linear_f32x3(n2s, c1[1], c1[2], rows, 4 * C, C, gelu=True, out="split")
        y = ops.linear_f32x3(hs, c2[1], c2[2], rows, C, 4 * C).reshape(B, T, C)
        if next_norm is None:
            return ops.scale_residual_layernorm_f32_(x, y, self.ls2.gamma, None), None
        return x, ops.scale_residual_layernorm_f32_(x, y, self.ls2.gamma, next_norm)

    def forward(self, x):
        if torch.is_autocast_enabled() and x.dtype == torch.float32 and x.is_cuda and not ops.is_differentiable():
            # fused glue (csrc/fused.hip): LayerNorm -> bf16 in one pass, LayerScale residual in one pass;
            # the residual stream itself stays fp32 exactly as under the reference's autocast
            # (x is updated IN PLACE: ViT.forward owns the residual stream and rebinds it every block)
            x = x.contiguous()
            y = self.attn(ops.add_layernorm(x, None, self.norm1, torch.bfloat16))
            ops.scale_residual_(x, y, self.ls1.gamma)
            y = self.mlp(ops.add_layernorm(x, None, self.norm2, torch.bfloat16))
            return ops.scale_residual_(x, y, self.ls2.gamma)
        x = x + self.attn(self.norm1(x)) * self.ls1.gamma
        return x + self.mlp(self.norm2(x)) * self.ls2.gamma


class _PatchEmbed(nn.Module):
    def __init__(self, dim, patch):
        super().__init__()
        self.proj = nn.Conv2d(3, dim, patch, patch)


SPARSE_SINGLE_CROP = True  # (module attribute for A/Bs) ViT_AE.pixel_features of one crop batch through the sparse up-projection; False: dense GEMM + fp32 features
F32_PROLOGUE = True  # (module attribute for A/Bs) the no-autocast ViT's patch unfolding / token assembly / first LayerNorm on the fused fp32 kernels; False: the op-by-op front end
F32_TAPS_SPLIT = True  # (module attribute for A/Bs) the no-autocast ViT's tap LayerNorms written side by side in the split layout; False: torch LayerNorm + cat + split pass


class SplitTaps:
    """The four tap LayerNorms of the no-autocast ViT side by side as ONE (B T, 2 K) bf16 matrix in the split layout of csrc/gemm_f32.hip
    (K = 4 D fp32-equivalent columns; the 5 prefix tokens of every image stay in place)."""

    def __init__(self, split, B, T, K):
        self.split, self.B, self.T, self.K = split, B, T, K


class ViT(nn.Module):
    """DINOv2 ViT (reg4, no_embed_class) with timm 0.9.12's state_dict keys; forward returns the four
    tapped, final-normed token maps like the reference subclass (F:24-42).  Accepts any S % 14 == 0 as
    long as ``pos_embed`` has (S/14)^2 rows (the reference is fixed to 224: F:62-63)."""

    def __init__(self, embed_dim=768, depth=12, num_heads=12, patch_size=14, img_size=224, reg_tokens=4):
        super().__init__()
        self.patch_size, self.depth = patch_size, depth
        self.patch_embed = _PatchEmbed(embed_dim, patch_size)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.reg_token = nn.Parameter(torch.zeros(1, reg_tokens, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, (img_size // patch_size) ** 2, embed_dim))
        self.blocks = nn.ModuleList([_Block(embed_dim, num_heads) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        self.head = nn.Linear(embed_dim, 1000)  # unused; kept so released checkpoints load strictly

    def forward(self, x, taps_side_by_side=False):
        """-> the four tapped LayerNorm outputs (F:24-42).  With `taps_side_by_side` on the fused autocast path
        they are returned as ONE (B, T, 4*D) tensor (tap k in columns [k*D, (k+1)*D)), written in place by the
        four LayerNorm launches instead of being concatenated afterwards.  `x` may be a PAIR of image batches
        (query crops, reference crops): they run as one batch without being concatenated first."""
        xa, xb = x if isinstance(x, (tuple, list)) else (x, None)
        if ops.vit_prologue_ok(xa, self) and (xb is None or xb.shape[1:] == xa.shape[1:]):
            x, n1 = ops.vit_prologue(xa, xb, self, self.blocks[0].norm1)
            return self._fused_blocks(x, n1, taps_side_by_side)
        if F32_PROLOGUE and ops.vit_prologue_f32_ok(xa, self) and (xb is None or xb.shape[1:] == xa.shape[1:]):
            x, ns = ops.vit_prologue_f32(xa, xb, self, self.blocks[0].norm1)  # (round 6: the fp32 twin of the fused prologue)
            return self._fused_blocks_f32(x, ns, taps_side_by_side)
        x = xa if xb is None else torch.cat([xa, xb], 0)
        B = x.shape[0]
        p = self.patch_size
        # patch conv 14x14/14 as one GEMM: (B, P, 3*14*14) @ W^T
        gh, gw = x.shape[2] // p, x.shape[3] // p
        patches = x.reshape(B, 3, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, 3 * p * p)
        x = ops.patch_embed(patches, self.patch_embed.proj)
        x = x + self.pos_embed
        x = torch.cat([self.cls_token.expand(B, -1, -1).to(x.dtype), self.reg_token.expand(B, -1, -1).to(x.dtype), x], 1)
        n = self.depth // 4
        taps = {self.depth - 1, self.depth - n - 1, self.depth - 2 * n - 1, self.depth - 3 * n - 1}
        outs = []
        if x.is_cuda and torch.is_autocast_enabled() and x.dtype == torch.float32 and x.shape[-1] % 4 == 0 \
                and not ops.is_differentiable():
            # fused glue: each block's LayerScale residual also produces the next LayerNorm (csrc/fused.hip)
            x = x.contiguous()
            return self._fused_blocks(x, ops.add_layernorm(x, None, self.blocks[0].norm1, torch.bfloat16), taps_side_by_side)
        if ops.vit_f32_fused_ok(x, self):
            x = x.contiguous()
            return self._fused_blocks_f32(x, ops.scale_residual_layernorm_f32_(x, None, None, self.blocks[0].norm1), taps_side_by_side)
        for i, blk in enumerate(self.blocks):
            x = blk(x)
            if i in taps:
                outs.append(ops.add_layernorm(x, None, self.norm)
                            if x.is_cuda and torch.is_autocast_enabled() and not ops.is_differentiable() else self.norm(x))
        return outs

    def _fused_blocks_f32(self, x, ns, taps_side_by_side):
        """The blocks of the no-autocast forward on the fp32-class GEMMs: x (B,T,D) fp32 residual stream (updated in place), ns = norm1(x) of
        block 0 in the split layout."""
        n = self.depth // 4
        taps = {self.depth - 1, self.depth - n - 1, self.depth - 2 * n - 1, self.depth - 3 * n - 1}
        outs = []
        # taps side by side (round 6): the four tap LayerNorms go straight into the column blocks of ONE split-layout matrix, the K = 4 D operand
        # of the up-projection -- no torch LayerNorm, no 1 GB concatenation, no split pass (4.3 GB of traffic per forward at B = 32, 518 x 518)
        wide = torch.empty(x.shape[0] * x.shape[1], 2 * 4 * x.shape[2], dtype=torch.bfloat16, device=x.device) if taps_side_by_side and F32_TAPS_SPLIT else None
        for i, blk in enumerate(self.blocks):
            x, ns = blk.forward_fused_f32(x, ns, self.blocks[i + 1].norm1 if i + 1 < len(self.blocks) else None)
            if i in taps:
                if wide is not None:
                    ops.scale_residual_layernorm_f32_(x, None, None, self.norm, wide=wide, block=len(outs))
                    outs.append(None)
                else:
                    outs.append(self.norm(x))
        return SplitTaps(wide, x.shape[0], x.shape[1], 4 * x.shape[2]) if wide is not None else outs


def _vit_fused_blocks(self, x, n1, taps_side_by_side):
    """The 12 blocks on the fused autocast path: x = fp32 residual stream (updated in place), n1 = norm1(x) of block 0 as bf16;
    each block's LayerScale residual also produces the next LayerNorm (csrc/fused.hip)."""
    n = self.depth // 4
    taps = {self.depth - 1, self.depth - n - 1, self.depth - 2 * n - 1, self.depth - 3 * n - 1}
    outs = []
    B, _, D = x.shape
    wide = torch.empty(B, x.shape[1], len(taps) * D, dtype=torch.bfloat16, device=x.device) if taps_side_by_side else None
    fold = ops.ln_fold_ok(B * x.shape[1], D)
    xb = stats = None
    for i, blk in enumerate(self.blocks):
        nxt = self.blocks[i + 1].norm1 if i + 1 < len(self.blocks) else None
        if fold:  # the LayerNorms live in the GEMM epilogues: the residual stream is only touched by proj / fc2
            xb, stats = blk.forward_folded(x, n1, xb, stats)
            n1 = None
        else:
            x, n1 = blk.forward_fused(x, n1, nxt)
        if i in taps:
            k = len(outs)
            outs.append(ops.add_layernorm(x, None, self.norm, out=None if wide is None else wide[:, :, k * D:(k + 1) * D]))
    return outs if wide is None else wide


ViT._fused_blocks = _vit_fused_blocks


def interpolate_pos_embed(pos_embed_ckpt, new_side):
    """Non-antialiased bicubic resampling of a square patch pos-embed (U:105-134). (1,P0,C)->(1,P1,C)."""
    P0, C = pos_embed_ckpt.shape[-2:]
    s0 = int(P0 ** 0.5)
    if s0 == new_side:
        return pos_embed_ckpt
    t = pos_embed_ckpt.reshape(-1, s0, s0, C).permute(0, 3, 1, 2)
    t = F.interpolate(t, size=(new_side, new_side), mode="bicubic", align_corners=False)
    return t.permute(0, 2, 3, 1).flatten(1, 2)


class ViT_AE(nn.Module):
    """F:45-236 (``up_type="linear"``, pyramid features).  ``pixel_features`` never materialises the
    (B,256,S,S) map: bilinear interpolation commutes with the pixel gather."""

    def __init__(self, cfg):
        super().__init__()
        assert cfg.up_type == "linear" and cfg.use_pyramid_feat, "only the configured variant is built"
        assert "reg4" in cfg.vit_type and "patch14" in cfg.vit_type
        dims = {"vit_small": (384, 12, 6), "vit_base": (768, 12, 12), "vit_large": (1024, 24, 16)}
        D, depth, heads = dims[cfg.vit_type[:cfg.vit_type.index("_patch")]]
        assert D == cfg.embed_dim
        self.out_dim = cfg.out_dim
        self.img_size = cfg.get("img_size", 224)
        self.vit = ViT(D, depth, heads, 14, self.img_size)
        self.output_upscaling = nn.Linear(4 * D, 16 * cfg.out_dim, bias=True)
        if cfg.get("pretrained", False):
            self.load_dinov2(cfg.vit_ckpt)

    def load_dinov2(self, path):
        """F:173-192: checkpoint = {"model": timm_state_dict}; pos_embed resampled as the reference does."""
        ck = torch.load(path, map_location="cpu")["model"]
        sd = self.vit.state_dict()
        for k in ("head.weight", "head.bias"):
            if k in ck and ck[k].shape != sd[k].shape:
                del ck[k]
        if "pos_embed" in ck:
            ck["pos_embed"] = interpolate_pos_embed(ck["pos_embed"], self.img_size // 14)
        self.vit.load_state_dict(ck, strict=False)

    def lowres_map(self, x):
        """(B,3,S,S) -> channels-last low-res feature map (B, 4S/14, 4S/14, 256)."""
        B, _, H, W = x.shape
        side = H // 14
        outs = self.vit(x)
        z = torch.cat([o[:, 5:, :] for o in outs], dim=2)
        z = ops.linear(z, self.output_upscaling).reshape(B, side, side, 4, 4, self.out_dim)
        return z.permute(0, 1, 3, 2, 4, 5).reshape(B, 4 * side, 4 * side, self.out_dim), (H, W)

    def upprojected(self, x):
        """(B,3,S,S) -> up-projection output in its native order (B, side, side, 4, 4, 256)."""
        B, _, H, W = x.shape
        side = H // 14
        outs = self.vit(x)
        z = torch.cat([o[:, 5:, :] for o in outs], dim=2)
        return ops.linear(z, self.output_upscaling).reshape(B, side, side, 4, 4, self.out_dim), (H, W)

    def upprojected_tokens(self, x):
        """(B,3,S,S) -> (z (B, tokens, 4, 4, 256), (H, W), tok_offset): the up-projection applied to the token
        tensor as the ViT leaves it.  On the fused autocast path the four taps were written side by side and
        the 5 class / register tokens stay in place (tok_offset = 5: the pixel sampler skips them by index
        math), so neither the tap concatenation nor the prefix slice (F:213) costs a copy."""
        B, _, H, W = x.shape
        return self.upproject(self.vit(x, taps_side_by_side=True), H, W)

    def upproject(self, outs, H, W):
        """The second half of `upprojected_tokens`, on what `self.vit(x, taps_side_by_side=True)` returned."""
        if isinstance(outs, SplitTaps):  # fp32: the taps as one split-layout operand (prefix tokens in place, as on the autocast path)
            lin = self.output_upscaling
            c = ops._f32x3_weights(lin)
            z = ops.linear_f32x3(outs.split, c[1], c[2], outs.B * outs.T, lin.weight.shape[0], outs.K)
            return z.reshape(outs.B, outs.T, 4, 4, self.out_dim), (H, W), outs.T - (H // 14) * (W // 14)
        B = outs.shape[0] if torch.is_tensor(outs) else outs[0].shape[0]
        if torch.is_tensor(outs):
            z = ops.linear(outs, self.output_upscaling)
            return z.reshape(B, outs.shape[1], 4, 4, self.out_dim), (H, W), outs.shape[1] - (H // 14) * (W // 14)
        z = ops.linear(torch.cat([o[:, 5:, :] for o in outs], dim=2), self.output_upscaling)
        return z.reshape(B, -1, 4, 4, self.out_dim), (H, W), 0

    def pixel_features(self, x, choose):
        if x.is_cuda and self.out_dim == 256 and not ops.is_differentiable():
            if SPARSE_SINGLE_CROP and ops.sparse_upproj_ok(x) and x.shape[-1] == x.shape[-2] and x.shape[-1] % 14 == 0:
                # one crop batch alone (a cached reference on the other side, or the reference being encoded): the sparse up-projection
                # as in the two-crop forward -- only the cells the chosen pixels read, bf16 features (round 6; was the dense 3072 -> 4096 GEMM)
                sd = x.shape[-1] // 14
                npre = self.vit.cls_token.shape[1] + self.vit.reg_token.shape[1]
                plan = ops.upproj_plan(choose, x.shape[-2], x.shape[-1], sd, npre, npre + sd * sd)
                acts = self.vit(x, taps_side_by_side=True)
                if torch.is_tensor(acts) and acts.shape[1] == plan["tok_stride"]:
                    return ops.sparse_pixel_features(acts, self.output_upscaling, plan)
                z, (H, W), off = self.upproject(acts, x.shape[-2], x.shape[-1])
                return ops.bilinear_sample_native(z, choose, H, W, tok_offset=off)
            z, (H, W), off = self.upprojected_tokens(x)
            return ops.bilinear_sample_native(z, choose, H, W, tok_offset=off)
        low, (H, W) = self.lowres_map(x)
        return ops.bilinear_sample_pixels(low, choose, H, W)


class ViTEncoderOneRef(nn.Module):
    def __init__(self, cfg, npoint=None):
        super().__init__()
        self.npoint = npoint
        self.rgb_net = ViT_AE(cfg)


# ------------------------------------------------------- transformer --------
class SinusoidalPositionalEmbedding(nn.Module):
    def __init__(self, d_model):
        super().__init__()
        self.d_model = d_model
        div = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
        self.register_buffer("div_term", div)


class GeometricStructureEmbedding(nn.Module):
    """T:287-350."""

    def __init__(self, cfg):
        super().__init__()
        self.sigma_d, self.sigma_a, self.angle_k = cfg.sigma_d, cfg.sigma_a, cfg.angle_k
        self.factor_a = 180.0 / (self.sigma_a * math.pi)
        self.embedding = SinusoidalPositionalEmbedding(cfg.hidden_dim)
        self.proj_d = nn.Linear(cfg.hidden_dim, cfg.hidden_dim)
        self.proj_a = nn.Linear(cfg.hidden_dim, cfg.hidden_dim)
        self.reduction_a = cfg.reduction_a
        if self.reduction_a not in ("max", "mean"):
            raise ValueError(f"Unsupported reduction mode: {self.reduction_a}.")

    def forward(self, points):
        return ops.geo_embedding(points, self)


class _MHA(nn.Module):
    def __init__(self, d, rpe):
        super().__init__()
        self.proj_q = nn.Linear(d, d)
        self.proj_k = nn.Linear(d, d)
        self.proj_v = nn.Linear(d, d)
        if rpe:
            self.proj_p = nn.Linear(d, d)


class _AttentionLayer(nn.Module):
    def __init__(self, d, rpe):
        super().__init__()
        self.attention = _MHA(d, rpe)
        self.linear = nn.Linear(d, d)
        self.norm = nn.LayerNorm(d)


class _AttentionOutput(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.expand = nn.Linear(d, 2 * d)
        self.squeeze = nn.Linear(2 * d, d)
        self.norm = nn.LayerNorm(d)

    def forward(self, x, out=None):
        if x.is_cuda and not ops.is_differentiable():  # squeeze + residual + LayerNorm: one launch under autocast
            return ops.ffn_add_layernorm(x, self.expand, self.squeeze, self.norm, out=out)
        assert out is None
        return self.norm(x + ops.linear(ops.linear(x, self.expand, relu=True), self.squeeze))


class TransformerLayer(nn.Module):
    """T:196-227 (cross, ``rpe=False``) and T:444-466 (RPE self-attention, ``rpe=True``); 4 heads."""

    def __init__(self, d, rpe, heads=4):
        super().__init__()
        self.heads = heads
        self.attention = _AttentionLayer(d, rpe)
        self.output = _AttentionOutput(d)

    def forward(self, x, mem, embed=None, out=None):
        """`out` (inference on the GPU only): a contiguous tensor of x's shape the layer's result is written into."""
        a = self.attention
        if mem is None:  # self-attention
            mem = x
        h = ops.token_attention(x, mem, a.attention, self.heads, embed)
        if x.is_cuda and not ops.is_differentiable():
            x = ops.linear_add_layernorm(h, a.linear, x, a.norm)
        else:
            x = a.norm(ops.linear(h, a.linear) + x)
        return self.output(x, out=out)


class GeometricTransformer(nn.Module):
    """T:469-514 with blocks ["self","cross"], sequential cross order (T:507-508)."""

    def __init__(self, d, heads=4):
        super().__init__()
        self.layers = nn.ModuleList([TransformerLayer(d, True, heads), TransformerLayer(d, False, heads)])

    def forward(self, f0, e0, f1, e1):
        e_all = _adjacent(e0, e1)
        if e_all is not None and f0.shape == f1.shape:
            # the self layer shares its weights between the clouds (T:498-499): run both as ONE batch of 2B
            # (the embeddings of the two clouds were produced back to back in one buffer -- and so were the features when the
            # previous block's cross layers wrote them, see forward_stacked)
            f = _adjacent(f0, f1)
            f = self.forward_stacked(torch.cat([f0, f1], 0) if f is None else f, e_all)
            B = f0.shape[0]
            return f[:B], f[B:]
        f0 = self.layers[0](f0, f0, e0)
        f1 = self.layers[0](f1, f1, e1)
        f0 = self.layers[1](f0, f1)
        f1 = self.layers[1](f1, f0)
        return f0, f1

    def forward_stacked(self, f, e_all):
        """Both clouds as one (2B, n, C) batch (cloud 0 = first half) with their embeddings stacked the same way; returns the
        block's output stacked the same way: on the GPU the two cross layers write the halves of ONE tensor (no concatenation
        before the next block's 2B self layer or the sparse-to-dense block's dense layer)."""
        B = f.shape[0] // 2
        f = self.layers[0](f, None, e_all)
        if ops.USE_STACKED_OUT and f.is_cuda and not ops.is_differentiable():
            out = torch.empty_like(f, memory_format=torch.contiguous_format)
            f0 = self.layers[1](f[:B], f[B:], out=out[:B])
            self.layers[1](f[B:], f0, out=out[B:])
            return out
        f0 = self.layers[1](f[:B], f[B:])
        return torch.cat([f0, self.layers[1](f[B:], f0)], 0)


def _adjacent(e0, e1):
    """(2B,...) view over e0 | e1 when they are contiguous halves of one allocation, else None."""
    if (e0.shape != e1.shape or e0.dtype != e1.dtype or not e0.is_contiguous() or not e1.is_contiguous()
            or e0.untyped_storage().data_ptr() != e1.untyped_storage().data_ptr()
            or e1.storage_offset() != e0.storage_offset() + e0.numel()):
        return None
    return e0.as_strided((2 * e0.shape[0],) + tuple(e0.shape[1:]), e0.stride())


class _LinearAttention(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.proj_q = nn.Linear(d, d)
        self.proj_k = nn.Linear(d, d)
        self.proj_v = nn.Linear(d, d)
        self.scale = nn.Parameter(torch.zeros(1, 1, d))


class _LinearAttentionLayer(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.attention = _LinearAttention(d)
        self.linear = nn.Linear(d, d)
        self.norm = nn.LayerNorm(d)


class LinearTransformerLayer(nn.Module):
    """T:517-612 focused linear attention, dense <- sparse."""

    def __init__(self, d, heads=4, focusing_factor=3):
        super().__init__()
        self.heads, self.focusing_factor = heads, focusing_factor
        self.attention = _LinearAttentionLayer(d)
        self.output = _AttentionOutput(d)

    def forward(self, x, mem, kv_skip=0):
        """`kv_skip`: the first rows of every batch of `mem` that are NOT keys / values (the background-token row of a stacked
        (2B, 1 + n, C) block output, read in place instead of sliced into a copy)."""
        a = self.attention
        h = ops.focused_linear_attention(x, mem, a.attention, self.heads, self.focusing_factor, kv_skip=kv_skip)
        if x.is_cuda and not ops.is_differentiable():
            x = ops.linear_add_layernorm(h, a.linear, x, a.norm)
        else:
            x = a.norm(ops.linear(h, a.linear) + x)
        return self.output(x)


class SparseToDenseTransformer(nn.Module):
    """T:615-671 (with_bg_token, replace_bg_token)."""

    def __init__(self, d, heads=4, focusing_factor=3):
        super().__init__()
        self.sparse_layer = GeometricTransformer(d, heads)
        self.dense_layer = LinearTransformerLayer(d, heads, focusing_factor)

    @staticmethod
    def _sample(dense, fps_idx):
        # T:655-662: rows are gathered from the tensor that STILL holds the bg token at row 0 with
        # indices of the bg-free cloud (off-by-one preserved for parity with trained weights, App-E.2)
        return torch.cat([dense[:, 0:1], ops.gather_rows(dense, fps_idx)], dim=1)

    def forward(self, d0, e0, i0, d1, e1, i1):
        f0, f1 = self.sparse_layer(self._sample(d0, i0), e0, self._sample(d1, i1), e1)
        if d0.shape == d1.shape and f0.shape == f1.shape:  # weight-shared dense layer: both clouds as 2B
            B = d0.shape[0]
            nn_ = self.dense_layer(torch.cat([d0[:, 1:], d1[:, 1:]], 0), torch.cat([f0[:, 1:], f1[:, 1:]], 0))
            n0, n1 = nn_[:B], nn_[B:]
        else:
            n0 = self.dense_layer(d0[:, 1:], f0[:, 1:])
            n1 = self.dense_layer(d1[:, 1:], f1[:, 1:])
        return torch.cat([f0[:, 0:1], n0], 1), torch.cat([f1[:, 0:1], n1], 1)

    def forward_stacked(self, dense, bg, e_all, idx_all):
        """Same block with the two clouds stacked as one batch of 2B and the background token kept BESIDE
        the dense features -- dense (2B,N,C) contiguous, bg (2B,1,C) -- so that no (B,N+1,C) tensor is
        re-assembled per block (the reference's layout costs two 134 MB concatenations per block at B=32).
        Returns the new (dense, bg)."""
        # T:655-662 with App-E.2's off-by-one: index i addresses row i of [bg | dense], i.e. bg for i == 0
        f = self.sparse_layer.forward_stacked(ops.gather_rows(dense, idx_all, off=1, alt=bg, prepend=True), e_all)
        if not ops.USE_STACKED_OUT:  # round 5's form (A/B): keys / values and background rows copied out of the block output
            return self.dense_layer(dense, f[:, 1:].contiguous()), f[:, 0:1].contiguous()
        return self.dense_layer(dense, f, kv_skip=1), f[:, 0:1]  # (the background rows stay where they are: a strided view)


# -------------------------------------------------------------- PE ----------
class _BN(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.bn = nn.BatchNorm2d(c)


class _ConvBN(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 1, bias=False)
        self.normlayer = _BN(cout)

    def folded(self):
        """Eval-mode BN folded into the 1x1 conv: y = relu(W' x + b')."""
        bn = self.normlayer.bn
        s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        return self.conv.weight.reshape(self.conv.weight.shape[0], -1) * s[:, None], bn.bias - bn.running_mean * s


class _SharedMLP(nn.Module):
    def __init__(self, chans):
        super().__init__()
        for i in range(len(chans) - 1):
            self.add_module(f"layer{i}", _ConvBN(chans[i], chans[i + 1]))
        self.n = len(chans) - 1

    def layers(self):
        return [getattr(self, f"layer{i}") for i in range(self.n)]

    def forward(self, x):
        """(B,C,N,S) through the unfolded layers: 1x1 Conv2d -> BatchNorm2d (train or eval statistics per module mode) ->
        ReLU.  The inference path never calls this (csrc/pe.hip runs the BN-folded chain); training does."""
        for l in self.layers():
            # train mode on the GPU: csrc/conv_train.hip (1 x 1 convolution, fwd + both gradients) and csrc/bn_train.hip (batch statistics + ReLU, fwd + bwd)
            x = ops.bn_relu(ops.conv1x1(x, l.conv), l.normlayer.bn)
        return x

    def forward_pooled(self, x):
        """`forward(x).max(dim=3)[0]` with the last layer's BatchNorm + ReLU fused into the pooling (training; csrc/bn_train.hip)."""
        ls = self.layers()
        for l in ls[:-1]:
            x = ops.bn_relu(ops.conv1x1(x, l.conv), l.normlayer.bn)
        return ops.bn_relu_maxpool(ops.conv1x1(x, ls[-1].conv), ls[-1].normlayer.bn)


class _Conv1d(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv1d(cin, cout, 1, bias=True)


class PositionalEncoding(nn.Module):
    """Fi:138-178: two-scale QueryAndLRFGroup -> SharedMLP[6,32,64,128] -> max over neighbours -> Conv1d."""

    def __init__(self, out_dim, r1, r2, nsample1, nsample2):
        super().__init__()
        self.r1, self.r2, self.ns1, self.ns2 = r1, r2, nsample1, nsample2
        self.mlp1 = _SharedMLP([6, 32, 64, 128])
        self.mlp2 = _SharedMLP([6, 32, 64, 128])
        self.mlp3 = _Conv1d(256, out_dim)

    def groups(self, pts):
        """Both scales' grouped features, max-pooled: (B,N,3) -> (B,N,256) fp32 = [scale 1 | scale 2].  Only the two
        fused HIP launches (no library GEMM, no inter-workgroup waits), so it may run on a side stream."""
        pts = pts.float()
        if ops.is_differentiable():
            # training: grouped features from the fused grouping kernel (constants w.r.t. the weights, as in the
            # reference, whose `_ext` outputs carry no gradient), then the REAL Conv2d / BatchNorm2d modules -- batch
            # statistics and running-stat updates in train mode (pytorch_utils.py:25-132) -- recorded by autograd
            with torch.autocast("cuda", enabled=False):
                f1 = self.mlp1.forward_pooled(ops.query_lrf_group(pts, self.r1, self.ns1))
                f2 = self.mlp2.forward_pooled(ops.query_lrf_group(pts, self.r2, self.ns2))
            return torch.cat([f1, f2], dim=1).transpose(1, 2)
        if pts.is_cuda and (torch.is_autocast_enabled() or ops.USE_F32X3) and self.r2 >= self.r1 and self.ns1 % 32 == 0 and self.ns2 % 32 == 0:
            # the wide scale first: its neighbour lists are the candidates of the narrow scale (csrc/pe.hip)
            f2, cand = ops.pe_group_mlp_max(pts, self.r2, self.ns2, self.mlp2, want_cand=True)
            f1 = ops.pe_group_mlp_max(pts, self.r1, self.ns1, self.mlp1, cand_in=cand)
        else:
            f1 = ops.pe_group_mlp_max(pts, self.r1, self.ns1, self.mlp1)  # (B,N,128)
            f2 = ops.pe_group_mlp_max(pts, self.r2, self.ns2, self.mlp2)
        return torch.cat([f1, f2], dim=2).float()

    def split_ok(self, pts):
        """Can the two scales be written straight into the split-layout operand of mlp3 (csrc/pe.hip -> csrc/gemm_f32.hip)?"""
        return (pts.is_cuda and torch.is_autocast_enabled() and not ops.is_differentiable() and ops.USE_F32X3 and self.r2 >= self.r1
                and self.ns1 % 32 == 0 and self.ns2 % 32 == 0 and self.mlp3.conv.weight.shape[0] % 256 == 0)

    def groups_split(self, pts, buf, b0):
        """`groups` of the clouds `pts` written into rows b0.. of `buf` ((Btot,N,512) bf16 = split layout of the (Btot,N,256) fp32
        features): no concatenation of the scales, no fp32 round trip; two launches, no library kernel (side-stream safe)."""
        pts = pts.float()
        _, cand = ops.pe_group_mlp_max(pts, self.r2, self.ns2, self.mlp2, want_cand=True, out_split=(buf, b0, 128))
        ops.pe_group_mlp_max(pts, self.r1, self.ns1, self.mlp1, cand_in=cand, out_split=(buf, b0, 0))
        return buf

    def _mlp3_split(self):
        conv = self.mlp3.conv
        key = (conv.weight._version, conv.weight.data_ptr(), conv.bias._version)
        cache = getattr(conv, "_f32x3_cache", None)
        if cache is None or cache[0] != key:
            with torch.no_grad():
                cache = (key, ops.split_f32(conv.weight.detach().float().reshape(conv.weight.shape[0], -1).contiguous()),
                         conv.bias.detach().float().contiguous())
            conv._f32x3_cache = cache
        return cache

    def project_add(self, buf, d):
        """bf16( d + bf16(mlp3(groups)) ): the fp32-forced PE (Fi:163-165) added to the bf16 features (Fi:77-80), as the
        epilogue of the fp32-class GEMM."""
        cache = self._mlp3_split()
        rows = buf.shape[0] * buf.shape[1]
        N, K = self.mlp3.conv.weight.shape[0], buf.shape[2] // 2
        return ops.linear_f32x3_bf16(buf, cache[1], cache[2], rows, N, K, resid=d.contiguous()).reshape(buf.shape[0], buf.shape[1], N)

    def project(self, feat):
        """mlp3 (Conv1d 256 -> out_dim, bias) on the concatenated scales; a library GEMM -> main stream only."""
        w = self.mlp3.conv.weight.reshape(self.mlp3.conv.weight.shape[0], -1)
        N, K = w.shape
        rows = feat.numel() // K
        with torch.autocast("cuda", enabled=False):  # Fi:163-165 forces fp32 for the whole PE
            if feat.is_cuda and not ops.is_differentiable() and ops.f32x3_ok(rows, N, K):
                cache = self._mlp3_split()
                return ops.linear_f32x3(ops.split_f32(feat.float().reshape(rows, K).contiguous()), cache[1], cache[2], rows, N, K).reshape(*feat.shape[:-1], N)
            return F.linear(feat, w.float(), self.mlp3.conv.bias.float())

    def forward(self, pts):
        return self.project(self.groups(pts))
