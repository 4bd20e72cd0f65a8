"""Generate the golden fixtures under tests/golden/ from the REFERENCE's own Python.

Runs ONLY in the build container (needs /root/reference; never on the GPU box,
never from tests).  It imports the reference modules as SURVEY.md App-G
describes (detectron2 logger stub, `_ext` bound to the C oracle, timm's
VisionTransformer stubbed by the oracle's ViT restatement -- the ViT arithmetic
itself is therefore NOT pinned by the reference, everything around it is),
runs them on seeded inputs, asserts that oracle/unopose_ref.py reproduces
every output, and stores inputs + reference outputs as small .npz files.

    python tests/golden/make_golden.py            # regenerate everything

Only tensors are written; no reference source is copied.
"""
import builtins
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.dont_write_bytecode = True

REF = "/root/reference"


def import_reference():
    sys.path.insert(0, REF)
    builtins.__POINTNET2_SETUP__ = True
    for name in ("detectron2", "detectron2.utils", "detectron2.utils.logger"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["detectron2.utils.logger"].log_first_n = lambda *a, **k: None
    sys.modules["detectron2.utils.logger"].log_every_n = lambda *a, **k: None

    # timm stub: VisionTransformer backed by the oracle's restatement (App-G step 5)
    from oracle import unopose_ref as R

    nn = torch.nn

    class _Attn(nn.Module):
        def __init__(self, dim, heads):
            super().__init__()
            self.heads = heads
            self.qkv = nn.Linear(dim, 3 * dim)
            self.proj = nn.Linear(dim, dim)

        def forward(self, x):
            B, N, C = x.shape
            qkv = self.qkv(x).reshape(B, N, 3, self.heads, C // self.heads).permute(2, 0, 3, 1, 4)
            a = torch.softmax((qkv[0] * (C // self.heads) ** -0.5) @ qkv[1].transpose(-2, -1), dim=-1) @ qkv[2]
            return self.proj(a.transpose(1, 2).reshape(B, N, C))

    class _LS(nn.Module):
        def __init__(self, dim):
            super().__init__()
            self.gamma = nn.Parameter(torch.ones(dim))

        def forward(self, x):
            return x * self.gamma

    class _Mlp(nn.Module):
        def __init__(self, dim):
            super().__init__()
            self.fc1 = nn.Linear(dim, 4 * dim)
            self.fc2 = nn.Linear(4 * dim, dim)

        def forward(self, x):
            return self.fc2(torch.nn.functional.gelu(self.fc1(x)))

    class _Block(nn.Module):
        def __init__(self, dim, heads, norm_layer):
            super().__init__()
            self.norm1, self.attn, self.ls1 = norm_layer(dim), _Attn(dim, heads), _LS(dim)
            self.norm2, self.mlp, self.ls2 = norm_layer(dim), _Mlp(dim), _LS(dim)

        def forward(self, x):
            x = x + self.ls1(self.attn(self.norm1(x)))
            return x + self.ls2(self.mlp(self.norm2(x)))

    class _PatchEmbed(nn.Module):
        def __init__(self, dim, patch, img):
            super().__init__()
            self.proj = nn.Conv2d(3, dim, patch, patch)
            self.num_patches = (img // patch) ** 2

        def forward(self, x):
            return self.proj(x).flatten(2).transpose(1, 2)

    class VisionTransformer(nn.Module):
        """Module-shaped stand-in with timm 0.9.12's attribute names and state_dict keys
        (reg_tokens=4, no_embed_class=True path only), so the reference's subclass forward
        (oneref_feature_extraction.py:28-42) runs on it and `load_state_dict(strict=True)`
        checks the key layout of SURVEY.md App-C."""

        def __init__(self, patch_size=14, embed_dim=768, depth=12, num_heads=12, init_values=None, reg_tokens=0,
                     no_embed_class=False, norm_layer=None, img_size=224, **kw):
            super().__init__()
            assert reg_tokens == 4 and no_embed_class
            self.patch_embed = _PatchEmbed(embed_dim, patch_size, img_size)
            self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
            self.reg_token = nn.Parameter(torch.zeros(1, reg_tokens, embed_dim))
            self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches, embed_dim))
            self.norm_pre = nn.Identity()
            self.blocks = nn.Sequential(*[_Block(embed_dim, num_heads, norm_layer) for _ in range(depth)])
            self.norm = norm_layer(embed_dim)
            self.head = nn.Linear(embed_dim, 1000)

        def _pos_embed(self, x):
            x = x + self.pos_embed
            return torch.cat([self.cls_token.expand(x.shape[0], -1, -1), self.reg_token.expand(x.shape[0], -1, -1), x], 1)

    timm = types.ModuleType("timm")
    timm.models = types.ModuleType("timm.models")
    timm.models.vision_transformer = types.ModuleType("timm.models.vision_transformer")
    timm.models.vision_transformer.VisionTransformer = VisionTransformer
    sys.modules["timm"] = timm
    sys.modules["timm.models"] = timm.models
    sys.modules["timm.models.vision_transformer"] = timm.models.vision_transformer

    import core.unopose.model.pointnet2.pointnet2_utils as pu
    from oracle.pointnet2_oracle import ext

    pu._ext = ext
    return pu


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"  wrote {name}.npz ({os.path.getsize(path) / 1024:.0f} KiB)")


def close(a, b, tol, what):
    a, b = torch.as_tensor(a).float(), torch.as_tensor(b).float()
    err = (a - b).abs().max().item()
    assert err <= tol, f"{what}: oracle vs reference max abs err {err} > {tol}"
    print(f"  {what}: oracle == reference (max abs err {err:.2e})")


def sub_sd(sd, prefix):
    n = len(prefix) + 1
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix + ".")}


def main():
    pu = import_reference()
    from oracle import unopose_ref as R
    from oracle.pointnet2_oracle import ext
    from helpers import object_cloud, constructed_similarity

    import core.unopose.model.transformer as T
    import core.unopose.utils.model_utils as U
    from core.unopose.model.oneref_predator_coarse_point_matching import CoarsePointMatchingOneRef
    from core.unopose.model.oneref_predator_fine_point_matching import FinePointMatchingOneRef, PositionalEncoding

    torch.set_grad_enabled(False)
    cfg = R.default_cfg()
    sd = R.random_state_dict(cfg, seed=0)
    g = torch.Generator().manual_seed(1234)

    def norm_cloud(n, repl=False):
        p = object_cloud(g, n, with_replacement=repl)
        c = p.mean(0, keepdim=True)
        return (p / (p - c).norm(dim=1).max()).contiguous()

    # ---------------------------------------------------------------- a8 global LRF
    print("lrf_global")
    pts = torch.stack([object_cloud(g, 300), object_cloud(g, 300, True)])
    cen = pts.mean(1, keepdim=True)
    r = torch.norm(pts - cen, dim=2).max(1)[0]
    ref = U.LRF(r)(cen.transpose(1, 2), pts.transpose(1, 2)).transpose(1, 2).contiguous()
    close(R.get_batch_lrf(pts), ref, 2e-5, "get_batch_lrf")
    save("lrf_global", pts=pts, out=ref)

    # ---------------------------------------------------------------- a5/a6 QueryAndLRFGroup
    print("query_lrf_group")
    xyz = torch.stack([norm_cloud(256), norm_cloud(256, True)])
    for rad, ns in ((0.2, 32), (0.4, 64)):
        grp = pu.QueryAndLRFGroup(rad, ns, use_xyz=True, use_feature=False)
        ref = grp(xyz.contiguous(), xyz.contiguous(), xyz.transpose(1, 2).contiguous())
        mine = R.query_and_lrf_group(xyz, rad, ns, ext)
        close(mine, ref, 5e-4, f"query_and_lrf_group r={rad}")
        save(f"query_lrf_group_r{rad}_ns{ns}", xyz=xyz, radius=np.float32(rad), nsample=ns, out=ref)

    # ---------------------------------------------------------------- a7 PositionalEncoding
    print("positional_encoding")
    fcfg = cfg.fine_point_matching
    pe = PositionalEncoding(256, r1=0.2, r2=0.4, nsample1=32, nsample2=64, use_lrf=True, use_xyz=True).eval()
    pe.load_state_dict(sub_sd(sd, "fine_point_matching.PE"), strict=True)
    ref = pe(xyz)
    pcfg = R.Cfg(dict(fcfg, pe_radius1=0.2, pe_radius2=0.4, nsample1=32, nsample2=64))
    close(R.positional_encoding(xyz, sd, "fine_point_matching.PE", pcfg, ext), ref, 2e-3, "positional_encoding")
    save("positional_encoding", xyz=xyz, r1=np.float32(0.2), r2=np.float32(0.4), ns1=32, ns2=64, out=ref)

    # ---------------------------------------------------------------- a11 geo embedding
    print("geo_embedding")
    geo = T.GeometricStructureEmbedding(cfg.geo_embedding).eval()
    geo.load_state_dict(sub_sd(sd, "geo_embedding"), strict=True)
    gp = torch.cat([torch.ones(2, 1, 3), torch.stack([norm_cloud(23), norm_cloud(23)])], 1)
    d_idx, a_idx = geo.get_embedding_indices(gp)
    ref = geo(gp)
    md, ma = R.geo_embedding_indices(gp, cfg.geo_embedding)
    close(md, d_idx, 1e-5, "geo d_indices")
    close(ma, a_idx, 1e-4, "geo a_indices")
    close(R.geo_embedding(gp, sd, "geo_embedding", cfg.geo_embedding), ref, 1e-4, "geo_embedding")
    save("geo_embedding", points=gp, d_idx=d_idx, a_idx=a_idx, out=ref)
    geo0, geo1 = ref[0:1], ref[1:2]

    # ---------------------------------------------------------------- a12/a13/a14 layers
    print("transformer layers")
    f0 = torch.randn(1, 24, 256, generator=g)
    f1 = torch.randn(1, 24, 256, generator=g)
    gt = T.GeometricTransformer(["self", "cross"], 256, 4, dropout=None, activation_fn="ReLU").eval()
    tp = "coarse_point_matching.transformers.0"
    gt.load_state_dict(sub_sd(sd, tp), strict=True)
    r_self = gt.layers[0](f0, f0, geo0)[0]
    r_cross = gt.layers[1](f0, f1)[0]
    r0, r1 = gt(f0, geo0, f1, geo1)
    close(R.transformer_layer(f0, f0, sd, tp + ".layers.0", embed=geo0), r_self, 2e-5, "RPE self layer")
    close(R.transformer_layer(f0, f1, sd, tp + ".layers.1"), r_cross, 2e-5, "cross layer")
    m0, m1 = R.geometric_transformer(f0, geo0, f1, geo1, sd, tp)
    close(m0, r0, 5e-5, "geometric_transformer f0")
    close(m1, r1, 5e-5, "geometric_transformer f1")
    save("transformer_layers", points=gp, f0=f0, f1=f1, rpe_self=r_self, cross=r_cross, gt0=r0, gt1=r1)

    print("sparse_to_dense")
    s2d = T.SparseToDenseTransformer(256, ["self", "cross"], num_heads=4, focusing_factor=3).eval()
    sp = "fine_point_matching.transformers.0"
    s2d.load_state_dict(sub_sd(sd, sp), strict=True)
    d0 = torch.randn(1, 101, 256, generator=g)
    d1 = torch.randn(1, 101, 256, generator=g)
    i0 = torch.randint(0, 100, (1, 23), generator=g, dtype=torch.int32)
    i1 = torch.randint(0, 100, (1, 23), generator=g, dtype=torch.int32)
    i0[0, 0] = 0
    ref_lin = s2d.dense_layer(d0[:, 1:].contiguous(), f0[:, 1:].contiguous())
    close(R.linear_transformer_layer(d0[:, 1:].contiguous(), f0[:, 1:].contiguous(), sd, sp + ".dense_layer"), ref_lin,
          5e-5, "linear transformer layer")
    rd0, rd1 = s2d(d0, geo0, i0, d1, geo1, i1)
    md0, md1 = R.sparse_to_dense_transformer(d0, geo0, i0, d1, geo1, i1, sd, sp, ext)
    close(md0, rd0, 1e-4, "sparse_to_dense f0")
    close(md1, rd1, 1e-4, "sparse_to_dense f1")
    save("sparse_to_dense", points=gp, d0=d0, d1=d1, i0=i0, i1=i1, sparse0=f0, linear=ref_lin, out0=rd0, out1=rd1)

    # ---------------------------------------------------------------- a18 procrustes
    print("weighted_procrustes")
    src = torch.randn(64, 40, 3, generator=g)
    Rg = torch.linalg.qr(torch.randn(64, 3, 3, generator=g))[0]
    Rg = Rg * torch.sign(torch.det(Rg)).reshape(-1, 1, 1)
    refp = src @ Rg.transpose(1, 2) + torch.randn(64, 1, 3, generator=g) + 0.01 * torch.randn(64, 40, 3, generator=g)
    w = torch.rand(64, 40, generator=g)
    w[w < 0.3] = 0
    Rr, tr = U.weighted_procrustes(src, refp, w.clone(), weight_thresh=0.001)
    Rm, tm = R.weighted_procrustes(src, refp, w, thresh=0.001)
    close(Rm, Rr, 1e-5, "procrustes R")
    close(tm, tr, 1e-5, "procrustes t")
    src3 = torch.randn(500, 3, 3, generator=g)
    ref3 = src3 @ Rg[:1].transpose(1, 2) + 0.02 * torch.randn(500, 3, 3, generator=g)
    R3, t3 = U.WeightedProcrustes()(src3, ref3, None)
    Rm3, tm3 = R.weighted_procrustes(src3, ref3, None, thresh=0.5)
    close(Rm3, R3, 1e-5, "procrustes (3 points) R")
    save("weighted_procrustes", src=src, ref=refp, w=w, R=Rr, t=tr, src3=src3, ref3=ref3, R3=R3, t3=t3)

    # ---------------------------------------------------------------- a17 / a20 pose heads
    print("coarse Rt (constructed similarity)")
    B, N = 2, 196
    p2 = torch.stack([norm_cloud(N), norm_cloud(N)])
    Rg2 = Rg[:B]
    tg2 = 0.1 * torch.randn(B, 3, generator=g)
    perm = torch.stack([torch.randperm(N, generator=g) for _ in range(B)])
    # p1 = R p2 + t on matched rows, some rows unmatched (background)
    p1 = torch.gather(p2, 1, perm.unsqueeze(2).expand(-1, -1, 3)) @ Rg2.transpose(1, 2) + tg2.unsqueeze(1)
    p1 = p1 + 0.003 * torch.randn(B, N, 3, generator=g)
    atten, score = constructed_similarity(perm, N, g, n_bg=40)
    rand = torch.rand(B, 18000, generator=g)
    _orig_rand = torch.rand
    torch.rand = lambda *a, **k: rand.clone()
    try:
        Rr, tr, sr = U.compute_coarse_Rt_overlap(atten, score, p1, p2, None, 6000, 300)
    finally:
        torch.rand = _orig_rand
    Rm, tm, sm, det = R.compute_coarse_rt_overlap(atten, score, p1, p2, rand, 6000, 300, detail=True)
    close(Rm, Rr, 1e-5, "coarse R")
    close(tm, tr, 1e-5, "coarse t")
    close(sm, sr, 1e-3, "coarse pose score")
    err = (Rr - Rg2).abs().max().item()
    print(f"  coarse R vs ground truth: {err:.3e}")
    save("coarse_rt", atten=atten, score=score, p1=p1, p2=p2, rand=rand, R=Rr, t=tr, pose_score=sr,
         hyp_idx=det["idx"].to(torch.int32), R_gt=Rg2, t_gt=tg2)

    print("fine Rt (constructed similarity)")
    N = 300
    p2 = torch.stack([norm_cloud(N), norm_cloud(N)])
    perm = torch.stack([torch.randperm(N, generator=g) for _ in range(B)])
    p1 = torch.gather(p2, 1, perm.unsqueeze(2).expand(-1, -1, 3)) @ Rg2.transpose(1, 2) + tg2.unsqueeze(1)
    p1 = p1 + 0.003 * torch.randn(B, N, 3, generator=g)
    atten, score = constructed_similarity(perm, N, g, n_bg=60)
    Rr, tr, sr = U.compute_fine_Rt_overlap(atten, score, p1, p2, None)
    Rm, tm, sm = R.compute_fine_rt_overlap(atten, score, p1, p2)
    close(Rm, Rr, 1e-5, "fine R")
    close(tm, tr, 1e-5, "fine t")
    close(sm, sr, 1e-5, "fine pose score")
    print(f"  fine R vs ground truth: {(Rr - Rg2).abs().max().item():.3e}")
    save("fine_rt", atten=atten, score=score, p1=p1, p2=p2, R=Rr, t=tr, pose_score=sr, R_gt=Rg2, t_gt=tg2)

    # ---------------------------------------------------------------- pos-embed resampling
    print("interpolate_pos_embed")
    pos = torch.randn(1, 37 * 37, 8, generator=g)

    class _M:
        pass

    m = _M()
    m.patch_embed = _M()
    m.patch_embed.num_patches = 256
    m.pos_embed = torch.zeros(1, 256, 8)
    ck = {"pos_embed": pos.clone()}
    U.interpolate_pos_embed(m, ck)
    save("interpolate_pos_embed", src=pos, out=ck["pos_embed"])

    # ---------------------------------------------------------------- a10 + a16 + a19 + a22
    # End-to-end on CONGRUENT pairs with "tamed" (trained-like) weights: the networks actually solve
    # these, so R / t are well conditioned and a 1e-4 comparison is meaningful.
    from core.unopose.model.oneref_grf_predator_pose_estimation_model import UNOPose
    from helpers import congruent_pair

    sdt = R.random_state_dict(cfg, seed=0, tame=0.1)

    def run_forward(tag, nq, nt, seed):
        print(f"UNOPose.forward [{tag}] nq={nq} nt={nt} (B=1, tamed weights, ViT = own restatement)")
        cfg_ref = R.default_cfg(fine_npoint=nq, feature_extraction=dict(freeze_vit=False))
        model = UNOPose(cfg_ref).eval()
        model.load_state_dict(sdt, strict=True)  # pins the whole key layout (App-C)
        gg = torch.Generator().manual_seed(seed)
        end_points, R_gt, t_gt = congruent_pair(gg, nq=nq, nt=nt, noise=5e-4)
        rand = torch.rand(1, 18000, generator=gg)
        torch.rand = lambda *a, **k: rand.clone()
        try:
            out = model(dict(end_points))
        finally:
            torch.rand = _orig_rand
        mine = R.unopose_forward(end_points, sdt, cfg_ref, rand, ext, detail=True)
        for k in ("init_R", "init_t", "pred_R", "pred_t"):
            close(mine[k], out[k], 1e-5, f"forward {k}")
        print(f"  pred_R vs ground truth {(out['pred_R'][0] - R_gt).abs().max().item():.2e}, "
              f"pred_t {(out['pred_t'][0] - t_gt).abs().max().item():.2e}, "
              f"score {out['pred_pose_score'].item():.3f}")
        taps_ref = model.feature_extraction.rgb_net.vit(end_points["rgb"])
        taps_mine = R.vit_taps(end_points["rgb"], sdt, "feature_extraction.rgb_net.vit")
        for a, b in zip(taps_mine, taps_ref):
            close(a, b, 1e-4, "ViT taps (functional oracle vs module stand-in; NOT timm)")
        dense = model.feature_extraction.get_img_feats(end_points["rgb"], end_points["rgb_choose"])
        close(mine["dense_fm"], dense, 1e-4, "ViT_AE post-processing + pixel gather")
        save(f"forward_{tag}", **{k: v for k, v in end_points.items()}, rand=rand, R_gt=R_gt, t_gt=t_gt,
             **{k: out[k] for k in ("init_R", "init_t", "init_pose_score", "pred_R", "pred_t", "pred_pose_score")},
             fps_idx_m=mine["fps_idx_m"], fps_idx_o=mine["fps_idx_o"], radius=mine["radius"],
             dense_fm_head=dense[:, :64])
        return model, mine, out, rand

    run_forward("full", 2048, 5000, 77)
    model, mine, out, rand = run_forward("cfg1", 1024, 2500, 78)

    print("coarse / fine matcher modules on the cfg1 intermediates")
    sp_m, sp_o = mine["sparse_pm"], mine["sparse_po"]
    lrf_m = torch.cat([torch.ones(1, 1, 3), mine["sparse_pm_lrf"]], 1)
    lrf_o = torch.cat([torch.ones(1, 1, 3), mine["sparse_po_lrf"]], 1)
    geo.load_state_dict(sub_sd(sdt, "geo_embedding"), strict=True)
    ge1, ge2 = geo(lrf_m), geo(lrf_o)
    close(ge1, mine["geo_m"], 1e-6, "geo embedding of the forward")
    torch.rand = lambda *a, **k: rand.clone()
    try:
        ep = model.coarse_point_matching(sp_m, mine["sparse_fm"], ge1, sp_o, mine["sparse_fo"], ge2, mine["radius"], {})
    finally:
        torch.rand = _orig_rand
    close(ep["init_R"], out["init_R"], 1e-6, "coarse module == forward")
    save("coarse_matcher", p1=sp_m, f1=mine["sparse_fm"], lrf1=lrf_m, p2=sp_o, f2=mine["sparse_fo"], lrf2=lrf_o,
         rand=rand, R=ep["init_R"], t=ep["init_t"], pose_score=ep["init_pose_score"])
    ep2 = {"init_R": ep["init_R"].clone(), "init_t": ep["init_t"].clone()}
    ep2 = model.fine_point_matching(mine["dense_pm"], mine["dense_fm"], ge1, mine["fps_idx_m"], mine["dense_po"],
                                    mine["dense_fo"], ge2, mine["fps_idx_o"], mine["radius"], ep2)
    close(ep2["pred_R"], out["pred_R"], 1e-6, "fine module == forward")
    Rm, tm, sm, det = R.fine_point_matching(mine["dense_pm"], mine["dense_fm"], ge1, mine["fps_idx_m"],
                                            mine["dense_po"], mine["dense_fo"], ge2, mine["fps_idx_o"],
                                            ep["init_R"], ep["init_t"], sdt, "fine_point_matching",
                                            cfg.fine_point_matching, ext, detail=True)
    save("fine_matcher", p1=mine["dense_pm"], f1=mine["dense_fm"], lrf1=lrf_m, i1=mine["fps_idx_m"],
         p2=mine["dense_po"], f2=mine["dense_fo"], lrf2=lrf_o, i2=mine["fps_idx_o"], init_R=ep["init_R"],
         init_t=ep["init_t"], radius=mine["radius"], R=ep2["pred_R"], t=ep2["pred_t"],
         pose_score=ep2["pred_pose_score"], f1_out=det["f1"][:, :64], f2_out=det["f2"][:, :64], score=det["score"],
         atten_rowmax=det["atten"].max(2)[0], atten_colmax=det["atten"].max(1)[0])
    production_size_layers(pu, sd, cfg)
    print("done")


def seeded(shape, seed, scale=1.0):
    """Fixture inputs too large to commit are regenerated from a seed (torch's CPU generator is
    deterministic for a given torch build; `tests/helpers.py::seeded_checked` re-derives them and checks the
    stored checksum before use)."""
    return scale * torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def checksum(t):
    t = t.double().flatten()
    return np.array([t.sum().item(), (t * torch.arange(1, t.numel() + 1, dtype=torch.float64)).sum().item() / t.numel()])


def production_size_layers(pu, sd, cfg):
    """Layer-level fixtures at the PRODUCTION sizes (197 tokens, 2048 dense points, PE radii 0.1 / 0.2 with
    64 / 256 neighbours) with UNTAMED random weights, from the reference modules themselves.  Large
    tensors are stored as row subsets (the index lists are in the fixture); large random inputs as seeds."""
    from oracle import unopose_ref as R
    from oracle.pointnet2_oracle import ext
    from helpers import object_cloud, constructed_similarity

    import core.unopose.model.transformer as T
    import core.unopose.utils.model_utils as U
    from core.unopose.model.oneref_predator_fine_point_matching import PositionalEncoding

    g = torch.Generator().manual_seed(4321)

    def norm_cloud(n, repl=False):
        p = object_cloud(g, n, with_replacement=repl)
        c = p.mean(0, keepdim=True)
        return (p / (p - c).norm(dim=1).max()).contiguous()

    # ---- a11 at n = 197 ----------------------------------------------------------------------------
    print("geo_embedding n=197")
    geo = T.GeometricStructureEmbedding(cfg.geo_embedding).eval()
    geo.load_state_dict(sub_sd(sd, "geo_embedding"), strict=True)
    # the points the model feeds here (M:28-47): global-LRF coordinates (centred, |p| <= 1) of the 196 FPS-selected
    # coarse points, background point (1,1,1) in front.  (Un-centred coordinates would make the reference's own
    # |x|^2 - 2xy + |y|^2 distances pure rounding noise at the close pairs.)
    def coarse_lrf_points(repl):
        cloud = object_cloud(g, 2048, with_replacement=repl)[None]
        lrf = R.get_batch_lrf(cloud)
        rad = torch.norm(cloud - cloud.mean(1, keepdim=True), dim=2).max(1)[0]
        idx = ext.furthest_point_sampling((cloud / rad.reshape(-1, 1, 1)).contiguous(), 196)
        return torch.gather(lrf, 1, idx.long().unsqueeze(2).expand(-1, -1, 3))[0]

    gp = torch.cat([torch.ones(2, 1, 3), torch.stack([coarse_lrf_points(False), coarse_lrf_points(True)])], 1)
    E = geo(gp)  # (2,197,197,256)
    close(R.geo_embedding(gp, sd, "geo_embedding", cfg.geo_embedding), E, 1e-4, "geo_embedding n=197")
    sel_i = torch.arange(0, 197, 16)
    sel_j = torch.arange(0, 197, 4)
    save("geo_embedding_n197", points=gp, sel_i=sel_i.to(torch.int32), sel_j=sel_j.to(torch.int32),
         out=E[:, sel_i][:, :, sel_j])

    # ---- a12 / a13: GeometricTransformer at n = 197 --------------------------------------------------
    print("GeometricTransformer n=197 (untamed)")
    gt = T.GeometricTransformer(["self", "cross"], 256, 4, dropout=None, activation_fn="ReLU").eval()
    tp = "coarse_point_matching.transformers.1"
    gt.load_state_dict(sub_sd(sd, tp), strict=True)
    f0, f1 = seeded((1, 197, 256), 501), seeded((1, 197, 256), 502)
    r_self = gt.layers[0](f0, f0, E[0:1])[0]
    r_cross = gt.layers[1](f0, f1)[0]
    r0, r1 = gt(f0, E[0:1], f1, E[1:2])
    close(R.transformer_layer(f0, f0, sd, tp + ".layers.0", embed=E[0:1]), r_self, 5e-5, "RPE self layer n=197")
    m0, m1 = R.geometric_transformer(f0, E[0:1], f1, E[1:2], sd, tp)
    close(m0, r0, 1e-4, "geometric_transformer n=197 f0")
    close(m1, r1, 1e-4, "geometric_transformer n=197 f1")
    save("geometric_transformer_n197", points=gp, f0=f0, f1=f1, rpe_self=r_self, cross=r_cross, gt0=r0, gt1=r1)

    # ---- a14 / a15: SparseToDenseTransformer at 2049 / 197 -------------------------------------------
    print("SparseToDenseTransformer 2049/197 (untamed)")
    s2d = T.SparseToDenseTransformer(256, ["self", "cross"], num_heads=4, focusing_factor=3).eval()
    sp = "fine_point_matching.transformers.1"
    s2d.load_state_dict(sub_sd(sd, sp), strict=True)
    d0, d1 = seeded((1, 2049, 256), 601), seeded((1, 2049, 256), 602)
    i0 = ext.furthest_point_sampling(norm_cloud(2048)[None], 196)
    i1 = ext.furthest_point_sampling(norm_cloud(2048, True)[None], 196)
    ref_lin = s2d.dense_layer(d0[:, 1:].contiguous(), f0[:, 1:].contiguous())
    close(R.linear_transformer_layer(d0[:, 1:].contiguous(), f0[:, 1:].contiguous(), sd, sp + ".dense_layer"), ref_lin,
          1e-4, "linear transformer layer 2048x196")
    att = s2d.dense_layer.attention.attention
    ref_core = att(d0[:, 1:].contiguous(), f0[:, 1:].contiguous(), f0[:, 1:].contiguous())  # LinearAttention.forward alone (T:531-568)
    close(R.linear_attention(d0[:, 1:].contiguous(), f0[:, 1:].contiguous(), sd, sp + ".dense_layer.attention.attention"),
          ref_core, 1e-4, "linear attention core 2048x196")
    rd0, rd1 = s2d(d0, E[0:1], i0, d1, E[1:2], i1)
    md0, md1 = R.sparse_to_dense_transformer(d0, E[0:1], i0, d1, E[1:2], i1, sd, sp, ext)
    close(md0, rd0, 2e-4, "sparse_to_dense 2049 f0")
    close(md1, rd1, 2e-4, "sparse_to_dense 2049 f1")
    rows = torch.arange(0, 2048, 8)
    rows1 = torch.cat([torch.zeros(1, dtype=torch.long), 1 + rows])  # bg row + every 8th dense row
    save("sparse_to_dense_2049", points=gp, d_seeds=np.array([601, 602]), d_checksum=np.stack([checksum(d0), checksum(d1)]),
         sparse0=f0, i0=i0, i1=i1, rows=rows.to(torch.int32), rows1=rows1.to(torch.int32), linear=ref_lin[:, rows],
         linear_core=ref_core[:, rows], out0=rd0[:, rows1], out1=rd1[:, rows1])

    # ---- a7: PositionalEncoding at the configured radii / neighbour counts, N = 2048 -----------------
    print("PositionalEncoding r=0.1/0.2 ns=64/256 N=2048")
    fcfg = cfg.fine_point_matching
    pe = PositionalEncoding(256, r1=fcfg.pe_radius1, r2=fcfg.pe_radius2, nsample1=fcfg.nsample1, nsample2=fcfg.nsample2,
                            use_lrf=True, use_xyz=True).eval()
    pe.load_state_dict(sub_sd(sd, "fine_point_matching.PE"), strict=True)
    xyz = torch.stack([norm_cloud(2048), norm_cloud(2048, True)])
    ref = pe(xyz)
    close(R.positional_encoding(xyz, sd, "fine_point_matching.PE", fcfg, ext), ref, 5e-3, "positional_encoding prod")
    for rad, ns in ((fcfg.pe_radius1, fcfg.nsample1), (fcfg.pe_radius2, fcfg.nsample2)):
        grp = pu.QueryAndLRFGroup(rad, ns, use_xyz=True, use_feature=False)
        o = grp(xyz.contiguous(), xyz.contiguous(), xyz.transpose(1, 2).contiguous())
        close(R.query_and_lrf_group(xyz, rad, ns, ext), o, 2e-3, f"query_and_lrf_group prod r={rad}")
    pts_sel = torch.arange(0, 2048, 4)
    save("positional_encoding_prod", xyz=xyz, r1=np.float32(fcfg.pe_radius1), r2=np.float32(fcfg.pe_radius2),
         ns1=fcfg.nsample1, ns2=fcfg.nsample2, sel=pts_sel.to(torch.int32), out=ref[:, pts_sel])

    # ---- a20 at 2049 x 2049 ---------------------------------------------------------------------------
    print("fine Rt 2049x2049 (constructed similarity, regenerated from its seed)")
    gg = torch.Generator().manual_seed(9001)
    N = 2048
    p2 = torch.stack([norm_cloud(N)])
    perm = torch.stack([torch.randperm(N, generator=gg)])
    Q = torch.linalg.qr(torch.randn(1, 3, 3, generator=gg))[0]
    Q = Q * torch.sign(torch.det(Q)).reshape(-1, 1, 1)
    tg = 0.1 * torch.randn(1, 3, generator=gg)
    p1 = torch.gather(p2, 1, perm.unsqueeze(2).expand(-1, -1, 3)) @ Q.transpose(1, 2) + tg.unsqueeze(1)
    p1 = p1 + 0.003 * torch.randn(1, N, 3, generator=gg)
    atten, score = constructed_similarity(perm, N, torch.Generator().manual_seed(9002), n_bg=300)
    Rr, tr, sr = U.compute_fine_Rt_overlap(atten, score, p1, p2, None)
    Rm, tm, sm = R.compute_fine_rt_overlap(atten, score, p1, p2)
    close(Rm, Rr, 1e-5, "fine R 2049")
    close(tm, tr, 1e-5, "fine t 2049")
    close(sm, sr, 1e-5, "fine score 2049")
    print(f"  fine R vs ground truth: {(Rr - Q).abs().max().item():.3e}")
    save("fine_rt_2049", p1=p1, p2=p2, perm=perm.to(torch.int32), sim_seed=9002, n_bg=300, sim_checksum=checksum(atten),
         score=score, R=Rr, t=tr, pose_score=sr, R_gt=Q, t_gt=tg)


if __name__ == "__main__":
    main()
