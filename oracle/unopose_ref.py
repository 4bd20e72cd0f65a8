"""ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by ``unopose_amd``).

Plain torch-CPU fp32 restatement of the Python part of UNOPose's forward hot
path, written as pure functions over a flat ``state_dict`` in the reference's
key layout.  Every function cites the reference lines it
follows.  Formulation is deliberately the reference's own (materialised
tensors, ``torch.svd``), NOT the fused formulation of the HIP kernels, so the
two are independent.

Pinning: ``tests/golden/make_golden.py`` imports the reference modules from
``/root/reference`` (build container only), runs them on seeded inputs with
``_ext`` bound to ``oracle/pointnet2_oracle.py`` and stores inputs + outputs as
fixtures; ``tests/test_oracle_golden.py`` checks this file against them.
The ViT (timm 0.9.12, absent from /root/reference) is PARITY UNPINNED: it is
restated from timm's published VisionTransformer semantics.

Abbreviations for citations: M = oneref_grf_predator_pose_estimation_model.py,
F = oneref_feature_extraction.py, U = utils/model_utils.py, T = model/transformer.py,
C = oneref_predator_coarse_point_matching.py, Fi = oneref_predator_fine_point_matching.py,
P = pointnet2/pointnet2_utils.py (all under core/unopose/).
"""
import math

import torch
import torch.nn.functional as F


class Cfg(dict):
    """dict with attribute access and .get, like the omegaconf node the reference receives."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return Cfg(v) if isinstance(v, dict) and not isinstance(v, Cfg) else v


def default_cfg(**over):
    """configs/main_cfg.py:128-181 (model.cfg)."""
    cfg = dict(
        coarse_npoint=196,
        fine_npoint=2048,
        feature_extraction=dict(vit_type="vit_base_patch14_reg4_dinov2", up_type="linear", embed_dim=768, out_dim=256,
                                use_pyramid_feat=True, pretrained=False, vit_ckpt=None, freeze_vit=True,
                                depth=12, num_heads=12),
        geo_embedding=dict(sigma_d=0.2, sigma_a=15, angle_k=3, reduction_a="max", hidden_dim=256),
        coarse_point_matching=dict(nblock=3, input_dim=256, hidden_dim=256, out_dim=256, temp=0.1, sim_type="cosine",
                                   normalize_feat=True, nproposal1=6000, nproposal2=300),
        fine_point_matching=dict(nblock=3, input_dim=256, hidden_dim=256, out_dim=256, pe_radius1=0.1, pe_radius2=0.2,
                                 focusing_factor=3, temp=0.1, sim_type="cosine", normalize_feat=True, use_lrf=True,
                                 use_xyz=True, nsample1=64, nsample2=256),
    )
    for k, v in over.items():
        if isinstance(v, dict):
            cfg[k].update(v)
        else:
            cfg[k] = v
    return Cfg(cfg)


def _lin(x, sd, p):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _ln(x, sd, p, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


# --------------------------------------------------------------------- LRF ---
def _lrf_core(x, r, eps=1e-10):
    """Shared body of LRF.forward (U:777-823) and LRF_batch.forward (P:436-481).

    x: (..., 3, M) = centre - neighbour;  r: broadcastable to (..., 1, 1).
    Returns the (..., 3, 3) frame with columns [xp, yp, zp]."""
    M = x.shape[-1]
    xxt = x @ x.transpose(-1, -2) / M
    v = torch.svd(xxt)[2]
    z0 = v[..., -1]  # (..., 3)
    proj = (z0.unsqueeze(-2) @ x).squeeze(-2)  # (..., M)
    vote = (proj > 1e-3).sum(-1) - (proj < -1e-3).sum(-1)
    sign = 1.0 - 2.0 * (vote < 0).to(x.dtype)
    zp = sign.unsqueeze(-1) * z0  # (..., 3)
    xn = -x  # neighbour - centre
    nrm = (zp.unsqueeze(-2) @ xn).squeeze(-2)  # (..., M)
    vi = xn - zp.unsqueeze(-1) * nrm.unsqueeze(-2)  # (..., 3, M)
    x_l2 = torch.sqrt((xn ** 2).sum(-2))  # (..., M)
    alpha = (r.squeeze(-1) - x_l2) ** 2
    beta = nrm * nrm
    vi_c = ((alpha * beta).unsqueeze(-2) * vi).sum(-1)  # (..., 3)
    xp = vi_c / (torch.sqrt((vi_c ** 2).sum(-1, keepdim=True)) + eps)
    yp = torch.cross(xp, zp, dim=-1)
    return torch.stack((xp, yp, zp), dim=-1)


def get_batch_lrf(pts, use_ref_rad=False):
    """M:78-93 + U:766-823. pts (B,N,3) -> (B,N,3)."""
    cen = pts.mean(1, keepdim=True)
    if use_ref_rad:
        r = torch.ones(pts.shape[0])
    else:
        r = torch.norm(pts - cen, dim=2).max(1)[0]
    x = (cen - pts).transpose(1, 2)  # (B,3,N) = centre - p_i
    lrf = _lrf_core(x, r[:, None, None])
    out = lrf.transpose(1, 2) @ ((pts - cen).transpose(1, 2) / r[:, None, None])
    return out.transpose(1, 2).contiguous()


def lrf_batch(xyz, group, r):
    """P:429-481. xyz (B,N,3), group (B,N,3,M) -> (B,N,3,M)."""
    c = xyz.unsqueeze(3)
    lrf = _lrf_core(c - group, torch.tensor(float(r)).reshape(1, 1, 1, 1))
    return lrf.transpose(2, 3) @ ((group - c) / r)


def query_and_lrf_group(xyz, radius, nsample, ext):
    """P:522-584 with use_xyz=True, features ignored (use_feature=False): (B,6,N,ns)."""
    idx = ext.ball_query(xyz.contiguous(), xyz.contiguous(), radius, nsample)
    grouped = ext.group_points(xyz.transpose(1, 2).contiguous(), idx)  # (B,3,N,ns)
    lrf_feat = lrf_batch(xyz, grouped.transpose(1, 2), radius).transpose(1, 2)
    rel = grouped - xyz.transpose(1, 2).unsqueeze(-1)
    return torch.cat([rel, lrf_feat], dim=1)


def shared_mlp(x, sd, p, nlayer=3):
    """pointnet2/pytorch_utils.py:25-48,78-132: 1x1 Conv2d(no bias) + BN2d(eval) + ReLU."""
    for i in range(nlayer):
        q = f"{p}.layer{i}"
        x = F.conv2d(x, sd[q + ".conv.weight"])
        x = F.batch_norm(x, sd[q + ".normlayer.bn.running_mean"], sd[q + ".normlayer.bn.running_var"],
                         sd[q + ".normlayer.bn.weight"], sd[q + ".normlayer.bn.bias"], False, 0.0, 1e-5)
        x = F.relu(x)
    return x


def positional_encoding(pts, sd, p, cfg, ext):
    """Fi:159-178. pts (B,N,3) -> (B,N,256)."""
    pts = pts.float().contiguous()
    f1 = shared_mlp(query_and_lrf_group(pts, cfg.pe_radius1, cfg.get("nsample1", 32), ext), sd, p + ".mlp1")
    f1 = f1.max(dim=3)[0]
    f2 = shared_mlp(query_and_lrf_group(pts, cfg.pe_radius2, cfg.get("nsample2", 64), ext), sd, p + ".mlp2")
    f2 = f2.max(dim=3)[0]
    feat = torch.cat([f1, f2], dim=1)
    feat = F.conv1d(feat, sd[p + ".mlp3.conv.weight"], sd[p + ".mlp3.conv.bias"])
    return feat.transpose(1, 2)


# ------------------------------------------------------------ transformer ---
def pairwise_distance(x, y):
    """U:230-257 (not normalized, channel last)."""
    xy = x @ y.transpose(-1, -2)
    x2 = (x ** 2).sum(-1).unsqueeze(-1)
    y2 = (y ** 2).sum(-1).unsqueeze(-2)
    return (x2 - 2 * xy + y2).clamp(min=0.0)


def sinusoidal(idx, d_model):
    """T:258-284: interleaved [sin w0, cos w0, sin w1, ...]."""
    div = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
    om = idx.reshape(-1, 1, 1) * div.reshape(1, -1, 1)
    return torch.cat([torch.sin(om), torch.cos(om)], dim=2).reshape(*idx.shape, d_model)


def geo_embedding_indices(points, cfg):
    """T:303-333."""
    B, N, _ = points.shape
    dist = torch.sqrt(pairwise_distance(points, points))
    d_idx = dist / cfg.sigma_d
    k = cfg.angle_k
    knn = dist.topk(k=k + 1, dim=2, largest=False)[1][:, :, 1:]
    knn_pts = torch.gather(points.unsqueeze(1).expand(B, N, N, 3), 2, knn.unsqueeze(3).expand(B, N, k, 3))
    ref = (knn_pts - points.unsqueeze(2)).unsqueeze(2).expand(B, N, N, k, 3)
    anc = (points.unsqueeze(1) - points.unsqueeze(2)).unsqueeze(3).expand(B, N, N, k, 3)
    sin_v = torch.linalg.norm(torch.cross(ref, anc, dim=-1), dim=-1)
    cos_v = (ref * anc).sum(-1)
    a_idx = torch.atan2(sin_v, cos_v) * (180.0 / (cfg.sigma_a * math.pi))
    return d_idx, a_idx


def geo_embedding(points, sd, p, cfg):
    """T:335-350. points (B,n,3) -> (B,n,n,256)."""
    d_idx, a_idx = geo_embedding_indices(points, cfg)
    d_emb = _lin(sinusoidal(d_idx, cfg.hidden_dim), sd, p + ".proj_d")
    a_emb = _lin(sinusoidal(a_idx, cfg.hidden_dim), sd, p + ".proj_a")
    a_emb = a_emb.max(dim=3)[0] if cfg.reduction_a == "max" else a_emb.mean(dim=3)
    return d_emb + a_emb


def _heads(x, h):
    B, n, c = x.shape
    return x.reshape(B, n, h, c // h).permute(0, 2, 1, 3)


def _mha(q_in, kv_in, sd, p, h=4, embed=None):
    """T:95-148 (vanilla) / T:353-405 (RPE when `embed` is given)."""
    q = _heads(_lin(q_in, sd, p + ".proj_q"), h)
    k = _heads(_lin(kv_in, sd, p + ".proj_k"), h)
    v = _heads(_lin(kv_in, sd, p + ".proj_v"), h)
    s = torch.einsum("bhnc,bhmc->bhnm", q, k)
    if embed is not None:
        pe = _lin(embed, sd, p + ".proj_p")
        B, n, m, c = pe.shape
        pe = pe.reshape(B, n, m, h, c // h).permute(0, 3, 1, 2, 4)
        s = s + torch.einsum("bhnc,bhnmc->bhnm", q, pe)
    s = F.softmax(s / (q.shape[-1] ** 0.5), dim=-1)
    o = (s @ v).permute(0, 2, 1, 3)
    return o.reshape(o.shape[0], o.shape[1], -1)


def _attn_output(x, sd, p):
    """T:178-193."""
    hdn = _lin(F.relu(_lin(x, sd, p + ".expand")), sd, p + ".squeeze")
    return _ln(x + hdn, sd, p + ".norm")


def transformer_layer(x, mem, sd, p, embed=None):
    """T:196-227 (cross) / T:444-466 (RPE self)."""
    hdn = _lin(_mha(x, mem, sd, p + ".attention.attention", embed=embed), sd, p + ".attention.linear")
    x = _ln(hdn + x, sd, p + ".attention.norm")
    return _attn_output(x, sd, p + ".output")


def geometric_transformer(f0, e0, f1, e1, sd, p):
    """T:469-514 with blocks ["self","cross"], parallel=False."""
    f0 = transformer_layer(f0, f0, sd, p + ".layers.0", embed=e0)
    f1 = transformer_layer(f1, f1, sd, p + ".layers.0", embed=e1)
    f0 = transformer_layer(f0, f1, sd, p + ".layers.1")
    f1 = transformer_layer(f1, f0, sd, p + ".layers.1")
    return f0, f1


def linear_attention(xq, xkv, sd, p, h=4, focusing=3):
    """T:517-568."""
    q, k, v = _lin(xq, sd, p + ".proj_q"), _lin(xkv, sd, p + ".proj_k"), _lin(xkv, sd, p + ".proj_v")
    scale = F.softplus(sd[p + ".scale"])
    q = (F.relu(q) + 1e-6) / scale
    k = (F.relu(k) + 1e-6) / scale
    qn, kn = q.norm(dim=-1, keepdim=True), k.norm(dim=-1, keepdim=True)
    q, k = q ** focusing, k ** focusing
    q = q / q.norm(dim=-1, keepdim=True) * qn
    k = k / k.norm(dim=-1, keepdim=True) * kn
    B = q.shape[0]

    def split(x):
        return _heads(x, h).reshape(B * h, x.shape[1], -1)

    q, k, v = split(q), split(k), split(v)
    i, j, c, d = q.shape[-2], k.shape[-2], k.shape[-1], v.shape[-1]
    z = 1 / (torch.einsum("bic,bc->bi", q, k.sum(dim=1)) + 1e-6)
    if i * j * (c + d) > c * d * (i + j):
        kv = torch.einsum("bjc,bjd->bcd", k, v)
        x = torch.einsum("bic,bcd,bi->bid", q, kv, z)
    else:
        x = torch.einsum("bij,bjd,bi->bid", torch.einsum("bic,bjc->bij", q, k), v, z)
    return x.reshape(B, h, i, d).permute(0, 2, 1, 3).reshape(B, i, h * d)


def linear_transformer_layer(x, mem, sd, p, focusing=3):
    """T:571-612."""
    hdn = _lin(linear_attention(x, mem, sd, p + ".attention.attention", focusing=focusing), sd, p + ".attention.linear")
    x = _ln(hdn + x, sd, p + ".attention.norm")
    return _attn_output(x, sd, p + ".output")


def _sample_feats(dense, fps_idx, ext):
    """T:655-662 -- NB the off-by-one: `dense` still carries the bg token at row 0."""
    g = ext.gather_points(dense.transpose(1, 2).contiguous(), fps_idx).transpose(1, 2)
    return torch.cat([dense[:, 0:1], g], dim=1)


def sparse_to_dense_transformer(d0, e0, i0, d1, e1, i1, sd, p, ext, focusing=3):
    """T:643-671."""
    f0, f1 = _sample_feats(d0, i0, ext), _sample_feats(d1, i1, ext)
    f0, f1 = geometric_transformer(f0, e0, f1, e1, sd, p + ".sparse_layer")
    n0 = linear_transformer_layer(d0[:, 1:].contiguous(), f0[:, 1:].contiguous(), sd, p + ".dense_layer", focusing)
    n1 = linear_transformer_layer(d1[:, 1:].contiguous(), f1[:, 1:].contiguous(), sd, p + ".dense_layer", focusing)
    return torch.cat([f0[:, 0:1], n0], 1), torch.cat([f1[:, 0:1], n1], 1)


# -------------------------------------------------------------- pose heads ---
def weighted_procrustes(src, ref, w=None, thresh=0.0, eps=1e-5):
    """U:667-743: R, t with ref ~ R src + t."""
    if w is None:
        w = torch.ones_like(src[:, :, 0])
    w = torch.where(w < thresh, torch.zeros_like(w), w)
    w = (w / (w.sum(1, keepdim=True) + eps)).unsqueeze(2)
    sc = (src * w).sum(1, keepdim=True)
    rc = (ref * w).sum(1, keepdim=True)
    H = (src - sc).transpose(1, 2) @ (w * (ref - rc))
    U, _, V = torch.svd(H)
    eye = torch.eye(3).repeat(src.shape[0], 1, 1)
    eye[:, -1, -1] = torch.sign(torch.det(V @ U.transpose(1, 2)))
    R = V @ eye @ U.transpose(1, 2)
    t = rc.transpose(1, 2) - R @ sc.transpose(1, 2)
    return R, t.squeeze(2)


def feature_similarity(f1, f2, temp):
    """U:260-282, cosine + normalize."""
    return F.normalize(f1, p=2, dim=2) @ F.normalize(f2, p=2, dim=2).transpose(1, 2) / temp


def _scores(atten, scores, n1):
    """C:68-76 / Fi:91-99 (eval: only `score` is consumed downstream)."""
    s1, s2 = scores[:, 1:(n1 + 1)], scores[:, (n1 + 2):]
    return torch.clamp(torch.sigmoid(torch.cat((s1, s2), dim=1).squeeze(-1)), 0, 1)


def _soft_assignment(atten, score1, score2):
    """U:434-446 / U:538-547: mutual softmax x overlap scores (bg row/col get score 1)."""
    B = atten.shape[0]
    one = torch.ones(B, 1)
    s1 = torch.cat((one, score1), 1)[:, :, None]
    s2 = torch.cat((one, score2), 1)[:, None, :]
    a = torch.softmax(atten, dim=2) * torch.softmax(atten, dim=1) * s1 * s2
    label1 = a[:, 1:, :].max(dim=2)[1]
    label2 = a[:, :, 1:].max(dim=1)[1]
    return a, label1, label2


def compute_coarse_rt_overlap(atten, score, pts1, pts2, rand, n1p=6000, n2p=300, detail=False):
    """U:411-490. `rand` (B, 3*n1p) replaces the in-forward torch.rand (U:462)."""
    B, N1, _ = pts1.shape
    N2 = pts2.shape[1]
    atten, pts1, pts2 = atten.float(), pts1.float(), pts2.float()
    a, l1, l2 = _soft_assignment(atten, score[:, :N1].float(), score[:, N2:].float())
    w1, w2 = (l1 > 0).float(), (l2 > 0).float()
    ps = a[:, 1:, 1:] * w1.unsqueeze(2) * w2.unsqueeze(1)
    ps = ps.reshape(B, N1 * N2) ** 1.5
    cs = torch.cumsum(ps, dim=1)
    cs = cs / (cs[:, -1].unsqueeze(1) + 1e-8)
    idx = torch.searchsorted(cs, rand)
    i1 = torch.clamp(idx.div(N2, rounding_mode="floor"), max=N1 - 1)
    i2 = torch.clamp(idx % N2, max=N2 - 1)
    p1 = torch.gather(pts1, 1, i1.unsqueeze(2).expand(-1, -1, 3)).reshape(B * n1p, 3, 3)
    p2 = torch.gather(pts2, 1, i2.unsqueeze(2).expand(-1, -1, 3)).reshape(B * n1p, 3, 3)
    rs, ts = weighted_procrustes(p2, p1, None, thresh=0.5)
    rs, ts = rs.reshape(B, n1p, 3, 3), ts.reshape(B, n1p, 1, 3)
    p1, p2 = p1.reshape(B, n1p, 3, 3), p2.reshape(B, n1p, 3, 3)
    dis = torch.norm((p1 - ts) @ rs - p2, dim=3).mean(2)
    top = torch.topk(dis, n2p, dim=1, largest=False)[1]
    rs2 = torch.gather(rs, 1, top.reshape(B, n2p, 1, 1).expand(-1, -1, 3, 3))
    ts2 = torch.gather(ts, 1, top.reshape(B, n2p, 1, 1).expand(-1, -1, 1, 3))
    tp = ((pts1.unsqueeze(1) - ts2) @ rs2).reshape(B * n2p, -1, 3)
    model = pts2.unsqueeze(1).expand(-1, n2p, -1, -1).reshape(B * n2p, -1, 3)
    d = torch.sqrt(pairwise_distance(tp, model)).min(2)[0].reshape(B, n2p, -1)
    sc = w1.unsqueeze(1).sum(2) / ((d * w1.unsqueeze(1)).sum(2) + 1e-8)
    pose_score, best = sc.max(1)
    R = torch.gather(rs2, 1, best.reshape(B, 1, 1, 1).expand(-1, -1, 3, 3)).squeeze(1)
    t = torch.gather(ts2, 1, best.reshape(B, 1, 1, 1).expand(-1, -1, 1, 3)).squeeze(2).squeeze(1)
    if detail:
        return R, t, pose_score, dict(idx=idx, rs=rs, ts=ts, dis=dis, top=top, sc=sc, best=best, w1=w1, cs=cs)
    return R, t, pose_score


def compute_fine_rt_overlap(atten, score, pts1, pts2, dis_thres=0.15):
    """U:527-566."""
    atten, pts1, pts2 = atten.float(), pts1.float(), pts2.float()
    N1 = pts1.shape[1]
    a, l1, l2 = _soft_assignment(atten, score[:, :N1], score[:, N1:])
    a = a[:, 1:, 1:] * (l1 > 0).float().unsqueeze(2) * (l2 > 0).float().unsqueeze(1)
    pred = (a / (a.sum(2, keepdim=True) + 1e-6)) @ pts2
    R, t = weighted_procrustes(pred, pts1, a.sum(2), thresh=0.001)
    pp = (pts1 - t.unsqueeze(1)) @ R
    dis = torch.sqrt(pairwise_distance(pp, pts2)).min(2)[0]
    mask = (l1 > 0).float()
    ps = ((dis < dis_thres).float() * mask).sum(1) / (mask.sum(1) + 1e-8)
    return R, t, ps * mask.mean(1)


def coarse_point_matching(p1, f1, g1, p2, f2, g2, sd, p, cfg, rand, detail=False):
    """C:46-117 (eval)."""
    B, n1 = f1.shape[:2]
    bg = sd[p + ".bg_token"].expand(B, -1, -1)
    f1 = torch.cat([bg, _lin(f1, sd, p + ".in_proj")], 1)
    f2 = torch.cat([bg, _lin(f2, sd, p + ".in_proj")], 1)
    for i in range(cfg.nblock):
        f1, f2 = geometric_transformer(f1, g1, f2, g2, sd, f"{p}.transformers.{i}")
    scores = _lin(torch.cat((f1, f2), 1), sd, f"{p}.score_heads.{cfg.nblock - 1}")
    atten = feature_similarity(_lin(f1, sd, p + ".out_proj"), _lin(f2, sd, p + ".out_proj"), cfg.temp)
    score = _scores(atten, scores, n1)
    out = compute_coarse_rt_overlap(atten, score, p1, p2, rand, cfg.nproposal1, cfg.nproposal2)
    if detail:
        return out + (dict(atten=atten, score=score, f1=f1, f2=f2),)
    return out


def fine_point_matching(p1, f1, g1, i1, p2, f2, g2, i2, init_R, init_t, sd, p, cfg, ext, detail=False):
    """Fi:58-135 (eval). Returns R, t (radius-normalised units), score."""
    B, n1 = p1.shape[:2]
    p1_ = (p1 - init_t.unsqueeze(1)) @ init_R
    bg = sd[p + ".bg_token"].expand(B, -1, -1)
    f1 = torch.cat([bg, _lin(f1, sd, p + ".in_proj") + positional_encoding(p1_, sd, p + ".PE", cfg, ext)], 1)
    f2 = torch.cat([bg, _lin(f2, sd, p + ".in_proj") + positional_encoding(p2, sd, p + ".PE", cfg, ext)], 1)
    for i in range(cfg.nblock):
        f1, f2 = sparse_to_dense_transformer(f1, g1, i1, f2, g2, i2, sd, f"{p}.transformers.{i}", ext,
                                             cfg.focusing_factor)
    scores = _lin(torch.cat((f1, f2), 1), sd, f"{p}.score_heads.{cfg.nblock - 1}")
    atten = feature_similarity(_lin(f1, sd, p + ".out_proj"), _lin(f2, sd, p + ".out_proj"), cfg.temp)
    score = _scores(atten, scores, n1)
    out = compute_fine_rt_overlap(atten, score, p1, p2)
    if detail:
        return out + (dict(atten=atten, score=score, f1=f1, f2=f2),)
    return out


# --------------------------------------------------------------------- ViT ---
def vit_attention_core(qkv, heads):
    """timm 0.9.12 `Attention.forward` between the qkv and proj linears (non-fused branch): qkv (B,T,3D) ->
    (B,T,D); q scaled by head_dim^-0.5, softmax over keys, heads concatenated."""
    B, T, D3 = qkv.shape
    D = D3 // 3
    hd = D // heads
    qkv = qkv.reshape(B, T, 3, heads, hd).permute(2, 0, 3, 1, 4)
    a = torch.softmax((qkv[0] * hd ** -0.5) @ qkv[1].transpose(-2, -1), dim=-1) @ qkv[2]
    return a.transpose(1, 2).reshape(B, T, D)


def vit_taps(x, sd, p, depth=12, heads=12, patch=14):
    """timm 0.9.12 VisionTransformer as subclassed at F:24-42 (PARITY UNPINNED, App-D).
    Returns [norm(x_b) for b in the 4 tap blocks], each (B, 5+P, D)."""
    B = x.shape[0]
    x = F.conv2d(x, sd[p + ".patch_embed.proj.weight"], sd[p + ".patch_embed.proj.bias"], stride=patch)
    x = x.flatten(2).transpose(1, 2)
    x = x + sd[p + ".pos_embed"]
    x = torch.cat([sd[p + ".cls_token"].expand(B, -1, -1), sd[p + ".reg_token"].expand(B, -1, -1), x], dim=1)
    D = x.shape[-1]
    hd = D // heads
    n = depth // 4
    taps = {depth - 1, depth - n - 1, depth - 2 * n - 1, depth - 3 * n - 1}
    outs = []
    for i in range(depth):
        q = f"{p}.blocks.{i}"
        y = _ln(x, sd, q + ".norm1", 1e-6)
        a = _lin(vit_attention_core(_lin(y, sd, q + ".attn.qkv"), heads), sd, q + ".attn.proj")
        x = x + a * sd[q + ".ls1.gamma"]
        y = _ln(x, sd, q + ".norm2", 1e-6)
        y = _lin(F.gelu(_lin(y, sd, q + ".mlp.fc1")), sd, q + ".mlp.fc2")
        x = x + y * sd[q + ".ls2.gamma"]
        if i in taps:
            outs.append(_ln(x, sd, p + ".norm", 1e-6))
    return outs


def vit_ae(x, sd, p, cfg):
    """F:200-236 (up_type="linear", pyramid feats): (B,3,S,S) -> (B,256,S,S)."""
    B, _, H, W = x.shape
    outs = vit_taps(x, sd, p + ".vit", cfg.get("depth", 12), cfg.get("num_heads", 12))
    z = torch.cat([o[:, 5:, :] for o in outs], dim=2)
    side = H // 14
    z = _lin(z, sd, p + ".output_upscaling").reshape(B, side, side, 4, 4, cfg.out_dim).permute(0, 5, 1, 3, 2, 4)
    z = z.reshape(B, cfg.out_dim, 4 * side, 4 * side)
    return F.interpolate(z, (H, W), mode="bilinear", align_corners=False)


def chosen_pixel_feats(img, choose):
    """U:215-227."""
    B, C, H, W = img.shape
    return torch.gather(img.reshape(B, C, H * W), 2, choose.unsqueeze(1).expand(-1, C, -1)).transpose(1, 2).contiguous()


def sample_pts_feats(pts, feats, npoint, ext, extra=None):
    """U:137-177: FPS + gathers. Returns pts, [extra], feats, idx."""
    idx = ext.furthest_point_sampling(pts.float().contiguous(), npoint)

    def g(x):
        return ext.gather_points(x.float().transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()

    if extra is None:
        return g(pts), g(feats), idx
    return g(pts), g(extra), g(feats), idx


def feature_extraction(end_points, sd, p, cfg, npoint, ext):
    """F:245-298 (test branch where the template arrives in end_points)."""
    dense_fm = chosen_pixel_feats(vit_ae(end_points["rgb"], sd, p + ".rgb_net", cfg), end_points["rgb_choose"])
    tem = end_points["tem1_pts"]
    radius = torch.norm(tem - tem.mean(1, keepdim=True), dim=2).max(1)[0]
    dense_pm = end_points["pts"] / (radius.reshape(-1, 1, 1) + 1e-6)
    tem_n = tem / (radius.reshape(-1, 1, 1) + 1e-6)
    tem_f = chosen_pixel_feats(vit_ae(end_points["tem1_rgb"], sd, p + ".rgb_net", cfg), end_points["tem1_choose"])
    dense_po, dense_fo, _ = sample_pts_feats(tem_n, tem_f, npoint, ext)
    return dense_pm, dense_fm, dense_po, dense_fo, radius


def unopose_forward(end_points, sd, cfg, rand, ext, detail=False):
    """M:25-76 (eval). Returns dict with init_*/pred_* like the reference."""
    dense_pm, dense_fm, dense_po, dense_fo, radius = feature_extraction(
        end_points, sd, "feature_extraction", cfg.feature_extraction, cfg.fine_npoint, ext)
    pm_lrf = get_batch_lrf(end_points["pts"])
    po_lrf = get_batch_lrf(end_points["tem1_pts"])  # NB 5000 rows, gathered with 2048-set indices (App-E.1)
    B = dense_pm.shape[0]
    bg = torch.ones(B, 1, 3)
    sp_m, sp_m_lrf, sf_m, idx_m = sample_pts_feats(dense_pm, dense_fm, cfg.coarse_npoint, ext, extra=pm_lrf)
    geo_m = geo_embedding(torch.cat([bg, sp_m_lrf], 1), sd, "geo_embedding", cfg.geo_embedding)
    sp_o, sp_o_lrf, sf_o, idx_o = sample_pts_feats(dense_po, dense_fo, cfg.coarse_npoint, ext, extra=po_lrf)
    geo_o = geo_embedding(torch.cat([bg, sp_o_lrf], 1), sd, "geo_embedding", cfg.geo_embedding)
    out = {}
    R0, t0, s0 = coarse_point_matching(sp_m, sf_m, geo_m, sp_o, sf_o, geo_o, sd, "coarse_point_matching",
                                       cfg.coarse_point_matching, rand)
    out.update(init_R=R0, init_t=t0, init_pose_score=s0)
    R, t, s = fine_point_matching(dense_pm, dense_fm, geo_m, idx_m, dense_po, dense_fo, geo_o, idx_o, R0, t0, sd,
                                  "fine_point_matching", cfg.fine_point_matching, ext)
    out.update(pred_R=R, pred_t=t * (radius.reshape(-1, 1) + 1e-6), pred_pose_score=s)
    if detail:
        out.update(dense_pm=dense_pm, dense_fm=dense_fm, dense_po=dense_po, dense_fo=dense_fo, radius=radius,
                   fps_idx_m=idx_m, fps_idx_o=idx_o, geo_m=geo_m, geo_o=geo_o, sparse_pm=sp_m, sparse_po=sp_o,
                   sparse_pm_lrf=sp_m_lrf, sparse_po_lrf=sp_o_lrf, sparse_fm=sf_m, sparse_fo=sf_o)
    return out


# ------------------------------------------------------ state_dict layout ---
def state_dict_spec(cfg, img_size=224):
    """Key -> shape of UNOPose's state_dict (verified against the
    reference modules by tests/golden/make_golden.py via strict load_state_dict)."""
    spec = {}
    fe = cfg.feature_extraction
    D, depth = fe.embed_dim, fe.get("depth", 12)
    v = "feature_extraction.rgb_net.vit"
    P = (img_size // 14) ** 2
    spec[v + ".cls_token"] = (1, 1, D)
    spec[v + ".reg_token"] = (1, 4, D)
    spec[v + ".pos_embed"] = (1, P, D)
    spec[v + ".patch_embed.proj.weight"] = (D, 3, 14, 14)
    spec[v + ".patch_embed.proj.bias"] = (D,)
    for i in range(depth):
        b = f"{v}.blocks.{i}"
        for nm in ("norm1", "norm2"):
            spec[f"{b}.{nm}.weight"] = (D,)
            spec[f"{b}.{nm}.bias"] = (D,)
        spec[b + ".attn.qkv.weight"] = (3 * D, D)
        spec[b + ".attn.qkv.bias"] = (3 * D,)
        spec[b + ".attn.proj.weight"] = (D, D)
        spec[b + ".attn.proj.bias"] = (D,)
        spec[b + ".ls1.gamma"] = (D,)
        spec[b + ".ls2.gamma"] = (D,)
        spec[b + ".mlp.fc1.weight"] = (4 * D, D)
        spec[b + ".mlp.fc1.bias"] = (4 * D,)
        spec[b + ".mlp.fc2.weight"] = (D, 4 * D)
        spec[b + ".mlp.fc2.bias"] = (D,)
    spec[v + ".norm.weight"] = (D,)
    spec[v + ".norm.bias"] = (D,)
    spec[v + ".head.weight"] = (1000, D)  # unused (timm default num_classes)
    spec[v + ".head.bias"] = (1000,)
    spec["feature_extraction.rgb_net.output_upscaling.weight"] = (16 * fe.out_dim, 4 * D)
    spec["feature_extraction.rgb_net.output_upscaling.bias"] = (16 * fe.out_dim,)

    H = cfg.geo_embedding.hidden_dim
    spec["geo_embedding.embedding.div_term"] = (H // 2,)
    for nm in ("proj_d", "proj_a"):
        spec[f"geo_embedding.{nm}.weight"] = (H, H)
        spec[f"geo_embedding.{nm}.bias"] = (H,)

    def lin(p, o, i):
        spec[p + ".weight"] = (o, i)
        spec[p + ".bias"] = (o,)

    def tlayer(p, d, rpe):
        for nm in ("proj_q", "proj_k", "proj_v") + (("proj_p",) if rpe else ()):
            lin(f"{p}.attention.attention.{nm}", d, d)
        lin(p + ".attention.linear", d, d)
        spec[p + ".attention.norm.weight"] = (d,)
        spec[p + ".attention.norm.bias"] = (d,)
        lin(p + ".output.expand", 2 * d, d)
        lin(p + ".output.squeeze", d, 2 * d)
        spec[p + ".output.norm.weight"] = (d,)
        spec[p + ".output.norm.bias"] = (d,)

    c = cfg.coarse_point_matching
    p = "coarse_point_matching"
    spec[p + ".bg_token"] = (1, 1, c.hidden_dim)
    lin(p + ".in_proj", c.hidden_dim, c.input_dim)
    lin(p + ".out_proj", c.out_dim, c.hidden_dim)
    for i in range(c.nblock):
        lin(f"{p}.score_heads.{i}", 1, c.hidden_dim)
    for i in range(c.nblock):
        tlayer(f"{p}.transformers.{i}.layers.0", c.hidden_dim, True)
        tlayer(f"{p}.transformers.{i}.layers.1", c.hidden_dim, False)

    f = cfg.fine_point_matching
    p = "fine_point_matching"
    d = f.hidden_dim
    spec[p + ".bg_token"] = (1, 1, d)
    lin(p + ".in_proj", d, f.input_dim)
    lin(p + ".out_proj", f.out_dim, d)
    lin(p + ".dis_proj", 3, 2 * d)  # never used in forward
    for m in ("mlp1", "mlp2"):
        chans = [6, 32, 64, 128]
        for li in range(3):
            q = f"{p}.PE.{m}.layer{li}"
            spec[q + ".conv.weight"] = (chans[li + 1], chans[li], 1, 1)
            for nm in ("weight", "bias", "running_mean", "running_var"):
                spec[f"{q}.normlayer.bn.{nm}"] = (chans[li + 1],)
            spec[q + ".normlayer.bn.num_batches_tracked"] = ()
    spec[p + ".PE.mlp3.conv.weight"] = (d, 256, 1)
    spec[p + ".PE.mlp3.conv.bias"] = (d,)
    for i in range(f.nblock):
        lin(f"{p}.score_heads.{i}", 1, d)
    for i in range(f.nblock):
        t = f"{p}.transformers.{i}"
        tlayer(t + ".sparse_layer.layers.0", d, True)
        tlayer(t + ".sparse_layer.layers.1", d, False)
        spec[t + ".dense_layer.attention.attention.scale"] = (1, 1, d)
        tlayer(t + ".dense_layer", d, False)
    return spec


def random_state_dict(cfg, seed=0, img_size=224, prefix=None, tame=None):
    """Seeded random weights in the reference key layout (trained-like magnitudes;
    BN running stats randomised as SURVEY.md 8(d) prescribes).  `prefix` keeps only
    keys under it (e.g. "coarse_point_matching").

    `tame` (e.g. 0.1) makes the weights behave like a TRAINED matcher on congruent
    inputs so that end-to-end R/t are well conditioned (pure random weights give
    all-background assignments and a chaotic argmax): the token-mixing
    projections (attention.linear, output.squeeze, PE.mlp3) are scaled by `tame`, so
    identical input features stay identical through the blocks, and the overlap
    score heads get bias +2 / weights x0.2 (overlap scores ~0.9)."""
    import zlib

    sd = {}
    for k, shape in state_dict_spec(cfg, img_size).items():
        if prefix is not None and not k.startswith(prefix):
            continue
        g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(k.encode())) % (2 ** 31))
        leaf = k.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            t = torch.tensor(100, dtype=torch.int64)
        elif leaf == "div_term":
            n = shape[0] * 2
            t = torch.exp(torch.arange(0, n, 2).float() * (-math.log(10000.0) / n))
        elif leaf == "running_var":
            t = 0.5 + torch.rand(shape, generator=g)
        elif leaf == "running_mean":
            t = 0.1 * torch.randn(shape, generator=g)
        elif leaf == "gamma":
            t = 0.05 + 0.45 * torch.rand(shape, generator=g)
        elif leaf in ("cls_token", "reg_token", "pos_embed", "bg_token"):
            t = 0.02 * torch.randn(shape, generator=g)
        elif leaf == "scale":
            t = 0.1 * torch.randn(shape, generator=g)
        elif leaf == "weight" and len(shape) == 1:  # LayerNorm / BatchNorm scale
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif leaf == "bias":
            t = 0.05 * torch.randn(shape, generator=g)
        else:  # Linear / conv weight
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            t = torch.randn(shape, generator=g) / math.sqrt(fan_in)
        if tame is not None:
            if k.endswith("attention.linear.weight") or k.endswith("output.squeeze.weight") \
                    or k.endswith("PE.mlp3.conv.weight"):
                t = t * tame
            elif "score_heads" in k:
                t = t * 0 + 2.0 if leaf == "bias" else t * 0.2
        sd[k] = t
    return sd
