"""Point-cloud geometry: frames, weighted Procrustes, radius normalisation, row gathers, the geometric structure embedding (eval kernels and the
table kernels under autograd) and the fused positional encoding (csrc/geom.hip, embed.hip, pe.hip, glue.hip).  Inputs are CUDA float32 tensors; there is
no CPU path (RuntimeError)."""
import ctypes

import torch
import torch.nn.functional as F

from .._lib import call, check_f32, lib, on_device, ptr, stream_ptr
from ..pointnet2 import _ext
from . import _state as st
from .common import _c, _own_glue, _params_key, note_fallback
from .dense import _lin, mlp


def lrf_global(pts, use_ref_rad=False):
    """get_batch_lrf (oneref_grf_predator_pose_estimation_model.py:78-93). (B,N,3)->(B,N,3)."""
    pts = _c(pts.float())
    check_f32(pts, "pts")
    B, N, _ = pts.shape
    out = torch.empty_like(pts)
    with on_device(pts.device):
        call("unopose_lrf_global", ptr(pts), B, N, int(bool(use_ref_rad)), ptr(out), stream_ptr())
    return out


def query_lrf_group(xyz, radius, nsample):
    """QueryAndLRFGroup(radius, nsample, use_xyz=True)(xyz, xyz, feats) (pointnet2_utils.py:522-584).
    (B,N,3) -> (B,6,N,nsample)."""
    xyz = _c(xyz.float())
    check_f32(xyz, "xyz")
    B, N, _ = xyz.shape
    out = torch.empty(B, 6, N, int(nsample), dtype=torch.float32, device=xyz.device)
    with on_device(xyz.device):
        call("unopose_query_lrf_group", ptr(xyz), B, N, float(radius), int(nsample), ptr(out), stream_ptr())
    return out


def lrf_group_idx(xyz, new_xyz, idx, radius):
    """The general form of QueryAndLRFGroup.forward (pointnet2_utils.py:548-565): neighbour lists `idx` (B,N,S) int32 given
    (ball_query around `new_xyz`, possibly re-drawn by sample_uniformly) -> (B,6,N,S) with channels 0-2 relative to
    new_xyz and the frame / channels 3-5 relative to xyz, as the reference computes them."""
    xyz, new_xyz = _c(xyz.float()), _c(new_xyz.float())
    check_f32(xyz, "xyz")
    check_f32(new_xyz, "new_xyz")
    if new_xyz.shape != xyz.shape:
        raise ValueError(f"QueryAndLRFGroup: LRF_batch(xyz, grouped) needs npoint == N, got xyz {tuple(xyz.shape)} "
                         f"new_xyz {tuple(new_xyz.shape)} (pointnet2_utils.py:432-436)")
    idx = _c(idx.to(torch.int32))
    B, N, _ = xyz.shape
    if idx.shape[:2] != (B, N) or not idx.is_cuda:
        raise ValueError(f"idx must be a (B,N,S) device tensor, got {tuple(idx.shape)}")
    S = idx.shape[2]
    out = torch.empty(B, 6, N, S, dtype=torch.float32, device=xyz.device)
    with on_device(xyz.device):
        call("unopose_lrf_group_idx", ptr(xyz), ptr(new_xyz), ptr(idx), B, N, float(radius), S, ptr(out), stream_ptr())
    return out


def weighted_procrustes(src, ref, weights=None, weight_thresh=0.0, eps=1e-5):
    """weighted_procrustes (utils/model_utils.py:667-743): R (M,3,3), t (M,3), ref ~ R src + t."""
    src, ref = _c(src.float()), _c(ref.float())
    check_f32(src, "src_points")
    check_f32(ref, "ref_points")
    M, N, _ = src.shape
    if weights is not None:
        weights = _c(weights.float())
        check_f32(weights, "weights")
    R = torch.empty(M, 3, 3, dtype=torch.float32, device=src.device)
    t = torch.empty(M, 3, dtype=torch.float32, device=src.device)
    with on_device(src.device):
        call("unopose_weighted_procrustes", ptr(src), ptr(ref), ptr(weights) if weights is not None else None, M, N,
             float(weight_thresh), float(eps), ptr(R), ptr(t), stream_ptr())
    return R, t


def cloud_radius(pts):
    """max_i |p_i - mean(p)| per cloud, pts (B,N,3) -> (B,) (the normalisation radius of the forward)"""
    if not _own_glue(pts):
        return torch.norm(pts - pts.mean(1, keepdim=True), dim=2).max(1)[0]
    p = _c(pts.float())
    out = torch.empty(p.shape[0], dtype=torch.float32, device=p.device)
    with on_device(p.device):
        call("unopose_cloud_radius", ptr(p), p.shape[0], p.shape[1], ptr(out), stream_ptr())
    return out


def scale_by_radius(x, radius, multiply=False, eps=1e-6):
    """x / (radius[b] + eps) (or x * (...)) for x (B, ...) fp32"""
    if not (_own_glue(x) and x.dtype == torch.float32):
        s = (radius + eps).reshape(-1, *([1] * (x.dim() - 1)))
        return x * s if multiply else x / s
    xc = _c(x)
    out = torch.empty_like(xc)
    B = xc.shape[0]
    if xc.numel() == 0:
        return out
    r = radius.float().reshape(-1)
    if r.numel() != B:  # (the torch expression broadcast a single radius over the batch: the kernel indexes radius[b])
        if r.numel() != 1:
            raise ValueError(f"scale_by_radius: {r.numel()} radii for a batch of {B}")
        r = r.expand(B)
    with on_device(x.device):
        call("unopose_scale_by_radius", ptr(xc), B, xc.numel() // B, ptr(_c(r)), float(eps), int(multiply), ptr(out), stream_ptr())
    return out


def gather_rows(feats, idx, off=0, alt=None, prepend=False, out=None):
    """out[b,j,:] = feats[b, idx[b,j] - off, :]  (the (B,N,C)-layout twin of gather_operation; avoids the two transpose copies
    around every reference call, model_utils.py:146-149, transformer.py:658).  `alt` (B,1,C) / (B,C): the row taken where
    idx - off < 0, and -- with `prepend` -- an extra row 0 of the output: the background-token sampling of the sparse-to-dense
    block in one launch (csrc/glue.hip; index cast + clamp + gather + compare + where + cat otherwise).  Rows are raw bytes: any
    element type whose row is a multiple of 4 bytes (the int64 pixel indices of the FPS subset are gathered as 8-byte rows)."""
    B, N, C = feats.shape
    J = idx.shape[1]
    if (feats.is_cuda and not st._DIFF and not feats.requires_grad and feats.is_contiguous() and idx.is_contiguous() and idx.dtype in (torch.int32, torch.int64)
            and (C * feats.element_size()) % 4 == 0 and B <= 65535):
        alt_stride = 0
        if alt is not None:
            alt = alt.reshape(B, C)
            if alt.dtype != feats.dtype or alt.stride(1) != 1 or 0 < alt.stride(0) < C:  # (rows any distance apart are read in place: the background
                alt = _c(alt.to(feats.dtype))                                            #  tokens as row 0 of a (B, 1 + n, C) tensor; distance 0 = one row for all)
            alt_stride = (alt.stride(0) if B > 1 else C) * feats.element_size()
            if alt_stride % 4:
                alt, alt_stride = alt.contiguous(), C * feats.element_size()
        if out is None:
            out = torch.empty(B, J + int(prepend), C, dtype=feats.dtype, device=feats.device)
        else:  # a destination the caller names (one half of a stacked tensor): written in place
            assert out.shape == (B, J + int(prepend), C) and out.dtype == feats.dtype and out.is_contiguous() and out.device == feats.device
        with on_device(feats.device):
            call("unopose_gather_rows", ptr(feats), B, N, C * feats.element_size(), ptr(idx), int(idx.dtype == torch.int64), J, int(off),
                 None if alt is None else ptr(alt), alt_stride, int(prepend), ptr(out), stream_ptr())
        return out
    if out is not None:
        out.copy_(gather_rows(feats, idx, off, alt, prepend))
        return out
    if off == 0 and alt is None:
        return torch.gather(feats, 1, idx.long().unsqueeze(2).expand(-1, -1, C))
    i = idx.long() - off
    g = torch.gather(feats, 1, i.clamp(min=0).unsqueeze(2).expand(-1, -1, C))
    if alt is not None:
        a = alt.reshape(B, 1, C).to(g.dtype)
        g = torch.where((i < 0).unsqueeze(-1), a, g)
        if prepend:
            g = torch.cat([a, g], 1)
    return g


def pairwise_distance(x, y):
    """model_utils.py:230-257."""
    xy = x @ y.transpose(-1, -2)
    return ((x ** 2).sum(-1).unsqueeze(-1) - 2 * xy + (y ** 2).sum(-1).unsqueeze(-2)).clamp(min=0.0)


def _bf16_split(w):
    """bf16 hi/lo parts of an fp32 tensor: w ~ hi + lo with ~2^-16 relative error."""
    hi = w.float().to(torch.bfloat16)
    lo = (w.float() - hi.float()).to(torch.bfloat16)
    return hi.contiguous(), lo.contiguous()


def _mfma_fragment_order(w):
    """(256 out, 256 in) -> [k/16][out/32][(k%16)/8][out%32][k%8], the B-operand order of
    v_mfma_f32_32x32x16_bf16 when wave w owns output channels [32w, 32w+32)."""
    return w.reshape(8, 32, 16, 2, 8).permute(2, 0, 3, 1, 4).contiguous()


_GEO_HINV = 4        # table nodes per unit index (csrc/embed.hip: GT_HINV)


_GEO_D_RANGE = 64    # distance indices the table covers (the kernel evaluates larger ones from the defining sum)


_GEO_TABLE_UNAVAILABLE = {}  # device -> True once the table kernel's LDS opt-in has failed there


def _geo_tables(m, key, npoint=4):
    """proj_d(sinus(x)) and proj_a(sinus(x)) without their biases on the grid x = (r - (npoint / 2 - 1)) / 4 (fp64 sum, stored fp32),
    cached per weight version and interpolation order: what `unopose_geo_embedding_table` interpolates.  None when the angle table
    would not fit the kernel's LDS."""
    cache = m.__dict__.setdefault("_hip_tables", {})
    c = cache.get(npoint)
    if c is not None and c[0] == key:
        return c[1]
    import math
    lo = npoint // 2 - 1
    rows_a = int(math.floor(math.pi * float(m.factor_a) * _GEO_HINV)) + npoint + 1
    rows_d = _GEO_D_RANGE * _GEO_HINV + npoint
    tabs = None
    if rows_a <= 80 - npoint:
        div = m.embedding.div_term.detach().double()

        def table(rows, lin):
            x = (torch.arange(rows, device=div.device, dtype=torch.float64) - float(lo)) / _GEO_HINV
            om = x[:, None] * div[None, :]
            s = torch.stack([torch.sin(om), torch.cos(om)], dim=-1).reshape(rows, -1)
            return (s @ lin.weight.detach().double().t()).float().contiguous()

        tabs = (table(rows_d, m.proj_d), table(rows_a, m.proj_a), m.proj_d.weight.detach().float().contiguous())
    cache[npoint] = (key, tabs)
    return tabs


def geo_embedding(points, m, out_dtype=None):
    """GeometricStructureEmbedding.forward (transformer.py:303-350) as ONE fused HIP kernel
    (sinusoid generation -> MFMA -> max-over-k epilogue; csrc/embed.hip).  Under autocast(bf16) the
    result is bf16 with plain bf16 operands (what proj_d / proj_a produce under autocast in the
    reference); otherwise fp32 with hi/lo-split operands (fp32-class accuracy).  Round 5: both results come
    from the table-interpolated kernel (`unopose_geo_embedding_table`: fp32 Lagrange interpolation on tables of the
    two projections -- 4-point for the bf16 result, no bf16 operand rounding at all, 6-point (error ~1e-6) for the
    fp32 result; 3x / 5x faster than the contractions); `GEO_TABLE` / `GEO_TABLE_F32` switch back."""
    if st._DIFF and torch.is_grad_enabled() and geo_embedding_train_ok(points, m):
        return _GeoEmbedFn.apply(_c(points.detach().float()), m.proj_d.weight, m.proj_d.bias, m.proj_a.weight, m.proj_a.bias, m)
    if st._DIFF or m.proj_d.weight.shape != (256, 256) or m.angle_k != 3 or points.shape[1] < 4:
        if not st._DIFF:
            note_fallback("geo_embedding", f"hidden_dim {tuple(m.proj_d.weight.shape)} / angle_k {m.angle_k} / {points.shape[1]} points (kernel: 256, 3, >= 4)")
        return geo_embedding_torch(points, m)  # other widths / k: op-by-op GPU composite
    points = _c(points.float())
    check_f32(points, "points")
    B, n, _ = points.shape
    if out_dtype is None:
        out_dtype = torch.bfloat16 if torch.is_autocast_enabled() else torch.float32
    bf16_out = out_dtype == torch.bfloat16
    cache = getattr(m, "_hip_cache", None)
    # (every tensor / number the derived state is computed from: an in-place edit of a bias alone, or a changed factor_a, rebuilds it)
    key = (m.proj_d.weight._version, m.proj_a.weight._version, m.proj_d.bias._version, m.proj_a.bias._version, m.proj_d.weight.data_ptr(),
           m.proj_d.weight.device, float(m.factor_a), float(m.sigma_d))
    if cache is None or cache[0] != key:
        assert m.proj_d.weight.shape == (256, 256) and m.angle_k == 3, "kernel is built for hidden_dim=256, k=3"
        wdh, wdl = (_mfma_fragment_order(t) for t in _bf16_split(m.proj_d.weight.detach()))
        wah, wal = (_mfma_fragment_order(t) for t in _bf16_split(m.proj_a.weight.detach()))
        bias = (m.proj_d.bias.detach().float() + m.proj_a.bias.detach().float()).contiguous()
        cache = (key, wdh, wdl, wah, wal, bias, m.embedding.div_term.detach().float().contiguous())
        m._hip_cache = cache
    _, wdh, wdl, wah, wal, bias, div = cache
    out = torch.empty(B, n, n, 256, dtype=out_dtype, device=points.device)
    knn = torch.empty(B, n, 3, dtype=torch.int32, device=points.device)
    if st.GEO_TABLE if bf16_out else st.GEO_TABLE_F32:
        npoint = 4 if bf16_out else 6
        tab = _geo_tables(m, key, npoint)
        if tab is not None and not _GEO_TABLE_UNAVAILABLE.get(points.device, False):
            try:
                with on_device(points.device):
                    call("unopose_geo_embedding_table", ptr(points), B, n, ptr(tab[0]), tab[0].shape[0], ptr(tab[1]), tab[1].shape[0],
                         ptr(bias), ptr(tab[2]), ptr(div), _GEO_HINV, npoint, float(m.sigma_d), float(m.factor_a),
                         int(m.reduction_a == "mean"), int(bf16_out), ptr(knn), ptr(out), stream_ptr())
                return out
            except RuntimeError as e:
                if "cannot reserve" not in str(e):
                    raise
                # a device that cannot give the table kernel its ~145 KiB of LDS: the matrix-core kernel below takes over, loudly, for good
                _GEO_TABLE_UNAVAILABLE[points.device] = True
                note_fallback("geo_embedding", f"table kernel unavailable on {points.device} ({e}): matrix-core kernel")
    with on_device(points.device):
        call("unopose_geo_embedding", ptr(points), B, n, ptr(wdh), ptr(wdl), ptr(wah), ptr(wal), ptr(bias), ptr(div),
             float(m.sigma_d), float(m.factor_a), int(m.reduction_a == "mean"), int(not bf16_out), int(bf16_out),
             ptr(knn), ptr(out), stream_ptr())
    return out


def geo_embedding_torch(points, m):
    """Op-by-op torch composite of the same function (A/B reference for tests and profiling)."""
    points = points.float()
    B, N, _ = points.shape
    with torch.autocast("cuda", enabled=False):
        dist = torch.sqrt(pairwise_distance(points, points))
        k = m.angle_k
        knn = dist.topk(k=k + 1, dim=2, largest=False)[1][:, :, 1:]
        knn_pts = torch.gather(points.unsqueeze(1).expand(B, N, N, 3), 2, knn.unsqueeze(3).expand(B, N, k, 3))
        ref = (knn_pts - points.unsqueeze(2)).unsqueeze(2).expand(B, N, N, k, 3)
        anc = (points.unsqueeze(1) - points.unsqueeze(2)).unsqueeze(3).expand(B, N, N, k, 3)
        sin_v = torch.linalg.norm(torch.cross(ref, anc, dim=-1), dim=-1)
        cos_v = (ref * anc).sum(-1)
        a_idx = torch.atan2(sin_v, cos_v) * m.factor_a
        d_idx = dist / m.sigma_d
        div = m.embedding.div_term

        def sinus(idx):
            om = idx.unsqueeze(-1) * div
            return torch.stack([torch.sin(om), torch.cos(om)], dim=-1).reshape(*idx.shape, -1)

        sd, sa = sinus(d_idx), sinus(a_idx)
    d_emb = _lin(sd, m.proj_d)
    a_emb = _lin(sa, m.proj_a)
    a_emb = a_emb.max(dim=3)[0] if m.reduction_a == "max" else a_emb.mean(dim=3)
    return d_emb + a_emb


def pe_group_mlp_max(pts, radius, nsample, mlp, bf16x3=None, cand_in=None, want_cand=False, out_split=None):
    """QueryAndLRFGroup -> SharedMLP[6,32,64,128] -> max over neighbours (fine matcher PE, Fi:167-174)
    as ONE HIP kernel (csrc/pe.hip): neighbour lists, frames and all MLP activations stay on chip;
    (B,N,3) -> (B,N,128) fp32.  Matrix-core precision: exact fp32 MFMA by default; under autocast(bf16)
    (or bf16x3=True) bf16 MFMA with hi/lo-split operands (~2^-16 relative error, ~5x the fp32 MFMA rate).
    Neighbour-list hand-off (bf16x3 kernel only): `want_cand=True` also returns (lists (B,N,nsample) int32,
    counts (B,N) int32) of this pass; passing such a pair from a LARGER-radius pass over the same points as
    `cand_in` lets this pass test those candidates instead of scanning the cloud (same result).
    `out_split` = (buf (Btot,N,2W) bf16 viewed as the split layout of a W-wide fp32 row, first cloud b0, first channel c0):
    the 128 channels go straight into that operand of csrc/gemm_f32.hip (bf16x3 kernel only); returns buf."""
    if bf16x3 is None:  # the hi/lo-split matrix-core form is the fp32-class arithmetic of every other contraction of the fp32 path too
        bf16x3 = torch.is_autocast_enabled() or st.USE_F32X3
    if [tuple(l.conv.weight.shape[:2]) for l in mlp.layers()] != [(32, 6), (64, 32), (128, 64)] or nsample % 32:
        note_fallback("pe_group_mlp_max", f"MLP widths / nsample {nsample} (kernel: 6-32-64-128, nsample % 32 == 0)")
        return pe_group_mlp_max_unfused(pts, radius, nsample, mlp)  # other widths: grouping kernel + GEMMs
    pts = _c(pts.float())
    check_f32(pts, "pts")
    B, N, _ = pts.shape
    cache = getattr(mlp, "_hip_cache", None)
    key = _params_key(mlp)  # (conv weights, BatchNorm affine AND running statistics)
    if cache is None or cache[0] != key:
        with torch.no_grad():
            flat = []
            for l in mlp.layers():
                w, b = l.folded()
                flat += [w.float().contiguous(), b.float().contiguous()]
        assert [tuple(t.shape) for t in flat[::2]] == [(32, 6), (64, 32), (128, 64)], "kernel is built for [6,32,64,128]"
        from .._lib import lib
        image = torch.empty(lib().unopose_pe_image_bytes(), dtype=torch.uint8, device=pts.device)
        with on_device(pts.device):
            call("unopose_pe_pack_weights", *(ptr(t) for t in flat), ptr(image), stream_ptr())
        cache = (key, flat, image)
        mlp._hip_cache = cache
    w1, b1, w2, b2, w3, b3 = cache[1]
    cand_out = None
    if out_split is not None:
        buf, b0, c0 = out_split
        assert bf16x3 and buf.dtype == torch.bfloat16 and buf.is_contiguous() and buf.shape[1] == N and c0 % 32 == 0 and b0 + B <= buf.shape[0]
        ld = buf.shape[2] // 2  # row width in 4-byte units
        with on_device(pts.device):
            if want_cand:
                cand_out = (torch.empty(B, N, int(nsample), dtype=torch.int32, device=pts.device),
                            torch.empty(B, N, dtype=torch.int32, device=pts.device))
            ci = cand_in if cand_in is not None else (None, None)
            dst = ctypes.c_void_p(buf.data_ptr() + (b0 * N * ld + c0) * 4)
            call("unopose_pe_group_mlp_max_packed_out", ptr(pts), B, N, float(radius), int(nsample), ptr(cache[2]),
                 None if ci[0] is None else ptr(ci[0]), None if ci[0] is None else ptr(ci[1]),
                 0 if ci[0] is None else int(ci[0].shape[2]), None if cand_out is None else ptr(cand_out[0]),
                 None if cand_out is None else ptr(cand_out[1]), dst, ld, 1, stream_ptr())
        return (buf, cand_out) if want_cand else buf
    out = torch.empty(B, N, 128, dtype=torch.float32, device=pts.device)
    with on_device(pts.device):
        if bf16x3:
            if want_cand:
                cand_out = (torch.empty(B, N, int(nsample), dtype=torch.int32, device=pts.device),
                            torch.empty(B, N, dtype=torch.int32, device=pts.device))
            ci = cand_in if cand_in is not None else (None, None)
            assert ci[0] is None or (ci[0].is_contiguous() and ci[0].shape[:2] == (B, N) and ci[1].shape == (B, N))
            call("unopose_pe_group_mlp_max_packed_cand", ptr(pts), B, N, float(radius), int(nsample), ptr(cache[2]),
                 None if ci[0] is None else ptr(ci[0]), None if ci[0] is None else ptr(ci[1]),
                 0 if ci[0] is None else int(ci[0].shape[2]), None if cand_out is None else ptr(cand_out[0]),
                 None if cand_out is None else ptr(cand_out[1]), ptr(out), stream_ptr())
        else:
            call("unopose_pe_group_mlp_max", ptr(pts), B, N, float(radius), int(nsample), ptr(w1), ptr(b1), ptr(w2),
                 ptr(b2), ptr(w3), ptr(b3), 0, ptr(out), stream_ptr())
    return (out, cand_out) if want_cand else out


def pe_group_mlp_max_unfused(pts, radius, nsample, mlp, chunk=4):
    """Same function as pe_group_mlp_max through the materialised (B,6,N,S) features (HIP fused
    ball-query+group+LRF, then torch GEMMs with BN folded): A/B reference for tests and profiling."""
    outs = []
    folded = [l.folded() for l in mlp.layers()]
    with torch.autocast("cuda", enabled=False):
        for b0 in range(0, pts.shape[0], chunk):
            x = query_lrf_group(pts[b0:b0 + chunk], radius, nsample)  # (b,6,N,S)
            x = x.permute(0, 2, 3, 1)
            for w, b in folded:
                x = F.relu(F.linear(x, w.float(), b.float()))
            outs.append(x.max(dim=2)[0])
    return torch.cat(outs, 0)


def furthest_point_sample(pts, npoint):
    return _ext.furthest_point_sampling(_c(pts.float()), npoint)


def _geo_grid(m, rows, npoint):
    """sinus(x_r) on the table grid x_r = (r - (npoint / 2 - 1)) / 4, (rows, 256) float64: T = S W^T is the table, dW = dT^T S its gradient."""
    div = m.embedding.div_term
    key = (rows, npoint, div._version, div.data_ptr())
    cache = m.__dict__.setdefault("_geo_grid_cache", {})
    if key not in cache:  # (depends on the frequencies only: built once, not per training step)
        d64 = div.detach().double()
        x = (torch.arange(rows, device=div.device, dtype=torch.float64) - float(npoint // 2 - 1)) / _GEO_HINV
        om = x[:, None] * d64[None, :]
        cache.clear() if len(cache) > 8 else None
        cache[key] = torch.stack([torch.sin(om), torch.cos(om)], dim=-1).reshape(rows, -1)
    return cache[key]


class _GeoEmbedFn(torch.autograd.Function):
    """GeometricStructureEmbedding.forward (transformer.py:303-350) with gradients for proj_d / proj_a (the points carry none: they are
    data).  Forward = the 6-point table kernel of the fp32 eval path (tables rebuilt from the current weights), which also records the
    arg-max of the three angle terms; backward = its mirror, scattering dE into a table-shaped gradient in LDS (csrc/embed.hip
    geo_embed_table_bwd_kernel), then dW = dT^T S_grid and db = sum_r dT[r] on the host side (two 256-wide matmuls)."""

    @staticmethod
    def forward(ctx, points, wd, bd, wa, ba, m):
        import math

        npoint = 6
        B, n, _ = points.shape
        rows_a = int(math.floor(math.pi * float(m.factor_a) * _GEO_HINV)) + npoint + 1
        rows_d = _GEO_D_RANGE * _GEO_HINV + npoint
        sd, sa = _geo_grid(m, rows_d, npoint), _geo_grid(m, rows_a, npoint)
        td = (sd @ wd.detach().double().t()).float().contiguous()
        ta = (sa @ wa.detach().double().t()).float().contiguous()
        bias = (bd.detach().float() + ba.detach().float()).contiguous()
        wdf = wd.detach().float().contiguous()
        div = m.embedding.div_term.detach().float().contiguous()
        out = torch.empty(B, n, n, 256, dtype=torch.float32, device=points.device)
        amax = torch.empty(B, n, n, 64, dtype=torch.int32, device=points.device)
        knn = torch.empty(B, n, 3, dtype=torch.int32, device=points.device)
        mean = int(m.reduction_a == "mean")
        with on_device(points.device):
            call("unopose_geo_embedding_train_forward", ptr(points), B, n, ptr(td), rows_d, ptr(ta), rows_a, ptr(bias), ptr(wdf), ptr(div), _GEO_HINV,
                 float(m.sigma_d), float(m.factor_a), mean, ptr(knn), ptr(out), ptr(amax), stream_ptr())
        ctx.save_for_backward(points, knn, amax, sd, sa)
        ctx.meta = (rows_d, rows_a, float(m.sigma_d), float(m.factor_a), mean, npoint)
        return out

    @staticmethod
    def backward(ctx, dE):
        points, knn, amax, sd, sa = ctx.saved_tensors
        rows_d, rows_a, sigma_d, factor_a, mean, npoint = ctx.meta
        B, n, _ = points.shape
        rd_l = min(rows_d, 16 * _GEO_HINV + npoint - 1)
        dE = _c(dE.float())
        G = lib().unopose_geo_embedding_train_workgroups(B, n)
        ws = torch.empty(G, rd_l + rows_a, 256, dtype=torch.float32, device=dE.device)
        full = torch.zeros(rows_d, 256, dtype=torch.float32, device=dE.device)
        past = torch.zeros(1, dtype=torch.int32, device=dE.device)
        with on_device(dE.device):
            call("unopose_geo_embedding_train_backward", ptr(points), ptr(knn), B, n, rows_d, rows_a, _GEO_HINV, sigma_d, factor_a, mean, ptr(dE), ptr(amax),
                 ptr(ws), ptr(full), ptr(past), stream_ptr())
        torch._assert_async(past == 0, "geo embedding backward: a distance index past the table (the clouds are not radius-normalised)")
        dT = ws.double().sum(0)
        dTd = full.double()
        dTd[:rd_l] += dT[:rd_l]
        dTa = dT[rd_l:]
        dwd, dwa = (dTd.t() @ sd).float(), (dTa.t() @ sa).float()
        dbd, dba = dTd.sum(0).float(), dTa.sum(0).float()
        return None, dwd, dbd, dwa, dba, None


def geo_embedding_train_ok(points, m):
    return (st.TRAIN_OWN_GEO and points.is_cuda and m.proj_d.weight.shape == (256, 256) and m.angle_k == 3 and points.shape[1] >= 4
            and m.proj_d.weight.dtype == torch.float32 and not torch.is_autocast_enabled()
            and int(__import__("math").floor(__import__("math").pi * float(m.factor_a) * _GEO_HINV)) + 7 <= 80 - 6)
