"""Matching heads: feature similarity, soft assignment, coarse hypothesis search and fine pose (csrc/posehead.hip, fineassign.hip, glue.hip) with their
torch composites."""
import torch
import torch.nn.functional as F

from .._lib import call, check_f32, on_device, ptr, stream_ptr
from . import _state as st
from .common import _c, _own_f32, _own_glue, note_mutation
from .geometry import pairwise_distance, weighted_procrustes
from .dense import bmm_nt_f32


def overlap_scores(scores, n1, halves=False):
    """clamp(sigmoid(.)) of the score heads' outputs (B, n_tot, 1) without the two background tokens -> (B, n_tot - 2) fp32.
    `halves`: scores is (2B, n1 + 1, 1), the head's output over both clouds as one batch of 2B (cloud 2 of pair b = batch B + b)."""
    if halves:
        B = scores.shape[0] // 2
        if not (_own_glue(scores) and scores.dtype in (torch.float32, torch.bfloat16)):
            return overlap_scores(torch.cat((scores[:B], scores[B:]), dim=1), n1)
        sc = _c(scores)
        out = torch.empty(B, 2 * n1, dtype=torch.float32, device=sc.device)
        with on_device(sc.device):
            call("unopose_overlap_scores", ptr(sc), int(sc.dtype == torch.bfloat16), B, 2 * (n1 + 1), n1, 1, ptr(out), stream_ptr())
        return out
    if not (_own_glue(scores) and scores.dtype in (torch.float32, torch.bfloat16)):
        s1, s2 = scores[:, 1:(n1 + 1)], scores[:, (n1 + 2):]
        return torch.clamp(torch.sigmoid(torch.cat((s1, s2), dim=1).squeeze(-1).float()), 0, 1)
    sc = _c(scores)
    B, n_tot = sc.shape[0], sc.shape[1]
    out = torch.empty(B, n_tot - 2, dtype=torch.float32, device=sc.device)
    with on_device(sc.device):
        call("unopose_overlap_scores", ptr(sc), int(sc.dtype == torch.bfloat16), B, n_tot, n1, 0, ptr(out), stream_ptr())
    return out


def set_first_rows_(x, row):
    """x[:, 0, :] = row for x (B, n, C) contiguous and ONE row (C,) of x's dtype (the background token in front of every pair), in place, one
    launch (csrc/glue.hip copy_rows)."""
    B, n, C = x.shape
    note_mutation()
    if not (_own_glue(x) and x.is_contiguous() and row.is_contiguous() and row.dtype == x.dtype and row.numel() == C and (C * x.element_size()) % 4 == 0 and B <= 65535):
        x[:, 0, :] = row.reshape(1, C).to(x.dtype)
        return x
    with on_device(x.device):
        call("unopose_copy_rows", ptr(row), 0, ptr(x), n * C * x.element_size(), B, C * x.element_size(), stream_ptr())
    return x


def pose_score(dis, w, thr):
    """sum [dis < thr] w / (sum w + 1e-8) * mean w per batch row (model_utils.py:559-566)"""
    if not _own_glue(dis):
        return ((dis < thr).float() * w).sum(1) / (w.sum(1) + 1e-8) * w.mean(1)
    d, ww = _c(dis.float()), _c(w.float())
    out = torch.empty(d.shape[0], dtype=torch.float32, device=d.device)
    with on_device(d.device):
        call("unopose_pose_score", ptr(d), ptr(ww), d.shape[0], d.shape[1], float(thr), ptr(out), stream_ptr())
    return out


def rigid_rows(p, t, R):
    """(p - t) @ R for row-vector points p (B,N,3), t (B,3), R (B,3,3) (Fi:69).  Under autocast the reference's `@` is a
    bf16 bmm (operands rounded to bf16, fp32 accumulation, bf16 result); here the same arithmetic as three broadcast
    multiply-adds, so that no library bf16 GEMM kernel is on the path (`own_gemm_ok`)."""
    if st.HIP_GEMM_ALL and torch.is_autocast_enabled() and _own_glue(p) and p.dtype == torch.float32 and p.dim() == 3:
        pc, out = _c(p), torch.empty(p.shape, dtype=torch.bfloat16, device=p.device)
        with on_device(p.device):
            call("unopose_rigid_rows_bf16", ptr(pc), pc.shape[0], pc.shape[1], ptr(_c(t.float())), ptr(_c(R.float())), ptr(out), stream_ptr())
        return out
    x = p - t.unsqueeze(1)
    if _own_f32(p) and not torch.is_autocast_enabled() and x.dtype == torch.float32:
        return bmm_nt_f32(x, R.float().transpose(1, 2))  # (x @ R)[n, j] = sum_k x[n, k] R[k, j]
    if st._DIFF or not (st.HIP_GEMM_ALL and p.is_cuda and torch.is_autocast_enabled()):
        return x @ R
    with torch.autocast("cuda", enabled=False):
        bf = torch.bfloat16
        xb, Rb = x.to(bf).float(), R.to(bf).float()
        y = xb[..., 0:1] * Rb[:, None, 0, :] + xb[..., 1:2] * Rb[:, None, 1, :] + xb[..., 2:3] * Rb[:, None, 2, :]
        return y.to(bf)


def feature_similarity(f1, f2, temp):
    """compute_feature_similarity, cosine + normalize (model_utils.py:260-282).
    Under autocast on the GPU the bf16 GEMM writes its fp32 accumulators straight out (`out_dtype`) with
    1/temp folded into the (small) left operand: one 4-byte write of the (B,N1,N2) matrix instead of a
    bf16 write, a division pass and the fp32 cast the pose heads ask for (1.6 GB per step at B=32)."""
    if st._DIFF:  # the reference's expression, dtype and all (model_utils.py:260-282)
        return F.normalize(f1, p=2, dim=2) @ F.normalize(f2, p=2, dim=2).transpose(1, 2) / temp
    if f1.is_cuda and torch.is_autocast_enabled() and st.HIP_GEMM_ALL and _own_f32(f1):
        # no library bf16 GEMM on the path (own_gemm_ok): the bf16-rounded normalised operands, multiplied by the exact-fp32 MFMA
        # kernel (exact products, fp32 sums -- what the bf16 bmm with fp32 output computes up to summation order)
        with torch.autocast("cuda", enabled=False):
            return bmm_nt_f32(normalize_rows_bf16(f1, temp, as_f32=True), normalize_rows_bf16(f2, 1.0, as_f32=True))
    if (_own_f32(f1) and not torch.is_autocast_enabled() and f1.dtype == torch.float32 and f2.dtype == torch.float32 and f1.shape[-1] == 256
            and f2.shape[-1] == 256):
        # fp32: both normalisations in one pass each (csrc/glue.hip, unrounded), 1 / temp in the product's epilogue
        return bmm_nt_f32(normalize_rows_f32(f1), normalize_rows_f32(f2), alpha=1.0 / temp)
    a, b = F.normalize(f1.float(), p=2, dim=2), F.normalize(f2.float(), p=2, dim=2)
    if f1.is_cuda and torch.is_autocast_enabled() and not st.HIP_GEMM_ALL:
        with torch.autocast("cuda", enabled=False):
            return torch.bmm((a / temp).to(torch.bfloat16), b.to(torch.bfloat16).transpose(1, 2), out_dtype=torch.float32)
    if f1.is_cuda and torch.is_autocast_enabled():
        with torch.autocast("cuda", enabled=False):
            return torch.bmm((a / temp).to(torch.bfloat16).float(), b.to(torch.bfloat16).float().transpose(1, 2))
    if _own_f32(f1) and a.dtype == torch.float32:
        return bmm_nt_f32(_c(a), _c(b), alpha=1.0 / temp)  # (1 / temp in the kernel's epilogue: no division pass over the (B, N1, N2) matrix -- 0.5 GB at 2049 x 2049)
    return (a @ b.transpose(1, 2)) / temp


def soft_assignment(atten, score1, score2):
    """Mutual softmax x overlap scores + bg-aware labels (model_utils.py:434-446, 538-547)."""
    B = atten.shape[0]
    one = torch.ones(B, 1, device=atten.device)
    s1 = torch.cat((one, score1), 1)[:, :, None]
    s2 = torch.cat((one, score2), 1)[:, None, :]
    a = torch.softmax(atten, dim=2) * torch.softmax(atten, dim=1) * s1 * s2
    label1 = a[:, 1:, :].max(dim=2)[1]
    label2 = a[:, :, 1:].max(dim=1)[1]
    return a, label1, label2


def coarse_pose_torch(atten, score, pts1, pts2, rand, n1p=6000, n2p=300):
    """compute_coarse_Rt_overlap (model_utils.py:411-490); `rand` (B,3*n1p) is the uniform draw the
    reference makes inside forward (:462).  [torch composite + HIP 3-point Procrustes]"""
    B, N1, _ = pts1.shape
    N2 = pts2.shape[1]
    atten, pts1, pts2 = atten.float(), pts1.float(), pts2.float()
    a, l1, l2 = soft_assignment(atten, score[:, :N1].float(), score[:, N2:].float())
    w1, w2 = (l1 > 0).float(), (l2 > 0).float()
    ps = (a[:, 1:, 1:] * w1.unsqueeze(2) * w2.unsqueeze(1)).reshape(B, N1 * N2) ** 1.5
    # torch's CPU cumsum accumulates float32 input in double; mirror that so searchsorted agrees
    cs = torch.cumsum(ps.double(), dim=1).float()
    cs = cs / (cs[:, -1].unsqueeze(1) + 1e-8)
    idx = torch.searchsorted(cs, rand.contiguous())
    i1 = torch.clamp(idx.div(N2, rounding_mode="floor"), max=N1 - 1)
    i2 = torch.clamp(idx % N2, max=N2 - 1)
    p1 = torch.gather(pts1, 1, i1.unsqueeze(2).expand(-1, -1, 3)).reshape(B * n1p, 3, 3)
    p2 = torch.gather(pts2, 1, i2.unsqueeze(2).expand(-1, -1, 3)).reshape(B * n1p, 3, 3)
    rs, ts = weighted_procrustes(p2, p1, None, 0.5)
    rs, ts = rs.reshape(B, n1p, 3, 3), ts.reshape(B, n1p, 1, 3)
    p1, p2 = p1.reshape(B, n1p, 3, 3), p2.reshape(B, n1p, 3, 3)
    dis = torch.norm((p1 - ts) @ rs - p2, dim=3).mean(2)
    top = torch.topk(dis, n2p, dim=1, largest=False)[1]
    rs2 = torch.gather(rs, 1, top.reshape(B, n2p, 1, 1).expand(-1, -1, 3, 3))
    ts2 = torch.gather(ts, 1, top.reshape(B, n2p, 1, 1).expand(-1, -1, 1, 3))
    tp = (pts1.unsqueeze(1) - ts2) @ rs2  # (B,n2p,N1,3)
    d = torch.sqrt(pairwise_distance(tp, pts2.unsqueeze(1))).min(3)[0]  # (B,n2p,N1)
    sc = w1.unsqueeze(1).sum(2) / ((d * w1.unsqueeze(1)).sum(2) + 1e-8)
    pose_score, best = sc.max(1)
    R = torch.gather(rs2, 1, best.reshape(B, 1, 1, 1).expand(-1, -1, 3, 3)).squeeze(1)
    t = torch.gather(ts2, 1, best.reshape(B, 1, 1, 1).expand(-1, -1, 1, 3)).squeeze(2).squeeze(1)
    return R, t, pose_score


def fine_pose_torch(atten, score, pts1, pts2, dis_thres=0.15):
    """compute_fine_Rt_overlap (model_utils.py:527-566).  [torch composite + HIP weighted Procrustes]"""
    atten, pts1, pts2 = atten.float(), pts1.float(), pts2.float()
    N1 = pts1.shape[1]
    a, l1, l2 = soft_assignment(atten, score[:, :N1].float(), score[:, N1:].float())
    a = a[:, 1:, 1:] * (l1 > 0).float().unsqueeze(2) * (l2 > 0).float().unsqueeze(1)
    rows = a.sum(2)
    pred = (a / (rows.unsqueeze(2) + 1e-6)) @ pts2
    R, t = weighted_procrustes(pred, pts1, rows, 0.001)
    pp = (pts1 - t.unsqueeze(1)) @ R
    dis = torch.sqrt(pairwise_distance(pp, pts2)).min(2)[0]
    mask = (l1 > 0).float()
    ps = ((dis < dis_thres).float() * mask).sum(1) / (mask.sum(1) + 1e-8)
    return R, t, ps * mask.mean(1)


def _assign_labels(atten, score1, score2):
    B, R, C = atten.shape
    dev = atten.device
    stats = torch.empty(2 * B * (R + C), dtype=torch.float32, device=dev)
    w1 = torch.empty(B, R - 1, dtype=torch.float32, device=dev)
    w2 = torch.empty(B, C - 1, dtype=torch.float32, device=dev)
    call("unopose_assign_labels", ptr(atten), B, R, C, ptr(score1), ptr(score2), ptr(stats), ptr(w1), ptr(w2),
         stream_ptr())
    return stats, w1, w2


def coarse_pose(atten, score, pts1, pts2, rand, n1p=6000, n2p=300):
    """compute_coarse_Rt_overlap (model_utils.py:411-490) on HIP kernels (csrc/posehead.hip): streaming
    assignment statistics, CDF + searchsorted + 3-point Procrustes + residual per hypothesis, top-k of the residuals (rank by
    counting), candidate scoring, pick of the best candidate (`USE_OWN_TOPK`; off: torch.topk / max / gather)."""
    B, N1, _ = pts1.shape
    N2 = pts2.shape[1]
    atten, pts1, pts2 = _c(atten.float()), _c(pts1.float()), _c(pts2.float())
    check_f32(atten, "atten")
    score1, score2 = _c(score[:, :N1].float()), _c(score[:, N2:].float())  # NB `N2:` (model_utils.py:440)
    rand = _c(rand.float())
    dev = atten.device
    with on_device(dev):
        stats, w1, w2 = _assign_labels(atten, score1, score2)
        cdf = torch.empty(B, N1 * N2, dtype=torch.float32, device=dev)
        rs = torch.empty(B, n1p, 3, 3, dtype=torch.float32, device=dev)
        ts = torch.empty(B, n1p, 3, dtype=torch.float32, device=dev)
        dis = torch.empty(B, n1p, dtype=torch.float32, device=dev)
        call("unopose_coarse_hypotheses", ptr(atten), B, N1 + 1, N2 + 1, ptr(score1), ptr(score2), ptr(stats),
             ptr(w1), ptr(w2), ptr(rand), n1p, ptr(pts1), ptr(pts2), ptr(cdf), ptr(rs), ptr(ts), ptr(dis),
             stream_ptr())
        own = st.USE_OWN_TOPK and n1p <= 16384 and n2p <= n1p
        if own:  # the n2p smallest residuals, ascending, ties by index (csrc/posehead.hip: rank by counting)
            top = torch.empty(B, n2p, dtype=torch.int64, device=dev)
            call("unopose_topk_smallest", ptr(dis), B, n1p, n2p, ptr(top), stream_ptr())
        else:
            top = torch.topk(dis, n2p, dim=1, largest=False)[1].contiguous()
        sc = torch.empty(B, n2p, dtype=torch.float32, device=dev)
        call("unopose_coarse_scores", ptr(pts1), ptr(pts2), B, N1, N2, ptr(rs), ptr(ts), n1p, ptr(top), n2p, ptr(w1),
             ptr(sc), stream_ptr())
        if own:  # first maximum of the scores, its hypothesis, that hypothesis's R and t: one launch
            R = torch.empty(B, 3, 3, dtype=torch.float32, device=dev)
            t = torch.empty(B, 3, dtype=torch.float32, device=dev)
            pose_score = torch.empty(B, dtype=torch.float32, device=dev)
            call("unopose_coarse_pick", ptr(sc), ptr(top), B, n2p, ptr(rs), ptr(ts), n1p, ptr(R), ptr(t), ptr(pose_score), stream_ptr())
            return R, t, pose_score
    pose_score, best = sc.max(1)
    hyp = torch.gather(top, 1, best.unsqueeze(1))  # (B,1)
    R = torch.gather(rs, 1, hyp.reshape(B, 1, 1, 1).expand(-1, -1, 3, 3)).squeeze(1)
    t = torch.gather(ts, 1, hyp.reshape(B, 1, 1).expand(-1, -1, 3)).squeeze(1)
    return R, t, pose_score


def fine_pose(atten, score, pts1, pts2, dis_thres=0.15):
    """compute_fine_Rt_overlap (model_utils.py:527-566) on HIP kernels: five streaming passes over the
    (B,N1+1,N2+1) similarity that write only O(N) statistics, weighted Procrustes (Jacobi), min-distance
    verification."""
    B, N1, _ = pts1.shape
    N2 = pts2.shape[1]
    atten, pts1, pts2 = _c(atten.float()), _c(pts1.float()), _c(pts2.float())
    check_f32(atten, "atten")
    score1, score2 = _c(score[:, :N1].float()), _c(score[:, N1:].float())
    dev = atten.device
    with on_device(dev):
        stats, w1, w2 = _assign_labels(atten, score1, score2)
        weight = torch.empty(B, N1, dtype=torch.float32, device=dev)
        pred = torch.empty(B, N1, 3, dtype=torch.float32, device=dev)
        call("unopose_fine_correspondences", ptr(atten), B, N1 + 1, N2 + 1, ptr(score1), ptr(score2), ptr(stats),
             ptr(w1), ptr(w2), ptr(pts2), ptr(weight), ptr(pred), stream_ptr())
        R, t = weighted_procrustes(pred, pts1, weight, 0.001)
        dis = torch.empty(B, N1, dtype=torch.float32, device=dev)
        call("unopose_min_dist", ptr(pts1), ptr(pts2), B, N1, N2, ptr(R), ptr(t), 1, ptr(dis), stream_ptr())
    return R, t, pose_score(dis, w1, dis_thres)


def fine_pose_fused_ok(f1, f2):
    """True when `fine_pose_from_features` may stand in for feature_similarity + fine_pose: HIP device, autocast (the
    bf16 product the reference's autocast matmul makes), inference, 256-wide features."""
    return (st.USE_FUSED_FINE and f1.is_cuda and torch.is_autocast_enabled() and not st._DIFF and f1.shape[-1] == 256
            and f2.shape[-1] == 256)


def normalize_rows_bf16(f, temp, as_f32=False):
    """bf16(F.normalize(f.float(), dim=-1) / temp) in one pass (csrc/glue.hip); f (...,256) bf16 or fp32.  `as_f32`: the same
    bf16-rounded values in an fp32 tensor (the operand type of `bmm_nt_f32`)."""
    if f.shape[-1] != 256 or f.dtype not in (torch.bfloat16, torch.float32):
        y = _c((F.normalize(f.float(), p=2, dim=-1) / temp).to(torch.bfloat16))
        return y.float() if as_f32 else y
    fc = _c(f)
    out = torch.empty(f.shape, dtype=torch.float32 if as_f32 else torch.bfloat16, device=f.device)
    with on_device(f.device):
        call("unopose_normalize_rows_bf16", ptr(fc), int(f.dtype == torch.bfloat16), fc.numel() // 256, 256, float(temp), ptr(out), int(as_f32),
             stream_ptr())
    return out


def normalize_rows_f32(f):
    """F.normalize(f, p=2, dim=-1) for fp32 f (...,256) in one pass (csrc/glue.hip, no rounding)."""
    fc = _c(f)
    out = torch.empty_like(fc)
    with on_device(f.device):
        call("unopose_normalize_rows_bf16", ptr(fc), 0, fc.numel() // 256, 256, 1.0, ptr(out), 2, stream_ptr())
    return out


def fine_pose_from_features(f1, f2, temp, score, pts1, pts2, dis_thres=0.15):
    """compute_feature_similarity (cosine, /temp; model_utils.py:260-282) + compute_fine_Rt_overlap (:527-566) with the
    similarity recomputed tile by tile inside the three reduction passes instead of stored: f1 (B,N1+1,256), f2
    (B,N2+1,256) are the out_proj features (row 0 = background token).  Same bf16 operands / fp32 accumulation as the
    autocast bmm of `feature_similarity`."""
    B, N1, _ = pts1.shape
    N2 = pts2.shape[1]
    assert f1.shape == (B, N1 + 1, 256) and f2.shape == (B, N2 + 1, 256)
    a, b = normalize_rows_bf16(f1, temp), normalize_rows_bf16(f2, 1.0)
    pts1, pts2 = _c(pts1.float()), _c(pts2.float())
    score1, score2 = _c(score[:, :N1].float()), _c(score[:, N1:].float())
    dev = pts1.device
    R, C = N1 + 1, N2 + 1
    with on_device(dev):
        ws = torch.empty(B * (R + C) + B * (-(-N1 // 256) - (-N2 // 256)), dtype=torch.float32, device=dev)
        w1 = torch.empty(B, N1, dtype=torch.float32, device=dev)
        w2 = torch.empty(B, N2, dtype=torch.float32, device=dev)
        weight = torch.empty(B, N1, dtype=torch.float32, device=dev)
        pred = torch.empty(B, N1, 3, dtype=torch.float32, device=dev)
        call("unopose_fine_assign", ptr(a), ptr(b), B, R, C, 256, 1.0 / temp, ptr(score1), ptr(score2), ptr(pts2), ptr(ws),
             ptr(w1), ptr(w2), ptr(weight), ptr(pred), stream_ptr())
        Rm, t = weighted_procrustes(pred, pts1, weight, 0.001)
        dis = torch.empty(B, N1, dtype=torch.float32, device=dev)
        call("unopose_min_dist", ptr(pts1), ptr(pts2), B, N1, N2, ptr(Rm), ptr(t), 1, ptr(dis), stream_ptr())
    return Rm, t, pose_score(dis, w1, dis_thres)
