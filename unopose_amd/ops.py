"""Fused geometry / attention / pose-head operators (C ABI part 2) on torch tensors.

Each function cites the reference Python it replaces.  Inputs must be CUDA
float32 tensors; there is no CPU path (RuntimeError).
"""
import torch

from ._lib import call, check_f32, ptr, stream_ptr


def _c(x):
    return x if x.is_contiguous() else x.contiguous()


def lrf_global(pts, use_ref_rad=False):
    """get_batch_lrf (oneref_grf_predator_pose_estimation_model.py:78-93). (B,N,3)->(B,N,3)."""
    pts = _c(pts.float())
    check_f32(pts, "pts")
    B, N, _ = pts.shape
    out = torch.empty_like(pts)
    with torch.cuda.device(pts.device):
        call("unopose_lrf_global", ptr(pts), B, N, int(bool(use_ref_rad)), ptr(out), stream_ptr())
    return out


def query_lrf_group(xyz, radius, nsample):
    """QueryAndLRFGroup(radius, nsample, use_xyz=True)(xyz, xyz, feats) (pointnet2_utils.py:522-584).
    (B,N,3) -> (B,6,N,nsample)."""
    xyz = _c(xyz.float())
    check_f32(xyz, "xyz")
    B, N, _ = xyz.shape
    out = torch.empty(B, 6, N, int(nsample), dtype=torch.float32, device=xyz.device)
    with torch.cuda.device(xyz.device):
        call("unopose_query_lrf_group", ptr(xyz), B, N, float(radius), int(nsample), ptr(out), stream_ptr())
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
    with torch.cuda.device(src.device):
        call("unopose_weighted_procrustes", ptr(src), ptr(ref), ptr(weights) if weights is not None else None, M, N,
             float(weight_thresh), float(eps), ptr(R), ptr(t), stream_ptr())
    return R, t
