"""Operator-level API of the reference's ``core/unopose/model/pointnet2/pointnet2_utils.py`` on the HIP
``_ext``: same names, argument order and autograd behaviour (``pointnet2_utils.py:51-289``), so reference
code that does ``from ...pointnet2_utils import gather_operation, furthest_point_sample`` keeps working
after rebinding the import.

Index-producing ops are non-differentiable; gather / group / interpolate back-propagate through the
``*_grad`` kernels (atomic scatter-add).  ``QueryAndLRFGroup`` uses the fused ball-query+group+LRF kernel
when it is called the way UNOPose calls it (centres == points, ``use_xyz``), else the op-by-op route.
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from . import _ext
from .. import ops


class FurthestPointSampling(Function):
    @staticmethod
    def forward(ctx, xyz, npoint):
        idx = _ext.furthest_point_sampling(xyz, npoint)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None


furthest_point_sample = FurthestPointSampling.apply


class GatherOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        ctx.for_backwards = (idx, features.size(2))
        return _ext.gather_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        idx, n = ctx.for_backwards
        return _ext.gather_points_grad(grad_out.contiguous(), idx, n), None


gather_operation = GatherOperation.apply


class ThreeNN(Function):
    @staticmethod
    def forward(ctx, unknown, known):
        dist2, idx = _ext.three_nn(unknown, known)
        dist = torch.sqrt(dist2)
        ctx.mark_non_differentiable(dist, idx)
        return dist, idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    @staticmethod
    def forward(ctx, features, idx, weight):
        ctx.save_for_backward(idx, weight)
        ctx.m = features.size(2)
        return _ext.three_interpolate(features, idx, weight)

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        return _ext.three_interpolate_grad(grad_out.contiguous(), idx, weight, ctx.m), None, None


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        ctx.for_backwards = (idx, features.size(2))
        return _ext.group_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        idx, n = ctx.for_backwards
        return _ext.group_points_grad(grad_out.contiguous(), idx, n), None


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        # NB the python argument order differs from the native one (pointnet2_utils.py:280)
        inds = _ext.ball_query(new_xyz, xyz, radius, nsample)
        ctx.mark_non_differentiable(inds)
        return inds

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


ball_query = BallQuery.apply


def _resample_uniformly(idx, nsample):
    """The reference's ``sample_uniformly`` loop (pointnet2_utils.py:343-351, 536-544) for all rows at once, on the device:
    each neighbour list becomes [its distinct indices, ascending] + [picks drawn uniformly with replacement from them], and
    ``unique_cnt`` (B,npoint) counts the distinct ones.  The picks are random in the reference too (torch.randint), so
    only the distribution is reproducible, not the draw."""
    s, _ = idx.sort(dim=-1)
    first = torch.ones_like(s, dtype=torch.bool)
    first[..., 1:] = s[..., 1:] != s[..., :-1]
    cnt = first.sum(-1, keepdim=True)  # (B,P,1)
    order = torch.sort((~first).to(torch.int8), dim=-1, stable=True).indices  # distinct entries first, still ascending
    uniq = s.gather(-1, order)
    pick = (torch.rand(idx.shape, device=idx.device) * cnt).long().minimum(cnt - 1)
    slot = torch.arange(nsample, device=idx.device).expand_as(idx)
    out = torch.where(slot < cnt, uniq, uniq.gather(-1, pick))
    # the reference builds unique_cnt with torch.zeros(...) on the host (P:344); same dtype and device here
    return out.to(idx.dtype).contiguous(), cnt.squeeze(-1).float().cpu()


class QueryAndGroup(nn.Module):
    """pointnet2_utils.py:292-367."""

    def __init__(self, radius, nsample, use_xyz=True, ret_grouped_xyz=False, normalize_xyz=False,
                 sample_uniformly=False, ret_unique_cnt=False):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz
        self.ret_grouped_xyz, self.normalize_xyz = ret_grouped_xyz, normalize_xyz
        self.sample_uniformly, self.ret_unique_cnt = sample_uniformly, ret_unique_cnt
        if ret_unique_cnt:
            assert sample_uniformly

    def forward(self, xyz, new_xyz, features=None):
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        if self.sample_uniformly:
            idx, unique_cnt = _resample_uniformly(idx, self.nsample)
        grouped_xyz = grouping_operation(xyz.transpose(1, 2).contiguous(), idx)
        grouped_xyz = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
        if self.normalize_xyz:
            grouped_xyz = grouped_xyz / self.radius
        if features is not None:
            grouped_features = grouping_operation(features, idx)
            new_features = torch.cat([grouped_xyz, grouped_features], dim=1) if self.use_xyz else grouped_features
        else:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            new_features = grouped_xyz
        ret = [new_features]
        if self.ret_grouped_xyz:
            ret.append(grouped_xyz)
        if self.ret_unique_cnt:
            ret.append(unique_cnt)
        return ret[0] if len(ret) == 1 else tuple(ret)


class QueryAndLRFGroup(nn.Module):
    """pointnet2_utils.py:484-584.  Output channels [p_k - c (3), R^T (p_k - c) / radius (3)].  Called the way UNOPose calls it
    (centres are the points themselves, no re-sampling) it is ONE fused ball-query + group + frame kernel; any other
    configuration takes ball_query -> (sample_uniformly) -> the same frame kernel fed with the neighbour lists."""

    def __init__(self, radius, nsample, use_xyz=False, use_feature=False, ret_grouped_xyz=False,
                 normalize_xyz=False, sample_uniformly=False, ret_unique_cnt=False):
        super().__init__()
        self.radius, self.nsample, self.use_xyz, self.use_feature = radius, nsample, use_xyz, use_feature
        self.ret_grouped_xyz, self.normalize_xyz = ret_grouped_xyz, normalize_xyz
        self.sample_uniformly, self.ret_unique_cnt = sample_uniformly, ret_unique_cnt
        if ret_unique_cnt:
            assert sample_uniformly

    def forward(self, xyz, new_xyz, features=None):
        same = new_xyz is xyz or (new_xyz.shape == xyz.shape and new_xyz.data_ptr() == xyz.data_ptr()
                                  and new_xyz.stride() == xyz.stride())
        idx = unique_cnt = None
        if same and not self.sample_uniformly:
            fused = ops.query_lrf_group(xyz, self.radius, self.nsample)  # (B,6,N,S)
        else:
            idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
            if self.sample_uniformly:
                idx, unique_cnt = _resample_uniformly(idx, self.nsample)
            fused = ops.lrf_group_idx(xyz, new_xyz, idx, self.radius)
        grouped_xyz, lrf_features = fused[:, :3], fused[:, 3:]
        if self.normalize_xyz:
            grouped_xyz = grouped_xyz / self.radius
        if features is not None:
            if not self.use_xyz:
                new_features = lrf_features
            else:  # the kernel's own (B,6,N,S) layout is already [grouped_xyz, lrf]
                new_features = torch.cat([grouped_xyz, lrf_features], dim=1) if self.normalize_xyz else fused
            if self.use_feature:
                if idx is None:
                    idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
                new_features = torch.cat([grouping_operation(features, idx), new_features], dim=1)
        else:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            new_features = lrf_features
        ret = [new_features]
        if self.ret_grouped_xyz:
            ret.append(grouped_xyz)
        if self.ret_unique_cnt:
            ret.append(unique_cnt)
        return ret[0] if len(ret) == 1 else tuple(ret)
