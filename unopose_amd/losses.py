"""Training losses and pose-noise augmentation of the matchers.

Plain torch (any device, autograd-recorded): these are O(B n^2) reductions over tensors the matchers already hold,
evaluated once per step.  Contract = core/unopose/utils/loss_utils.py:111-203 (`get_weighted_bce_loss`,
`compute_overlap_loss`), :265-274 (`process_loss`) and core/unopose/utils/model_utils.py:285-333 (`aug_pose_noise`):
same end_points keys, same values (tests/golden/make_train_golden.py captures them from the reference)."""
import math

import numpy as np
import torch
import torch.nn.functional as F


FUSED_LABELS = True  # A/B attribute: False = the (B, n1, n2) distance matrix through torch on the GPU as well


def _pairwise_sq_dist(x, y):
    """|x_i - y_j|^2 in the reference's expansion (model_utils.py:230-257) -- the thresholds below sit on these values."""
    xy = x @ y.transpose(-1, -2)
    return ((x ** 2).sum(-1).unsqueeze(-1) - 2 * xy + (y ** 2).sum(-1).unsqueeze(-2)).clamp(min=0.0)


def _infonce(atten, label1, label2):
    """0.5 (CE over the columns for every query row + CE over the rows for every reference column), background = class 0
    (loss_utils.py:181-187).  On a HIP device: unopose_amd.ops.infonce_two_way (streaming statistics + one gradient pass)."""
    if atten.is_cuda:
        from . import ops

        return ops.infonce_two_way(atten, label1, label2)
    a = atten.float()
    l1 = F.cross_entropy(a.transpose(1, 2)[:, :, 1:], label1, reduction="none").mean(1)  # classes = columns, per query row
    l2 = F.cross_entropy(a[:, :, 1:], label2, reduction="none").mean(1)
    return 0.5 * (l1 + l2)


def weighted_bce(pred, target):
    """Class-balanced binary cross entropy per sample: positives weighted by the negative fraction and vice versa
    (loss_utils.py:111-129).  pred, target (B, n) in [0, 1] -> (B,)."""
    pos_frac = target.sum(1, keepdim=True) / target.size(1)
    weight = torch.where(target >= 0.5, 1 - pos_frac, pos_frac)
    return (weight * F.binary_cross_entropy(pred, target, reduction="none")).mean(1)


def overlap_losses(end_points, atten_list, score_list, saliency_list, pts1, pts2, gt_R, gt_t, predator_thres=0.15,
                   dis_thres=0.15, prefix="coarse"):
    """Per-block overlap-score, saliency and correspondence (InfoNCE) losses plus the monitoring scalars of the last
    block (loss_utils.py:132-203).  pts1 = query (tgt), pts2 = reference (src); (gt_R, gt_t) maps reference to query.
    Adds `<prefix>_score_loss<i>`, `_saliency_loss<i>`, `_atten_loss<i>`, `_acc`, `_fg_num`, `_dis`, each (B,)."""
    with torch.autocast(pts1.device.type, enabled=False):  # fp32 island: BCE on probabilities is not autocast-safe
        return _overlap_losses(end_points, [a.float() for a in atten_list], [s.float() for s in score_list],
                               [s.float() for s in saliency_list], pts1.float(), pts2.float(), gt_R.float(), gt_t.float(),
                               predator_thres, dis_thres, prefix)


def _overlap_losses(end_points, atten_list, score_list, saliency_list, pts1, pts2, gt_R, gt_t, predator_thres, dis_thres, prefix):
    n1 = pts1.shape[1]
    in_ref = (pts1 - gt_t.unsqueeze(1)) @ gt_R  # query points expressed in the reference frame
    fused = in_ref.is_cuda and FUSED_LABELS
    if fused:
        # csrc/glue.hip: nearest partner and "any partner close" per point, both directions, without the (B, n1, n2) matrix
        from . import ops

        d1, nn1, any1 = ops.nearest_partner(in_ref, pts2, predator_thres, over_b=True)
        d2, nn2, any2 = ops.nearest_partner(in_ref, pts2, predator_thres, over_b=False)
        overlap = torch.cat([any1, any2], 1).to(score_list[0].dtype)
    else:
        dist = torch.sqrt(_pairwise_sq_dist(in_ref, pts2))  # (B, n1, n2)
        close = dist <= predator_thres
        overlap = torch.cat([close.any(2), close.any(1)], 1).to(score_list[0].dtype)  # a point overlaps if ANY partner is close
    for i, score in enumerate(score_list):
        end_points[f"{prefix}_score_loss{i}"] = weighted_bce(score.float(), overlap)
    for i, sal in enumerate(saliency_list):
        end_points[f"{prefix}_saliency_loss{i}"] = weighted_bce(sal.float(), overlap)
    # nearest partner as the class label, 0 = background (no partner within dis_thres)
    if not fused:
        d1, nn1 = dist.min(2)
        d2, nn2 = dist.min(1)
    label1 = torch.where(d1 <= dis_thres, nn1 + 1, torch.zeros_like(nn1))
    label2 = torch.where(d2 <= dis_thres, nn2 + 1, torch.zeros_like(nn2))
    for i, atten in enumerate(atten_list):
        end_points[f"{prefix}_atten_loss{i}"] = _infonce(atten, label1, label2)
    pred = atten_list[-1][:, 1:, :].max(dim=2)[1]
    fg = (pred > 0).float()
    end_points[f"{prefix}_acc"] = (pred == label1).float().mean(1)
    end_points[f"{prefix}_fg_num"] = fg.sum(1)
    partner = torch.gather(pts2, 1, (fg * (pred - 1)).long().unsqueeze(2).expand(-1, -1, 3))
    end_points[f"{prefix}_dis"] = (torch.norm(partner - in_ref, dim=2) * fg).sum(1) / (fg.sum(1) + 1e-8)
    return end_points


def process_loss(end_points):
    """Scalar view of the step (loss_utils.py:265-274): every `coarse_*` / `fine_*` entry averaged over the batch,
    `loss` = mean over the batch of min(sum of all *loss* entries, 100)."""
    out, total = {}, 0
    for k, v in end_points.items():
        if "coarse_" in k or "fine_" in k:
            out[k] = v.mean()
            if "loss" in k:
                total = total + v
    out["loss"] = torch.clamp(total, max=100.0).mean()
    return out


def aug_pose_noise(gt_R, gt_t, std_rots=(15, 10, 5, 1.25, 1), max_rot=45, sel_std_trans=(0.2, 0.2, 0.2), max_trans=0.8):
    """Ground-truth pose perturbed by clamped Gaussian Euler angles (one std picked per call from `std_rots`) and a
    clamped Gaussian translation; z kept positive (model_utils.py:285-333).  Random-stream use mirrors the reference:
    np.random for the std, the CPU generator for the angles, the pose's device generator for the translation."""
    B, dev = gt_R.size(0), gt_R.device
    std = np.random.choice(list(std_rots))
    ang = torch.normal(mean=0, std=std, size=(B, 3)).to(dev).clamp(min=-max_rot, max=max_rot) * (math.pi / 180.0)
    c, s = torch.cos(ang), torch.sin(ang)
    one, zero = torch.ones(B, device=dev, dtype=gt_R.dtype), torch.zeros(B, device=dev, dtype=gt_R.dtype)

    def mat(rows):
        return torch.stack([torch.stack(r, 1) for r in rows], 1)

    rz = mat([[c[:, 0], -s[:, 0], zero], [s[:, 0], c[:, 0], zero], [zero, zero, one]])
    rx = mat([[one, zero, zero], [zero, c[:, 1], -s[:, 1]], [zero, s[:, 1], c[:, 1]]])
    ry = mat([[c[:, 2], zero, s[:, 2]], [zero, one, zero], [-s[:, 2], zero, c[:, 2]]])
    noise_t = torch.normal(mean=torch.zeros(B, 3, device=dev), std=torch.tensor(list(sel_std_trans), device=dev).view(1, 3))
    t = gt_t + noise_t.clamp(min=-max_trans, max=max_trans)
    t[:, 2] = t[:, 2].clamp(min=1e-6)
    return (gt_R @ (rz @ rx @ ry)).detach(), t.detach()
