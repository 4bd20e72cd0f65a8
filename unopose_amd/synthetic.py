"""Synthetic (query, reference) RGB-D pairs for smoke / bench / tests.

Both clouds are the SAME ellipsoid surface seen under two poses (relative rotation <= 50 deg), 0.5 mm
depth noise, every surface point tied to one pixel of one shared crop -- the geometry the matcher is
trained for, so nothing in the forward degenerates (no all-background assignments).  Shapes / dtypes
are the reference's input contract."""
import math

import torch


def random_rotation(gen, max_deg=50.0):
    axis = torch.randn(3, generator=gen)
    axis = axis / axis.norm()
    ang = math.radians(max_deg) * torch.rand((), generator=gen).item()
    K = torch.tensor([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return torch.eye(3) + math.sin(ang) * K + (1 - math.cos(ang)) * (K @ K)


def congruent_pair(gen, nq=2048, nt=5000, S=224, noise=0.0):
    """One pair (B=1 tensors) + ground truth (R, t) with p_query = R p_ref + t."""
    axes = 0.05 + 0.06 * torch.rand(3, generator=gen)
    v = torch.randn(nt, 3, generator=gen)
    v = v / v.norm(dim=1, keepdim=True)
    v[:, 2] = -v[:, 2].abs()
    obj = v * axes
    uv = ((obj[:, :2] / axes[:2]) * 0.48 + 0.5) * (S - 1)
    pix = (uv[:, 1].round().long().clamp(0, S - 1) * S + uv[:, 0].round().long().clamp(0, S - 1))
    Rq, Rr = random_rotation(gen, 25.0), random_rotation(gen, 25.0)
    tq = torch.tensor([0.02, -0.03, 0.8]) + 0.05 * torch.randn(3, generator=gen)
    tr = torch.tensor([-0.04, 0.01, 0.7]) + 0.05 * torch.randn(3, generator=gen)
    sel = torch.randperm(nt, generator=gen)[:nq]
    q = obj[sel] @ Rq.T + tq + noise * torch.randn(nq, 3, generator=gen)
    r = obj @ Rr.T + tr + noise * torch.randn(nt, 3, generator=gen)
    img = torch.randn(1, 3, S, S, generator=gen)
    ep = dict(pts=q[None].contiguous(), tem1_pts=r[None].contiguous(), rgb=img, tem1_rgb=img.clone(),
              rgb_choose=pix[sel][None].contiguous(), tem1_choose=pix[None].contiguous())
    R = Rq @ Rr.T
    t = tq - R @ tr
    return ep, R, t


def make_batch(B, nq=2048, nt=5000, S=224, seed=0, noise=5e-4, device="cpu"):
    """B independent pairs stacked into one end_points dict; also returns (R_gt (B,3,3), t_gt (B,3))."""
    gen = torch.Generator().manual_seed(seed)
    eps, Rs, ts = [], [], []
    for _ in range(B):
        ep, R, t = congruent_pair(gen, nq, nt, S, noise)
        eps.append(ep)
        Rs.append(R)
        ts.append(t)
    out = {k: torch.cat([e[k] for e in eps], 0).to(device) for k in eps[0]}
    return out, torch.stack(Rs).to(device), torch.stack(ts).to(device)


def make_shared_reference_batch(B, per_ref=8, nq=2048, nt=5000, S=224, seed=0, noise=5e-4, device="cpu"):
    """B pairs in groups of `per_ref` queries that look at the SAME reference view (the BOP one-reference test set pairs
    many query instances with few reference views: oneref_feature_extraction.py:252-263) -- every group is one
    ellipsoid + one reference pose, each of its queries an own pose and an own subset of the surface points.
    Returns (end_points, ref_keys (one hashable per pair), R_gt, t_gt)."""
    gen = torch.Generator().manual_seed(seed)
    eps, Rs, ts, keys = [], [], [], []
    for b in range(B):
        if b % per_ref == 0:
            ep0, _, _ = congruent_pair(gen, nq, nt, S, noise)
        # a fresh query of the same object: re-pose the reference cloud (noise-free surface = R_r^T (r - t_r) up to its noise)
        Rq = random_rotation(gen, 25.0)
        tq = torch.tensor([0.02, -0.03, 0.8]) + 0.05 * torch.randn(3, generator=gen)
        sel = torch.randperm(nt, generator=gen)[:nq]
        r = ep0["tem1_pts"][0]
        c = r.mean(0)
        q = (r[sel] - c) @ Rq.T + tq + noise * torch.randn(nq, 3, generator=gen)  # p_q = Rq (p_r - c) + tq
        ep = dict(ep0, pts=q[None].contiguous(), rgb_choose=ep0["tem1_choose"][:, sel].contiguous())
        eps.append(ep)
        Rs.append(Rq)
        ts.append(tq - Rq @ c)
        keys.append(("view", seed, b // per_ref))
    out = {k: torch.cat([e[k] for e in eps], 0).to(device) for k in eps[0]}
    return out, keys, torch.stack(Rs).to(device), torch.stack(ts).to(device)


def make_train_batch(B, nq=2048, nt=5000, S=224, seed=0, noise=5e-4, device="cpu"):
    """`make_batch` + the two labels a training item carries (`rotation_label` (B,3,3), `translation_label` (B,3):
    pfoneref_training_dataset_v2.py:560-590); the pose noise of the coarse stage is drawn inside the forward."""
    batch, R, t = make_batch(B, nq, nt, S, seed, noise, device)
    batch["rotation_label"], batch["translation_label"] = R, t
    return batch


def trained_like_(model, seed=0, tame=0.1):
    """In-place seeded init that behaves like a trained matcher on congruent pairs (there are no
    checkpoints in this environment): default torch init, BN running stats randomised, token-mixing
    projections scaled by `tame`, overlap heads biased to ~0.9, LayerScale at O(0.1-0.5)."""
    gen = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            leaf = name.rsplit(".", 1)[-1]
            if name.endswith("attention.linear.weight") or name.endswith("output.squeeze.weight") \
                    or name.endswith("PE.mlp3.conv.weight"):
                p.mul_(tame)
            elif "score_heads" in name:
                p.copy_(torch.full_like(p, 2.0) if leaf == "bias" else p * 0.2)
            elif leaf == "gamma":
                p.copy_((0.05 + 0.45 * torch.rand(p.shape, generator=gen)).to(p))
            elif leaf in ("cls_token", "reg_token", "pos_embed"):
                p.copy_((0.02 * torch.randn(p.shape, generator=gen)).to(p))
        for name, b in model.named_buffers():
            if name.endswith("running_var"):
                b.copy_((0.5 + torch.rand(b.shape, generator=gen)).to(b))
            elif name.endswith("running_mean"):
                b.copy_((0.1 * torch.randn(b.shape, generator=gen)).to(b))
    return model
