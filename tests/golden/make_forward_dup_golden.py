"""End-to-end fixture from the REFERENCE's own forward on a DUPLICATE-HEAVY query cloud: the provider's small-mask
branch (pfoneref_bop_test_dataset_v2.py:200-203) samples the observed points WITH replacement when the mask holds fewer pixels than
n_sample_observed_point -- here 1024 query points drawn from 205 distinct ones (a 20 % mask).  On such clouds only ~64 % of the points have
a well-conditioned local frame (tests/test_parity_prod_gpu.py), the per-point comparisons of PE / fine tokens are strict only there; this
fixture closes the gap at the level that matters: the POSE of the whole forward.  Same recipe as make_forward_diffimg_golden.py (reference
modules imported with the App-G stubs, tamed weights, the coarse stage's uniform draw injected); oracle == reference asserted before
anything is written.
    python tests/golden/make_forward_dup_golden.py   ->  tests/golden/forward_dup.npz"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (sets sys.path for the repo root and tests/)


def main():
    MG.import_reference()
    from oracle import unopose_ref as R
    from oracle.pointnet2_oracle import ext
    from core.unopose.model.oneref_grf_predator_pose_estimation_model import UNOPose
    from helpers import congruent_pair

    torch.set_grad_enabled(False)
    cfg = R.default_cfg()
    sdt = R.random_state_dict(cfg, seed=0, tame=0.1)
    nq, nt = 1024, 2500
    cfg_ref = R.default_cfg(fine_npoint=nq, feature_extraction=dict(freeze_vit=False))
    model = UNOPose(cfg_ref).eval()
    model.load_state_dict(sdt, strict=True)
    gg = torch.Generator().manual_seed(123)
    end_points, R_gt, t_gt = congruent_pair(gg, nq=nq, nt=nt, noise=1e-3)
    # the small-mask branch: the mask holds 205 pixels, the provider draws 1024 of them with replacement
    distinct = nq // 5
    keep = torch.randperm(nq, generator=gg)[:distinct]
    idx = keep[torch.randint(0, distinct, (nq,), generator=gg)]
    end_points["pts"] = end_points["pts"][:, idx].contiguous()
    end_points["rgb_choose"] = end_points["rgb_choose"][:, idx].contiguous()
    n_unique = len(torch.unique(idx))
    print("distinct query points:", n_unique, "of", nq)
    assert n_unique <= distinct
    rand = torch.rand(1, 18000, generator=gg)
    orig = torch.rand
    torch.rand = lambda *a, **k: rand.clone()
    try:
        out = model(dict(end_points))
    finally:
        torch.rand = orig
    mine = R.unopose_forward(end_points, sdt, cfg_ref, rand, ext, detail=True)
    for k in ("init_R", "init_t", "pred_R", "pred_t"):
        e = (mine[k] - out[k]).abs().max().item()
        print(f"oracle vs reference {k}: {e:.2e}")
        assert e < 1e-5
    print("pred_R vs ground truth %.2e, pred_t %.2e, score %.3f" % ((out["pred_R"][0] - R_gt).abs().max().item(), (out["pred_t"][0] - t_gt).abs().max().item(),
                                                                   out["pred_pose_score"].item()))
    np.savez_compressed(os.path.join(HERE, "forward_dup.npz"),
                        **{k: v.numpy() for k, v in end_points.items()}, rand=rand.numpy(), R_gt=R_gt.numpy(), t_gt=t_gt.numpy(),
                        n_unique=np.int64(n_unique),
                        **{k: out[k].numpy() for k in ("init_R", "init_t", "init_pose_score", "pred_R", "pred_t", "pred_pose_score")},
                        fps_idx_m=mine["fps_idx_m"].numpy(), fps_idx_o=mine["fps_idx_o"].numpy(), radius=mine["radius"].numpy())


if __name__ == "__main__":
    main()
