"""One more end-to-end fixture from the REFERENCE's own forward: a pair whose query and reference crops are
DIFFERENT images (every other end-to-end fixture shares one image between the two views), plus depth noise and a 40-degree relative
rotation.  Same recipe as make_golden.py::run_forward (reference modules imported with the App-G stubs, tamed weights, the coarse
stage's uniform draw injected), the oracle asserted equal to the reference at 1e-5 before anything is written.
    python tests/golden/make_forward_diffimg_golden.py   ->  tests/golden/forward_diffimg.npz"""
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
    gg = torch.Generator().manual_seed(91)
    end_points, R_gt, t_gt = congruent_pair(gg, nq=nq, nt=nt, noise=1e-3)
    # the reference view gets its own image: a different random field, brighter and smoother than the query crop
    img2 = torch.randn(1, 3, 224, 224, generator=gg)
    img2 = 0.6 * img2 + 0.4 * torch.nn.functional.avg_pool2d(img2, 5, 1, 2) + 0.3
    end_points["tem1_rgb"] = img2.contiguous()
    assert (end_points["tem1_rgb"] - end_points["rgb"]).abs().mean() > 0.5
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
    print("pred_R vs ground truth %.2e, score %.3f" % ((out["pred_R"][0] - R_gt).abs().max().item(), out["pred_pose_score"].item()))
    np.savez_compressed(os.path.join(HERE, "forward_diffimg.npz"),
                        **{k: v.numpy() for k, v in end_points.items()}, rand=rand.numpy(), R_gt=R_gt.numpy(), t_gt=t_gt.numpy(),
                        **{k: out[k].numpy() for k in ("init_R", "init_t", "init_pose_score", "pred_R", "pred_t", "pred_pose_score")},
                        fps_idx_m=mine["fps_idx_m"].numpy(), fps_idx_o=mine["fps_idx_o"].numpy(), radius=mine["radius"].numpy())


if __name__ == "__main__":
    main()
