"""End-to-end fixture from the REFERENCE's own forward with UNTAMED random weights.
Random weights make the pose itself meaningless, so what is pinned here is every STAGE of the reference forward:
the sampling indices, the dense pixel features, the token features entering `out_proj` of the coarse and of the fine matcher
(captured with forward hooks on the reference's modules), the coarse pose it fed to its fine stage and the final pose.  The GPU test
feeds the reference's FPS subset and its `init_R / init_t` to the HIP path, so nothing in it depends on an ulp-level tie.
Same recipe as make_golden.py::run_forward (reference modules imported with the App-G stubs, `_ext` = the C oracle, the coarse
stage's uniform draw injected); the torch-CPU oracle is asserted equal to the reference before anything is written.
    python tests/golden/make_forward_untamed_golden.py   ->  tests/golden/forward_untamed.npz"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (sets sys.path for the repo root and tests/)

ROWS = 96  # token rows kept per tensor (evenly spaced)


def main():
    MG.import_reference()
    from oracle import unopose_ref as R
    from oracle.pointnet2_oracle import ext
    from core.unopose.model.oneref_grf_predator_pose_estimation_model import UNOPose
    from helpers import congruent_pair

    torch.set_grad_enabled(False)
    nq, nt = 2048, 5000
    cfg_ref = R.default_cfg(fine_npoint=nq, feature_extraction=dict(freeze_vit=False))
    sd = R.random_state_dict(R.default_cfg(), seed=0)  # untamed: the weights of the GPU tests' `model` fixture
    model = UNOPose(cfg_ref).eval()
    model.load_state_dict(sd, strict=True)
    gg = torch.Generator().manual_seed(2041)
    end_points, R_gt, t_gt = congruent_pair(gg, nq=nq, nt=nt, noise=5e-4)
    rand = torch.rand(1, 18000, generator=gg)

    cap = {"coarse": [], "fine": []}
    hooks = [model.coarse_point_matching.out_proj.register_forward_hook(lambda m, i, o: cap["coarse"].append(i[0].clone())),
             model.fine_point_matching.out_proj.register_forward_hook(lambda m, i, o: cap["fine"].append(i[0].clone())),
             model.feature_extraction.register_forward_hook(lambda m, i, o: cap.__setitem__("feat", o))]
    orig = torch.rand
    torch.rand = lambda *a, **k: rand.clone()
    try:
        out = model(dict(end_points))
    finally:
        torch.rand = orig
        for h in hooks:
            h.remove()
    assert len(cap["coarse"]) == 2 and len(cap["fine"]) == 2
    dense_pm, dense_fm, dense_po, dense_fo, radius = cap["feat"][:5]

    mine = R.unopose_forward(end_points, sd, cfg_ref, rand, ext, detail=True)
    c_det = R.coarse_point_matching(mine["sparse_pm"], mine["sparse_fm"], mine["geo_m"], mine["sparse_po"], mine["sparse_fo"], mine["geo_o"], sd,
                                    "coarse_point_matching", cfg_ref.coarse_point_matching, rand, detail=True)[-1]
    f_det = R.fine_point_matching(mine["dense_pm"], mine["dense_fm"], mine["geo_m"], mine["fps_idx_m"], mine["dense_po"], mine["dense_fo"],
                                  mine["geo_o"], mine["fps_idx_o"], mine["init_R"], mine["init_t"], sd, "fine_point_matching",
                                  cfg_ref.fine_point_matching, ext, detail=True)[-1]

    def close(a, b, tol, what):
        e = float((a - b).abs().max() / max(float(b.abs().max()), 1e-12))
        print(f"oracle vs reference {what}: {e:.2e}")
        assert e < tol, what

    close(mine["dense_pm"], dense_pm, 1e-6, "dense_pm")
    close(mine["dense_po"], dense_po, 1e-6, "dense_po")
    close(mine["dense_fm"], dense_fm, 1e-4, "dense_fm")
    close(mine["dense_fo"], dense_fo, 1e-4, "dense_fo")
    close(c_det["f1"], cap["coarse"][0], 1e-4, "coarse f1")
    close(c_det["f2"], cap["coarse"][1], 1e-4, "coarse f2")
    close(f_det["f1"], cap["fine"][0], 2e-4, "fine f1")
    close(f_det["f2"], cap["fine"][1], 2e-4, "fine f2")
    for k in ("init_R", "init_t", "pred_R", "pred_t"):
        close(mine[k], out[k], 1e-4, k)

    rows_c = torch.linspace(0, cap["coarse"][0].shape[1] - 1, ROWS).long()
    rows_f = torch.linspace(0, cap["fine"][0].shape[1] - 1, ROWS).long()
    rows_d = torch.linspace(0, nq - 1, ROWS).long()
    np.savez_compressed(
        os.path.join(HERE, "forward_untamed.npz"),
        **{k: v.numpy() for k, v in end_points.items()}, rand=rand.numpy(),
        **{k: out[k].numpy() for k in ("init_R", "init_t", "init_pose_score", "pred_R", "pred_t", "pred_pose_score")},
        fps_idx_m=mine["fps_idx_m"].numpy(), fps_idx_o=mine["fps_idx_o"].numpy(), radius=radius.numpy(),
        ref_fps_idx=ext.furthest_point_sampling((end_points["tem1_pts"] / (radius.reshape(-1, 1, 1) + 1e-6)).contiguous(), nq).numpy(),
        dense_po=dense_po.numpy(), rows_c=rows_c.numpy(), rows_f=rows_f.numpy(), rows_d=rows_d.numpy(),
        dense_fm_rows=dense_fm[:, rows_d].numpy(), dense_fo_rows=dense_fo[:, rows_d].numpy(),
        dense_fm_absmax=np.float32(dense_fm.abs().max()), dense_fo_absmax=np.float32(dense_fo.abs().max()),
        coarse_f1_rows=cap["coarse"][0][:, rows_c].numpy(), coarse_f2_rows=cap["coarse"][1][:, rows_c].numpy(),
        coarse_f1_absmax=np.float32(cap["coarse"][0].abs().max()), coarse_f2_absmax=np.float32(cap["coarse"][1].abs().max()),
        fine_f1_rows=cap["fine"][0][:, rows_f].numpy(), fine_f2_rows=cap["fine"][1][:, rows_f].numpy(),
        fine_f1_absmax=np.float32(cap["fine"][0].abs().max()), fine_f2_absmax=np.float32(cap["fine"][1].abs().max()))
    sz = os.path.getsize(os.path.join(HERE, "forward_untamed.npz"))
    print(f"forward_untamed.npz written ({sz / 1e6:.2f} MB)")


if __name__ == "__main__":
    main()
