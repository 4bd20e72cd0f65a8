"""Training-path fixtures from the REFERENCE (build container only; imports /root/reference as make_golden.py does):
  train_losses.npz   compute_overlap_loss / process_loss on random block outputs, aug_pose_noise under fixed seeds
  train_forward.npz  UNOPose.forward in TRAIN mode (B = 2, 512 dense / 196 coarse points, stand-in ViT, injected pose
                     noise): every loss entry, BatchNorm running statistics after the step, gradient summaries
    python tests/golden/make_train_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True

from make_golden import import_reference, save  # noqa: E402
from train_case import GRAD_KEYS, make_train_batch, random_block_outputs  # noqa: E402


def main():
    import_reference()
    from oracle import unopose_ref as R
    import core.unopose.utils.loss_utils as LU
    import core.unopose.utils.model_utils as U
    import core.unopose.model.oneref_predator_coarse_point_matching as C
    from core.unopose.model.oneref_grf_predator_pose_estimation_model import UNOPose

    # ---- losses on random block outputs
    g = torch.Generator().manual_seed(7)
    case = random_block_outputs(g)
    ep = LU.compute_overlap_loss({}, case["atten"], case["score"], case["sal"], case["p1"], case["p2"], case["R"], case["t"],
                                 predator_thres=0.15, dis_thres=0.3, loss_str="coarse_hard")
    info = LU.process_loss(dict(ep))
    out = {f"ep__{k}": v for k, v in ep.items()}
    out.update({f"info__{k}": v for k, v in info.items()})
    np.random.seed(3)
    torch.manual_seed(3)
    aR, at = U.aug_pose_noise(case["R"], case["t"])
    # LR multiplier of the configured schedule (configs/main_cfg.py:112-125) at a spread of iterations
    from lib.torch_utils.solver.lr_scheduler import flat_and_anneal_lr_scheduler

    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
    total = 188340
    _, f = flat_and_anneal_lr_scheduler(opt, total_iters=total, warmup_method="linear", warmup_factor=0.001, warmup_iters=1000,
                                        anneal_point=min(1000 / total, 1.0), anneal_method="cosine", target_lr_factor=0.0,
                                        return_function=True)
    its = np.array([0, 1, 10, 500, 999, 1000, 1001, 5000, 94170, 150000, 188339, 188340, 200000])
    save("train_losses", aug_R=aR, aug_t=at, sched_iters=its, sched_factor=np.array([f(int(i)) for i in its]), **out)

    # ---- the whole training forward + backward
    thr = dict(loss_predator_thres=0.15, loss_dis_thres=0.3)  # configs/main_cfg.py:158-159,174-175
    cfg = R.default_cfg(fine_npoint=512, feature_extraction=dict(freeze_vit=False), coarse_point_matching=thr,
                        fine_point_matching=thr)
    sd = R.random_state_dict(cfg, seed=0, tame=0.1)
    model = UNOPose(cfg)
    model.load_state_dict(sd, strict=True)
    for p in model.feature_extraction.rgb_net.vit.parameters():
        p.requires_grad_(False)
    model.train()
    batch, aug = make_train_batch()
    C.aug_pose_noise = lambda gt_r, gt_t, *a, **k: (aug[0].clone(), aug[1].clone())
    torch.set_grad_enabled(True)
    ep = model(dict(batch))
    info = LU.process_loss(ep)
    info["loss"].backward()
    out = {f"ep__{k}": v.detach() for k, v in ep.items() if "coarse_" in k or "fine_" in k}
    out["loss"] = info["loss"].detach()
    params = dict(model.named_parameters())
    for k in GRAD_KEYS:
        gr = params[k].grad
        out[f"gradnorm__{k}"] = gr.norm()
        out[f"gradhead__{k}"] = gr.flatten()[:16].clone()
    bn = model.fine_point_matching.PE.mlp1.layer0.normlayer.bn
    out["bn_running_mean"], out["bn_running_var"] = bn.running_mean.clone(), bn.running_var.clone()
    print({k: (float(v) if v.numel() == 1 else tuple(v.shape)) for k, v in out.items() if "loss" in k or "gradnorm" in k})
    save("train_forward", **out)


if __name__ == "__main__":
    main()
