"""One training step of UNOPose on MI355X and its data-parallel wrapper (BASELINE configs[3]).

The reference's loop (core/unopose/engine/engine.py:136-220, engine_utils.py:53-83): forward in train mode -> `process_loss`
-> backward -> NaN / inf gradients zeroed -> optional clipping -> Adam(lr 1e-4, betas (0.5, 0.999), eps 1e-6) -> flat-and-
anneal LR multiplier.  Data parallelism is replicas + gradient all-reduce: torch DistributedDataParallel over RCCL (backend
"nccl" on ROCm), gradient buckets reduced during backward; no other collective.  The frozen DINOv2 backbone runs on the
fused inference kernels under no_grad; everything trainable runs op-by-op under autograd (ops.differentiable)."""
import math

import torch

from .losses import process_loss


def flat_and_anneal_factor(it, total_iters, warmup_iters=1000, warmup_factor=0.001, anneal_point=None, target_lr_factor=0.0):
    """LR multiplier at iteration `it`: linear warm-up from `warmup_factor`, flat at 1, cosine anneal from
    `anneal_point * total_iters` to `target_lr_factor` at `total_iters`, constant after (lib/torch_utils/solver/
    lr_scheduler.py:148-265 with the configured methods: warmup "linear", anneal "cosine", configs/main_cfg.py:112-125)."""
    if anneal_point is None:
        anneal_point = min(1000 / total_iters, 1.0)
    if it < warmup_iters:
        a = float(it) / warmup_iters
        return (1 - warmup_factor) * a + warmup_factor
    start = anneal_point * total_iters
    if it < start:
        return 1.0
    if it < total_iters:
        return target_lr_factor + 0.5 * (1 - target_lr_factor) * (1 + math.cos(math.pi * (float(it) - start) / (total_iters - start)))
    return target_lr_factor


def build_optimizer(model, lr=1e-4, total_iters=188340, **sched):
    """Adam as configured (configs/main_cfg.py:97-110) over the trainable parameters + the flat-and-anneal schedule."""
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=lr, betas=(0.5, 0.999), eps=1e-6, weight_decay=0.0)
    return opt, torch.optim.lr_scheduler.LambdaLR(opt, lambda it: flat_and_anneal_factor(it, total_iters, **sched))


def freeze_backbone(model):
    """`freeze_vit=True` (configs/main_cfg.py:141, oneref_feature_extraction.py:194-198): the ViT's parameters leave
    the optimiser; the up-projection stays trainable.  Parameters no forward path reads (`vit.head`, `dis_proj`) are
    frozen as well, so DistributedDataParallel needs no unused-parameter search."""
    for p in model.feature_extraction.rgb_net.vit.parameters():
        p.requires_grad_(False)
    for p in model.fine_point_matching.dis_proj.parameters():
        p.requires_grad_(False)
    return model


def wrap_ddp(model, device=None):
    """Replicas + bucketed gradient all-reduce overlapped with backward (RCCL over xGMI when the process group is "nccl").
    broadcast_buffers=False as the reference configures it (BN running statistics stay per rank)."""
    from torch.nn.parallel import DistributedDataParallel as DDP

    ids = None if device is None or torch.device(device).type != "cuda" else [torch.device(device).index]
    return DDP(model, device_ids=ids, broadcast_buffers=False, find_unused_parameters=False)


def zero_nonfinite_grads_(model):
    """engine_utils.py:14-18: NaN -> 0, +inf -> 1e5, -inf -> -1e5, in place."""
    for p in model.parameters():
        if p.grad is not None:
            torch.nan_to_num(p.grad, nan=0.0, posinf=1e5, neginf=-1e5, out=p.grad)


def train_step(model, batch, optimizer, scheduler=None, clip_max_norm=None, amp_dtype=None):
    """forward (train mode) -> process_loss -> backward -> gradient hygiene -> optimiser (+ scheduler) step.
    Returns the scalar dict of process_loss (detached)."""
    model.train()
    with torch.autocast("cuda", dtype=amp_dtype or torch.bfloat16, enabled=amp_dtype is not None):
        out = model(dict(batch))
        info = process_loss(out)
    loss = info["loss"]
    finite = torch.isfinite(loss.detach()).all()  # (a device flag: read at the END of the step -- reading it here would stall the host
    #                                                between forward and backward: same-box 119.1 / 111.0 / 115.2 ms with the early check, 111.1 / 110.8 / 112.1 with this one)
    optimizer.zero_grad(set_to_none=True)
    loss.backward()
    zero_nonfinite_grads_(model)
    if clip_max_norm is not None:
        torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.grad is not None], clip_max_norm)
    optimizer.step()
    if scheduler is not None:
        scheduler.step()
    if not finite:  # the reference asserts a finite loss every iteration (engine.py); a NaN step has been cleaned by the gradient hygiene above
        raise FloatingPointError(f"non-finite loss {loss.detach()}")
    return {k: v.detach() for k, v in info.items()}
