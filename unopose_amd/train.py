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


MULTI_TENSOR_HYGIENE = True  # (module attribute for A/Bs) zero_nonfinite_grads_ as one launch over all gradients; False: torch.nan_to_num per parameter


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


_PINNED = {}  # device -> [ring of pinned int64 staging buffers, next slot]


def _table_to_device(values, dev):
    """A small int64 table on `dev` through a ring of four persistent pinned buffers (asynchronous copy, no allocation per step; train_step
    synchronises with the device once per step, so a slot is never overwritten while its copy is still queued)."""
    n = len(values)
    ring = _PINNED.get(dev)
    if ring is None or ring[0][0].numel() < n:
        ring = _PINNED[dev] = [[torch.empty(max(n, 1024), dtype=torch.int64).pin_memory() for _ in range(4)], 0]
    buf = ring[0][ring[1] % 4]
    ring[1] += 1
    buf[:n] = torch.tensor(values, dtype=torch.int64)
    return buf[:n].to(dev, non_blocking=True)


def zero_nonfinite_grads_(model):
    """engine_utils.py:14-18: NaN -> 0, +inf -> 1e5, -inf -> -1e5, in place.  On the GPU all fp32 contiguous gradients go through ONE launch
    (csrc/glue.hip nan_to_num_multi: a device table of pointers / sizes, refilled every step -- the gradients are new tensors after zero_grad(set_to_none=True));
    whatever does not fit that (other dtypes, strided or CPU gradients) keeps torch.nan_to_num."""
    grads = [p.grad for p in model.parameters() if p.grad is not None]
    multi = {}
    for g in grads:
        if g.is_cuda and g.dtype == torch.float32 and g.is_contiguous() and MULTI_TENSOR_HYGIENE:
            multi.setdefault(g.device, []).append(g)
        else:
            torch.nan_to_num(g, nan=0.0, posinf=1e5, neginf=-1e5, out=g)
    for dev, gs in multi.items():
        from ._lib import call, on_device, ptr, stream_ptr
        gs = [g for g in gs if g.numel()]
        for i in range(0, len(gs), 65535):
            part = gs[i:i + 65535]
            table = _table_to_device([g.data_ptr() for g in part] + [g.numel() for g in part], dev)
            with on_device(dev):
                call("unopose_nan_to_num_multi", ptr(table), ptr(table[len(part):]), len(part), max(g.numel() for g in part), 0.0, 1e5, -1e5, stream_ptr())


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
