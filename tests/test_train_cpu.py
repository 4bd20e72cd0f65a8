"""CPU: host logic of the training path against fixtures captured from the reference
(tests/golden/make_train_golden.py): losses, pose-noise augmentation, LR schedule; plus a world_size-2 gloo run of
train_step under DistributedDataParallel (the N > 1 training path: replicas + gradient all-reduce)."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from train_case import random_block_outputs
from unopose_amd import losses, train

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def test_overlap_losses_and_process_loss_match_reference():
    z = _load("train_losses")
    case = random_block_outputs(torch.Generator().manual_seed(7))
    ep = losses.overlap_losses({}, case["atten"], case["score"], case["sal"], case["p1"], case["p2"], case["R"], case["t"], 0.15,
                               0.3, "coarse_hard")
    want = {k[4:]: v for k, v in z.items() if k.startswith("ep__")}
    assert set(ep) == set(want)
    for k, v in want.items():
        assert torch.allclose(ep[k], v, rtol=1e-6, atol=1e-6), k
    info = losses.process_loss(ep)
    for k, v in z.items():
        if k.startswith("info__"):
            assert torch.allclose(info[k[6:]], v, rtol=1e-6, atol=1e-6), k
    assert 0 < float(ep["coarse_hard_acc"].mean()) < 1 and float(ep["coarse_hard_fg_num"].min()) > 0  # a non-degenerate case


def test_aug_pose_noise_matches_reference_stream():
    z = _load("train_losses")
    case = random_block_outputs(torch.Generator().manual_seed(7))
    np.random.seed(3)
    torch.manual_seed(3)
    R, t = losses.aug_pose_noise(case["R"], case["t"])
    assert torch.allclose(R, z["aug_R"], atol=1e-6) and torch.allclose(t, z["aug_t"], atol=1e-6)
    assert torch.allclose(R @ R.transpose(1, 2), torch.eye(3).expand_as(R), atol=1e-5) and (t[:, 2] >= 1e-6).all()


def test_flat_and_anneal_schedule_matches_reference():
    z = _load("train_losses")
    got = [train.flat_and_anneal_factor(int(i), 188340) for i in z["sched_iters"]]
    assert np.allclose(got, z["sched_factor"].numpy(), rtol=1e-12, atol=1e-15)
    opt, sched = train.build_optimizer(torch.nn.Linear(2, 2), lr=1e-4, total_iters=188340)
    assert opt.defaults["betas"] == (0.5, 0.999) and opt.defaults["eps"] == 1e-6
    assert abs(opt.param_groups[0]["lr"] - 1e-4 * 0.001) < 1e-12  # iteration 0: warm-up factor


class _Toy(torch.nn.Module):
    """Stand-in with UNOPose's training contract: returns an end_points dict holding per-sample `*_loss*` entries."""

    def __init__(self):
        super().__init__()
        self.lin = torch.nn.Linear(6, 3)

    def forward(self, ep):
        y = self.lin(ep["x"])
        ep["fine_atten_loss0"] = ((y - ep["y"]) ** 2).mean(1)
        ep["coarse_hard_acc"] = torch.ones(y.shape[0])
        return ep


def _ddp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    model = train.wrap_ddp(_Toy())
    opt, sched = train.build_optimizer(model, lr=1e-2, total_iters=100, warmup_iters=2)
    g = torch.Generator().manual_seed(100 + rank)  # every rank its own shard
    batch = {"x": torch.randn(4, 6, generator=g), "y": torch.randn(4, 3, generator=g)}
    first = last = None
    for _ in range(20):
        info = train.train_step(model, batch, opt, sched, clip_max_norm=35.0)
        first = first if first is not None else float(info["loss"])
        last = float(info["loss"])
    w = torch.cat([p.detach().flatten() for p in model.parameters()])
    ws = [torch.empty_like(w) for _ in range(world)]
    dist.all_gather(ws, w)
    if rank == 0:
        q.put((first, last, float((ws[0] - ws[1]).abs().max())))
    dist.destroy_process_group()


def test_train_step_ddp_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    mp.spawn(_ddp_worker, args=(2, 29533, q), nprocs=2, join=True)
    first, last, spread = q.get()
    assert last < first and spread == 0.0  # the loss goes down and the replicas stay bit-identical (all-reduced gradients)


def test_zero_nonfinite_grads():
    m = torch.nn.Linear(2, 2)
    m.weight.grad = torch.tensor([[float("nan"), float("inf")], [-float("inf"), 1.0]])
    train.zero_nonfinite_grads_(m)
    assert m.weight.grad.tolist() == [[0.0, 1e5], [-1e5, 1.0]]
