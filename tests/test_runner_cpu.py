"""CPU: host logic of the sharded runner -- shard rule, pose composition, CSV format, and a
world_size-2 gloo run whose gathered CSV must equal the single-process one."""
import os
import tempfile

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from unopose_amd import runner


def test_shard_range_is_inference_sampler_rule():
    for total in (1, 7, 8, 9, 100, 1023):
        for world in (1, 2, 3, 4, 8):
            parts = [list(runner.shard_range(total, world, r)) for r in range(world)]
            flat = [i for p in parts for i in p]
            assert flat == list(range(total))  # contiguous, exact cover, global order
            sizes = [len(p) for p in parts]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def test_compose_pose_and_csv_format():
    R = torch.eye(3)[None].repeat(2, 1, 1)
    t = torch.tensor([[0.1, 0.2, 0.3], [0.0, 0.0, 1.0]])
    T = torch.eye(4)[None].repeat(2, 1, 1)
    T[:, :3, 3] = torch.tensor([1.0, 2.0, 3.0])
    R2, t2 = runner.compose_pose(R, t, T)
    assert torch.allclose(t2, t + torch.tensor([1.0, 2.0, 3.0]))
    line = runner.csv_line(48, 1, 5, np.float32(0.5), np.eye(3, dtype=np.float32).reshape(9),
                           np.array([10.0, 20.5, 300.25], np.float32), 0.25)
    assert line == "48,1,5,0.5,1.0 0.0 0.0 0.0 1.0 0.0 0.0 0.0 1.0,10.0 20.5 300.25,0.25\n"


class _StubModel:
    """Deterministic stand-in with UNOPose.forward's output contract."""

    def __call__(self, ep):
        B = ep["pts"].shape[0]
        c = ep["pts"].mean(dim=(1, 2))
        ep["pred_R"] = torch.eye(3)[None].repeat(B, 1, 1) * (1 + c.reshape(B, 1, 1))
        ep["pred_t"] = torch.stack([c, 2 * c, 3 * c], 1)
        ep["pred_pose_score"] = torch.sigmoid(c)
        return ep


def _images(n_img=7):
    g = torch.Generator().manual_seed(0)
    out = []
    for i in range(n_img):
        n_inst = 1 + (i * 5) % 19  # exercises chunking by instance_batch_size=16
        out.append(dict(pts=torch.randn(1, n_inst, 64, 3, generator=g), rgb=torch.zeros(1, n_inst, 1),
                        rgb_choose=torch.zeros(1, n_inst, 1), tem1_rgb=torch.zeros(1, n_inst, 1),
                        tem1_choose=torch.zeros(1, n_inst, 1), tem1_pts=torch.zeros(1, n_inst, 1),
                        tem1_pose=torch.eye(4)[None, None].repeat(1, n_inst, 1, 1),
                        score=torch.rand(1, n_inst, 1, generator=g), scene_id=48 + i // 3, img_id=i,
                        obj_id=torch.randint(1, 22, (1, n_inst), generator=g), seg_time=0.0))
    return out


def _strip_time(lines):
    return [l.rsplit(",", 1)[0] for l in lines]


def _worker(rank, world, port, path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = torch.nn.Linear(4, 4)
    if rank != 0:
        torch.nn.init.zeros_(m.weight)
    ref = m.weight.detach().clone()
    runner.broadcast_module_(m, 0)
    flag = torch.tensor([float(torch.equal(m.weight, ref)) if rank == 0 else float(m.weight.abs().sum() > 0)])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    assert flag.item() == 1.0
    runner.inference_and_save(_StubModel(), _images(), path, 16)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_run_equals_single_process():
    with tempfile.TemporaryDirectory() as d:
        single = os.path.join(d, "single.csv")
        lines = runner.inference_and_save(_StubModel(), _images(), single, 16)
        assert len(lines) == sum(1 + (i * 5) % 19 for i in range(7))
        multi = os.path.join(d, "multi.csv")
        mp.spawn(_worker, args=(2, 29517, multi), nprocs=2, join=True)
        assert _strip_time(open(multi).readlines()) == _strip_time(open(single).readlines())
        assert os.path.exists(multi.replace(".csv", ".json"))
