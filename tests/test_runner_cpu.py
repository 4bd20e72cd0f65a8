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


class _StubEncoder:
    """encode_reference stand-in: features are a function of the view alone; counts encodes."""

    def __init__(self):
        self.calls = 0

    def encode_reference(self, rgb, choose, pts):
        self.calls += rgb.shape[0]
        return dict(ref_dense_po=pts * 2, ref_dense_fo=pts + 1, ref_radius=pts.sum(dim=(1, 2)), ref_lrf=pts - 1)


def test_reference_cache_is_lru_and_never_evicts_the_current_batch():
    """max_items smaller than the number of distinct views (the FIFO of round 1 raised KeyError here)."""
    enc = _StubEncoder()
    cache = runner.ReferenceCache(enc, max_items=2)
    views = {k: torch.full((4, 3), float(i)) for i, k in enumerate("ABCDE")}

    def look(keys):
        pts = torch.stack([views[k] for k in keys])
        out = cache.lookup(list(keys), pts, pts, pts)
        assert torch.equal(out["ref_dense_po"], pts * 2) and torch.equal(out["ref_radius"], pts.sum(dim=(1, 2)))
        assert len(cache.store) <= 2

    look("AB")
    assert enc.calls == 2
    look("AC")  # C is encoded while A (needed by this very call) is the oldest entry
    assert enc.calls == 3 and set(cache.store) == {"A", "C"}  # B, least recently used, went
    look("A")
    look("D")  # evicts C (A was refreshed by the hit)
    assert set(cache.store) == {"A", "D"} and enc.calls == 4
    look("ABCDE")  # a batch with more distinct views than the cache holds still resolves
    assert enc.calls == 7 and (cache.hits, cache.misses) == (1 + 1 + 2, 7)
    # entries own their memory (an eviction frees it): not views into the encoder's batch tensor
    e = next(iter(cache.store.values()))["ref_dense_po"]
    assert e.untyped_storage().nbytes() == e.numel() * e.element_size()


def test_detections_json_keeps_the_reference_layout():
    """oneref_inference_utils_v1.py:31,112-113,134: deep copy of dataset.dets (lists of full detection dicts,
    filtered detections included) + pred_R / pred_t on the entries `inst_ids` names."""
    import json

    imgs = _images(3)
    dets = {}
    for im in imgs:
        n = im["pts"].shape[1]
        key = f"{im['scene_id']:06d}_{im['img_id']:06d}"
        dets[key] = [dict(scene_id=im["scene_id"], image_id=im["img_id"], category_id=5, score=0.5, bbox=[1, 2, 3, 4],
                          time=0.1, segmentation={"size": [4, 4], "counts": [16]}) for _ in range(n + 2)]
        im["inst_ids"] = torch.arange(1, n + 1)[None]  # detection 0 and the last one were filtered out

    class _Set(list):
        pass

    data = _Set(imgs)
    data.dets = dets
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "r.csv")
        runner.inference_and_save(_StubModel(), data, path, 16)
        out = json.load(open(path.replace(".csv", ".json")))
    assert set(out) == set(dets) and "pred_R" not in dets[next(iter(dets))][1]  # the input is not mutated
    for key, lst in out.items():
        assert len(lst) == len(dets[key])
        assert "pred_R" not in lst[0] and "pred_R" not in lst[-1] and lst[0]["bbox"] == [1, 2, 3, 4]
        for det in lst[1:-1]:
            assert len(det["pred_R"]) == 9 and len(det["pred_t"]) == 3 and det["segmentation"]["counts"] == [16]


def test_runner_reproduces_the_reference_loop_capture():
    """tests/golden/runner_capture.json: CSV rows (minus the wall-clock column) and the detections JSON written by the
    reference's inference_and_save_oneref_v1 on tests/runner_case.py's inputs with the same stand-in model."""
    import json

    from runner_case import StubModel, make_case

    cap = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "runner_capture.json")))
    images, dets = make_case()
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "result.csv")
        lines = runner.inference_and_save(StubModel(), images, path, instance_batch_size=4, dets=dets)
        out_dets = json.load(open(path.replace(".csv", ".json")))
    assert _strip_time(lines) == cap["csv_without_time"]
    assert [len(l.split(",")) for l in lines] == cap["n_time_fields"]
    assert all(float(l.rsplit(",", 1)[1]) > 0.25 for l in lines)  # time = loop time + the detector's seg_time
    assert out_dets == cap["dets"]
    assert "pred_R" not in dets["000048_000001"][1]  # caller's detections untouched (deep copy)
