"""bench.py -- (query,ref) pairs/s of UNOPose.forward on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--batch 32] [--img 224] [--dtype bf16|fp32]

A "step" is one UNOPose.forward over a batch of B synthetic (query,ref) pairs already resident in HBM
(BASELINE configs[1]: batch 32, 2048 query points, 5000->2048 reference points, 224x224 crops; the
518x518 stress shape with --img 518).  N>1: one process per GPU (torchrun / RCCL), each rank owns its
own B pairs (the ref-target list shards embarrassingly: weak scaling), weights are broadcast from rank 0
once, poses are gathered to rank 0 at the end of the timed region.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--img", type=int, default=224)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def hip_event_time(fn, iters, stream):
    """Average duration (s) of `fn` measured with events recorded on `stream` (the stream the kernels
    are launched on)."""
    with torch.cuda.stream(stream):
        fn()
        stream.synchronize()
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record(stream)
        for _ in range(iters):
            fn()
        e.record(stream)
        stream.synchronize()
    return s.elapsed_time(e) * 1e-3 / iters


def roofline_leg(model, batch):
    """Roofline of the dominant hand-written kernel of the step, timed live with HIP events on the
    inputs of this very workload.  See DESIGN.md 'Measurement' for the algorithmic-bytes model."""
    from unopose_amd import ops

    B = batch["pts"].shape[0]
    pts = batch["pts"].float()
    radius = torch.norm(batch["tem1_pts"] - batch["tem1_pts"].mean(1, keepdim=True), dim=2).max(1)[0]
    x = (pts / (radius.reshape(-1, 1, 1) + 1e-6)).contiguous()
    N = x.shape[1]
    stream = torch.cuda.current_stream()
    out = {}
    legs = []
    for r, ns in ((0.1, 64), (0.2, 256)):
        t = hip_event_time(lambda: ops.query_lrf_group(x, r, ns), 10, stream)
        byts = 4 * B * (3 * N + 6 * N * ns)  # read the cloud once, write (B,6,N,ns)
        legs.append(dict(kernel=f"query_lrf_group_kernel(ns={ns})", seconds=t, bytes=byts, GBps=byts / t / 1e9))
    # group_points (the reference's `_ext` gather): HBM-write-bound, the kernel the north star prices
    from unopose_amd.pointnet2 import _ext

    idx = _ext.ball_query(x, x, 0.2, 256)
    xt = x.transpose(1, 2).contiguous()
    t = hip_event_time(lambda: _ext.group_points(xt, idx), 20, stream)
    byts = 4 * B * (N * 256 + 3 * N * 256 + 3 * N)
    legs.append(dict(kernel="group_points_lds_kernel(ns=256)", seconds=t, bytes=byts, GBps=byts / t / 1e9))
    dom = max(legs[:2], key=lambda l: l["seconds"])
    out["roofline"] = dict(bound="hbm", kernel=dom["kernel"], achieved=dom["GBps"], peak=8000.0, unit="GB/s",
                           frac=dom["GBps"] / 8000.0, traffic=None)
    out["roofline_other"] = [dict(kernel=l["kernel"], bound="hbm", achieved=l["GBps"], peak=8000.0, unit="GB/s",
                                  frac=l["GBps"] / 8000.0, us=l["seconds"] * 1e6) for l in legs]
    return out


def cpu_baseline_leg(img):
    """The oracle (torch-CPU port of the reference forward + the C `_ext` port) timed on this host's
    cores on a bounded sample: ONE pair at the workload's shapes, one forward."""
    from oracle import unopose_ref as R
    from oracle.pointnet2_oracle import ext as oext
    from unopose_amd.synthetic import congruent_pair

    torch.set_num_threads(min(32, os.cpu_count()))
    g = torch.Generator().manual_seed(1)
    ep, _, _ = congruent_pair(g, 2048, 5000, img, 5e-4)
    cfg = R.default_cfg()
    sd = R.random_state_dict(cfg, seed=0, img_size=img, tame=0.1)
    rand = torch.rand(1, 18000, generator=g)
    t0 = time.perf_counter()
    with torch.no_grad():
        R.unopose_forward(ep, sd, cfg, rand, oext)
    dt = time.perf_counter() - t0
    return dict(value=1.0 / dt, unit="pairs/s", cores=torch.get_num_threads(), kind="port",
                sample=f"1 pair (2048 query / 5000 reference points, {img}x{img} crops), 1 forward, fp32, "
                       f"torch {torch.get_num_threads()} threads + C `_ext` port")


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI

    from unopose_amd.model import UNOPose, default_model_cfg
    from unopose_amd.synthetic import make_batch, trained_like_

    torch.set_grad_enabled(False)
    torch.manual_seed(0)
    cfg = default_model_cfg(feature_extraction=dict(img_size=args.img))
    model = trained_like_(UNOPose(cfg)).to(dev).eval()
    if world > 1:  # one broadcast of the flat weights from rank 0
        flat = torch.cat([p.data.reshape(-1) for p in model.parameters()] +
                         [b.data.float().reshape(-1) for b in model.buffers()])
        dist.broadcast(flat, 0)
        off = 0
        for t in list(model.parameters()) + list(model.buffers()):
            n = t.numel()
            t.data.copy_(flat[off:off + n].reshape(t.shape).to(t.dtype))
            off += n
    B = args.batch
    batch, R_gt, t_gt = make_batch(B, 2048, 5000, args.img, seed=100 + rank, device=dev)
    batch["coarse_rand"] = torch.rand(B, 18000, device=dev)
    amp = args.dtype == "bf16"

    def step():
        ep = dict(batch)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            out = model(ep)
        return out

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    poses = torch.cat([out["pred_R"].reshape(B, 9), out["pred_t"], out["pred_pose_score"].reshape(B, 1)], 1)
    if world > 1:  # gather of poses to rank 0 (the reference lacks it: every rank writes the same file)
        gathered = [torch.empty_like(poses) for _ in range(world)] if rank == 0 else None
        dist.gather(poses, gathered, 0)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()

    rot_err = (out["pred_R"] - R_gt).abs().amax(dim=(1, 2))
    res = {
        "metric": "(query,ref) pairs/sec forward",
        "value": world * B * args.steps / dt,
        "unit": "pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic",
        "config": {"workload": f"UNOPose.forward, batch {B} pairs/GPU, 2048 query pts, 5000->2048 reference pts, "
                               f"196 coarse pts, {args.img}x{args.img} crops, DINOv2 ViT-B/14 reg4, "
                               f"random-init (trained-like) weights",
                   "sharding": f"dp{world} (independent pairs, weights broadcast, poses gathered)"},
        "sanity": {"median_rot_err_vs_gt": rot_err.median().item(),
                   "frac_pairs_solved(<0.05)": (rot_err < 0.05).float().mean().item()},
    }
    if rank == 0 and world == 1:
        if not args.no_roofline:
            res.update(roofline_leg(model, batch))
        if not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline_leg(args.img)
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
