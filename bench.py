"""bench.py -- (query,ref) pairs/s of UNOPose.forward on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--batch 32] [--img 224] [--dtype bf16|fp32]

A "step" is one UNOPose.forward over a batch of B synthetic (query,ref) pairs already resident in HBM
(BASELINE configs[1]: batch 32, 2048 query points, 5000->2048 reference points, 518x518 crops; the
reference's own 224x224 contract with --img 224).  N>1: one process per GPU (torchrun / RCCL), each rank owns its
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
    ap.add_argument("--img", type=int, default=518,
                    help="crop side: 518 = BASELINE configs[1] / north_star (default); 224 = the reference's own contract")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--graph", action="store_true", help="replay the forward as one hipGraph (same GPU time: "
                    "the step is GPU-bound, not launch-bound, at every batch size measured)")
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


def roofline_leg(model, batch, img):
    """Roofline of the dominant hand-written kernel of the step, timed live with HIP events on the stream
    it runs on: the ViT patch attention at 518x518 crops (12 launches = the largest share of the step
    after the library GEMMs), the fused positional-encoding kernel (S=256 launch) at 224x224.  The other
    rows BASELINE's north star prices follow in `roofline_other`.  Work models: DESIGN.md section 4."""
    from unopose_amd import ops
    from unopose_amd.pointnet2 import _ext

    B = batch["pts"].shape[0]
    pts = batch["pts"].float()
    radius = torch.norm(batch["tem1_pts"] - batch["tem1_pts"].mean(1, keepdim=True), dim=2).max(1)[0]
    x = (pts / (radius.reshape(-1, 1, 1) + 1e-6)).contiguous()
    N = x.shape[1]
    stream = torch.cuda.current_stream()
    pe = model.fine_point_matching.PE
    rows = []

    def row(kernel, bound, work, unit_scale, peak, unit, seconds, note=None):
        ach = work / seconds / unit_scale
        r = dict(kernel=kernel, bound=bound, achieved=ach, peak=peak, unit=unit, frac=ach / peak,
                 us=seconds * 1e6)
        if note:
            r["note"] = note
        rows.append(r)
        return r

    # ViT patch attention: 4 T^2 64 flop per (image, head), 2B images x 12 heads, T = 5 + (img/14)^2 tokens
    T = 5 + (img // 14) ** 2
    qkv = torch.randn(2 * B, T, 2304, device=x.device).bfloat16()
    t = hip_event_time(lambda: ops.vit_attention(qkv, 12), 10, stream)
    vit = row("vit_attn_kernel(T=%d)" % T, "mfma", 2.0 * B * 12 * 4.0 * T * T * 64, 1e12, 2500.0, "TFLOP/s", t,
              "QK^T + PV flops; softmax exp/sum VALU work shares the issue port with the matrix core")
    del qkv
    # PE, S=256, bf16 hi/lo-split matrix cores; 20864 flop per neighbour row
    t = hip_event_time(lambda: ops.pe_group_mlp_max(x, pe.r2, pe.ns2, pe.mlp2, bf16x3=True), 10, stream)
    dom = row("pe_group_mlp_max_bf16x3_kernel(S=%d)" % pe.ns2, "mfma", B * N * pe.ns2 * 20864.0, 1e12, 2500.0,
              "TFLOP/s", t, "algorithmic fp32-equivalent flops; the kernel issues 3 bf16 MFMAs per product")
    t = hip_event_time(lambda: ops.pe_group_mlp_max(x, pe.r2, pe.ns2, pe.mlp2, bf16x3=False), 5, stream)
    row("pe_group_mlp_max_kernel(S=%d, exact fp32 MFMA)" % pe.ns2, "mfma", B * N * pe.ns2 * 20864.0, 1e12, 157.3,
        "TFLOP/s", t)
    # geometric embedding: 8 n^2 256^2 flop per cloud
    n = model.coarse_npoint + 1
    gp = torch.cat([torch.ones(B, 1, 3, device=x.device), x[:, :n - 1]], 1).contiguous()
    t = hip_event_time(lambda: ops.geo_embedding(gp, model.geo_embedding, out_dtype=torch.bfloat16), 10, stream)
    row("geo_embed_kernel<bf16>", "mfma", B * 8.0 * n * n * 256 * 256, 1e12, 2500.0, "TFLOP/s", t)
    # group_points (the reference's `_ext` gather): HBM-write-bound
    idx = _ext.ball_query(x, x, 0.2, 256)
    xt = x.transpose(1, 2).contiguous()
    t = hip_event_time(lambda: _ext.group_points(xt, idx), 20, stream)
    row("group_points_lds_kernel(S=256)", "hbm", 4.0 * B * (N * 256 + 3 * N * 256 + 3 * N), 1e9, 8000.0, "GB/s", t)
    t = hip_event_time(lambda: _ext.ball_query(x, x, 0.2, 256), 20, stream)
    row("ball_query_kernel(S=256)", "hbm", 4.0 * B * (3 * N + 3 * N + N * 256), 1e9, 8000.0, "GB/s", t)
    tem = batch["tem1_pts"].float().contiguous()
    t = hip_event_time(lambda: _ext.furthest_point_sampling(tem, 2048), 3, stream)
    rows.append(dict(kernel="fps_kernel(5000->2048)", bound="latency", us=t * 1e6, us_per_iteration=t * 1e6 / 2047))
    if img >= 448:  # 12 attention launches outweigh the 2 PE launches once T^2 grows
        dom, key = vit, "vit_attn_kernel"
    else:
        key = "pe_group_mlp_max_bf16x3_kernel"
    traffic = _pmc_traffic(key, B) if (key != "vit_attn_kernel" or T == 1374) else None
    out = dict(roofline=dict(bound=dom["bound"], kernel=dom["kernel"], achieved=dom["achieved"], peak=dom["peak"],
                             unit=dom["unit"], frac=dom["frac"], traffic=traffic, note=dom["note"]),
               roofline_other=[r for r in rows if r is not dom])
    return out


def _pmc_traffic(kernel, B):
    """HBM bytes per launch from the committed PMC passes (profiles/r01_pmc_summary.json: FETCH_SIZE and
    WRITE_SIZE collected in separate rocprofv3 runs at B=32, FETCH_SIZE doubled per the gfx950 note);
    null when the batch differs from the profiled one."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
    if B != 32 or not os.path.exists(path):
        return None
    k = json.load(open(path))["kernels"].get(kernel)
    return None if k is None else dict(hbm_bytes_per_launch=k["hbm_bytes_per_launch"], source="profiles/r01_pmc_summary.json")


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
    # UNOPOSE_BENCH_BACKEND=gloo is a TEST hook: it lets the N>1 code path run with several ranks sharing one
    # GPU (collectives staged through host memory); the driver's runs use the default, RCCL over xGMI.
    # Keep such shared-GPU runs short and at --img 224: concurrent stream-K library GEMMs of two processes can
    # starve each other's workgroups.
    backend = os.environ.get("UNOPOSE_BENCH_BACKEND", "nccl")
    local = local if backend == "nccl" else local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    comm_dev = dev if backend == "nccl" else torch.device("cpu")
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI
        else:
            dist.init_process_group(backend)

    from unopose_amd.model import UNOPose, default_model_cfg
    from unopose_amd.synthetic import make_batch, trained_like_

    torch.set_grad_enabled(False)
    torch.manual_seed(0)
    cfg = default_model_cfg(feature_extraction=dict(img_size=args.img))
    model = trained_like_(UNOPose(cfg)).to(dev).eval()
    if world > 1:  # one flat broadcast of the weights from rank 0 over RCCL / xGMI
        from unopose_amd.runner import broadcast_module_

        broadcast_module_(model, 0)
    B = args.batch
    batch, R_gt, t_gt = make_batch(B, 2048, 5000, args.img, seed=100 + rank, device=dev)
    batch["coarse_rand"] = torch.rand(B, 18000, device=dev)
    amp = args.dtype == "bf16"

    graphed = None
    if args.graph:
        from unopose_amd.graph import GraphedForward

        graphed = GraphedForward(model, batch, torch.bfloat16 if amp else None)

    def step():
        ep = dict(batch)
        if graphed is not None:  # inputs copied into the graph's static buffers, one hipGraphLaunch
            return graphed(ep)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            return model(ep)

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
        poses = poses.to(comm_dev)
        gathered = [torch.empty_like(poses) for _ in range(world)] if rank == 0 else None
        dist.gather(poses, gathered, 0)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=comm_dev, dtype=torch.float64)
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
        "launch": "hipGraph replay" if args.graph else "eager",
        "sanity": {"median_rot_err_vs_gt": rot_err.median().item(),
                   "frac_pairs_solved(<0.05)": (rot_err < 0.05).float().mean().item()},
    }
    if rank == 0 and world == 1:
        if not args.no_roofline:
            res.update(roofline_leg(model, batch, args.img))
        if not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline_leg(args.img)
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
