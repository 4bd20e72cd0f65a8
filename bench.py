"""bench.py -- (query,ref) pairs/s of UNOPose.forward on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--batch 32] [--img 518] [--dtype bf16|fp32]

A "step" is one UNOPose.forward over a batch of B synthetic (query,ref) pairs already resident in HBM
(BASELINE configs[1]: batch 32, 2048 query points, 5000->2048 reference points, 518x518 crops; the
reference's own 224x224 contract with --img 224).

N > 1: one process per GPU over RCCL.  Launched either by the driver's
``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`` (RANK / LOCAL_RANK / WORLD_SIZE in
the environment) or directly as ``python bench.py --gpus N``: then THIS process touches no GPU, starts the N
ranks as child processes (fresh interpreters, never an exec of a process that has initialised HIP) and exits
with their status.  Each rank owns its own B pairs (the ref-target list shards embarrassingly: weak scaling,
`runner.shard_range` = the reference's InferenceSampler rule), weights are broadcast from rank 0 once, poses
are gathered to rank 0 at the end of the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# (the same default unopose_amd/__init__.py sets, here because this process touches the GPU before it imports the package: the runner
#  uses up to nine streams, the HIP runtime maps them onto 4 hardware queues by default and streams sharing a queue serialise)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--img", type=int, default=518,
                    help="crop side: 518 = BASELINE configs[1] / north_star (default); 224 = the reference's own contract")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--graph", action="store_true", help="replay the forward as one hipGraph (same GPU time: "
                    "the step is GPU-bound, not launch-bound, at every batch size measured)")
    ap.add_argument("--inflight", type=int, default=2,
                    help="forwards in flight per GPU (unopose_amd.pipeline.PipelinedForward): consecutive steps are dealt "
                         "round-robin over this many HIP streams; a step is still one forward over one batch, and the "
                         "latency-bound matcher of one batch runs underneath the ViT of the next.  1 = one stream")
    ap.add_argument("--stages", default="auto", choices=["auto", "0", "1"],
                    help="PipelinedForward(stages=...): 1 = ViT half and matcher half of every forward on two streams, 0 = whole "
                         "forwards side by side, auto = by ViT size")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-all-cores", action="store_true",
                    help="also time the CPU port once with EVERY hardware thread (adds ~10 min on a 256-thread host: 292 s per B=8 forward, "
                         "0.027 pairs/s -- profiles/r03_bench_n1_bf16_s518_first_with_all_cores.json holds that run)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-fp32", action="store_true", help="skip the extra fp32 (reference default precision) leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the bounded extra legs (`ref_cached`, `contract_224`, `train`)")
    ap.add_argument("--train", action="store_true",
                    help="time the TRAINING step instead (BASELINE configs[3]: --batch 8 pairs per rank, --train-pts 4096 points per cloud, "
                         "224x224 crops, frozen backbone, Adam; N > 1: DistributedDataParallel, gradient all-reduce over RCCL).  A step = "
                         "forward in train mode + process_loss + backward + gradient hygiene + optimiser step")
    ap.add_argument("--train-pts", type=int, default=4096, help="query points and fine_npoint of the training step (the reference's own "
                    "training config: 2048 with 5000 reference points)")
    ap.add_argument("--train-dtype", default="fp32", choices=["fp32", "bf16"], help="fp32 = the reference's training default (train.amp off)")
    ap.add_argument("--dry-run", action="store_true",
                    help="TEST HOOK (tests/test_bench_cpu.py): replace the model step by a host stub so that the launch / "
                         "rendezvous / barrier / gather / JSON path runs without a GPU (backend gloo); never a measurement")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------------------
# N > 1 without a launcher: start the ranks as children BEFORE anything touches the GPU
# ------------------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n):
    """Start `n` fresh interpreters of this script, one per GPU, with the torchrun environment contract
    (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR=127.0.0.1, MASTER_PORT); rank 0 inherits stdout (the JSON line).
    Returns the worst exit code.  The parent has not initialised HIP (argparse + subprocess only)."""
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


_T0 = time.perf_counter()


def log(msg):
    """Progress on stderr (stdout carries only the JSON line)."""
    print("[bench %7.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


def hip_event_time(fn, iters, stream, warm=1):
    """Average duration (s) of `fn` measured with events recorded on `stream` (the stream the kernels
    are launched on)."""
    import torch

    with torch.cuda.stream(stream):
        for _ in range(warm):
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
    """Roofline of the dominant hand-written kernel of the step, timed live with HIP events on the stream it runs on:
    `gemm256_kernel` (csrc/gemm_kernel.h, launched by csrc/gemm.hip), the linear layers of the ViT -- 48 launches per step, the largest share of
    the step's GPU time.  `roofline` prices the four shapes of one ViT-B block (qkv, proj, fc1 + bias + GELU, fc2) at
    the step's row count M = 2B x T flop-weighted: achieved = (sum of their flops) / (sum of their launch times).
    The other rows BASELINE's north star prices follow in `roofline_other`.  Work models: DESIGN.md section 4."""
    import torch

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

    # the ViT linears: the four shapes of one block, bf16, M = 2B x T rows, T = 5 + (img/14)^2 tokens -- in the form the step runs them.
    # Round 6: with the residual + LayerNorm passes folded into the GEMMs (ops.USE_LN_FOLD; 47 of a forward's 48 launches) that is
    # proj / fc2 with the fp32 residual stream in their epilogue (EPI 5) and qkv / fc1 with LayerNorm applied in theirs (EPI 6 / 7): the
    # launches are longer than the plain-epilogue ones (timed beside them, `plain_epilogue_us`) because they absorb the 23
    # scale_residual_layernorm launches of a forward (2.9 ms), and their algorithmic bytes include the residual stream
    T = 5 + (img // 14) ** 2
    M = 2 * B * T
    fold = ops.ln_fold_ok(M, 768)
    shapes, flops, secs, alg_bytes, secs_plain = [], 0.0, 0.0, 0.0, 0.0
    import torch.nn as nn

    norm = nn.LayerNorm(768, eps=1e-6).to(x.device)
    gamma = nn.Parameter(0.05 + 0.45 * torch.rand(768, device=x.device))
    for name, K_, N_, gelu in (("qkv", 768, 2304, False), ("proj", 768, 768, False), ("fc1", 768, 3072, True),
                               ("fc2", 3072, 768, False)):
        a = torch.randn(M, K_, device=x.device).bfloat16()
        w = (torch.randn(N_, K_, device=x.device) / K_ ** 0.5).bfloat16()
        bias = torch.randn(N_, device=x.device)
        t_plain = hip_event_time(lambda: ops.linear_bf16_hip(a, w, bias, gelu), 20, stream, warm=3)
        abytes = 2.0 * (M * K_ + N_ * K_ + M * N_) + 4.0 * N_  # A, W, C in bf16 + fp32 bias, each once
        form = "bias" + ("+GELU" if gelu else "")
        t = t_plain
        if fold:
            lin = nn.Linear(K_, N_).to(x.device)
            if N_ == 768:  # producer: residual epilogue
                xres = torch.randn(M, N_, device=x.device)
                t = hip_event_time(lambda: ops.linear_residual_(xres, a, lin, gamma), 20, stream, warm=3)
                abytes += 8.0 * M * N_ + 8.0 * M * (N_ // 256)  # + the fp32 residual stream read and written, the row partials
                form = "LayerScale residual on the fp32 stream + bf16 rows + row partial sums (EPI 5)"
                del xres
            else:  # consumer: LayerNorm in the epilogue
                _, st = ops.linear_residual_(torch.zeros(M, K_, device=x.device), a, nn.Linear(K_, K_).to(x.device), gamma)
                t = hip_event_time(lambda: ops.linear_lnfold(a, st, lin, norm, gelu=gelu), 20, stream, warm=3)
                abytes += 8.0 * M * (K_ // 256) + 4.0 * N_
                form = "LayerNorm applied in the epilogue" + ("+GELU (EPI 7)" if gelu else " (EPI 6)")
                del st
            del lin
        r = row("vit_linear_%s(M=%d,K=%d,N=%d; %s)" % (name, M, K_, N_, form), "mfma", 2.0 * M * K_ * N_, 1e12, 2500.0, "TFLOP/s", t)
        r["algorithmic_bytes"] = abytes
        r["hbm_gbps_algorithmic"] = abytes / t / 1e9
        r["plain_epilogue_us"] = t_plain * 1e6
        r["plain_epilogue_frac"] = 2.0 * M * K_ * N_ / t_plain / 1e12 / 2500.0
        shapes.append(r)
        flops += 2.0 * M * K_ * N_
        secs += t
        secs_plain += t_plain
        alg_bytes += abytes
        del a, w, bias
    traffic = _pmc_traffic_gemm(B, T, alg_bytes)
    gemm = dict(bound="mfma", kernel="gemm256_kernel<EPI, false, false> (bf16): the four linears of one ViT-B block at M=%d as the step runs them%s, flop-weighted"
                                      % (M, " (residual + LayerNorm folded into their epilogues)" if fold else ""),
                achieved=flops / secs / 1e12, peak=2500.0, unit="TFLOP/s", frac=flops / secs / 1e12 / 2500.0, traffic=traffic,
                launches=4, us=secs * 1e6, flop=flops, algorithmic_bytes=alg_bytes, shapes=shapes,
                plain_epilogue=dict(us=secs_plain * 1e6, achieved=flops / secs_plain / 1e12, frac=flops / secs_plain / 1e12 / 2500.0,
                                    what="the same four shapes with bias / bias + GELU epilogues only (round 5's launches; a forward then needs 23 separate "
                                         "residual + LayerNorm passes of 810 MB each: 2.9 ms per step)"),
                note="hand-written bf16 MFMA GEMM (256x256 tiles, half-tile LDS-DMA stream with counted waits, ping-pong wave groups, "
                     "persistent XCD-aware tile walk with dynamic tile tickets), bias / GELU in the epilogue; `achieved` = sum of the four shapes' flops / sum of their HIP-event launch times; "
                     "12 blocks x 4 launches per step")
    rows[:] = []
    # ViT patch attention: 4 T^2 64 flop per (image, head), 2B images x 12 heads
    qkv = torch.randn(2 * B, T, 2304, device=x.device).bfloat16()
    t = hip_event_time(lambda: ops.vit_attention(qkv, 12), 10, stream)
    vit = row("vit_attn_kernel(T=%d)" % T, "mfma", 2.0 * B * 12 * 4.0 * T * T * 64, 1e12, 2500.0, "TFLOP/s", t,
              "QK^T + PV flops; softmax exp/sum VALU work shares the issue port with the matrix core")
    vit["traffic"] = _pmc_traffic("vit_attn_kernel", B) if T == 1374 else None
    del qkv
    # the fp32-class linears (bf16 hi/lo split, 3 MFMAs per product): the reference's default precision
    if ops.USE_F32X3:
        for name, K_, N_, epi in (("qkv", 768, 2304, 0), ("fc1", 768, 3072, 1)):
            a_s = ops.split_f32(torch.randn(M, K_, device=x.device))
            w_s = ops.split_f32(torch.randn(N_, K_, device=x.device) / K_ ** 0.5)
            bias = torch.randn(N_, device=x.device)
            t = hip_event_time(lambda: ops.linear_f32x3(a_s, w_s, bias, M, N_, K_, gelu=bool(epi), out="split"), 5, stream)
            row("vit_linear_f32x3_%s(M=%d,K=%d,N=%d%s)" % (name, M, K_, N_, ",+bias+GELU" if epi else ""), "mfma",
                2.0 * M * K_ * N_, 1e12, 2500.0 / 3.0, "TFLOP/s", t,
                "fp32-equivalent flops against 1/3 of the bf16 dense peak (3 bf16 MFMAs per product)")
            del a_s, w_s, bias
        # the fp32-class ViT attention (split layout in and out, csrc/vit_attn_f32s.hip)
        qs = ops.split_f32(torch.randn(2 * B * T, 2304, device=x.device))
        t = hip_event_time(lambda: ops.vit_attention_f32_ss(qs, 2 * B, T, 12), 5, stream)
        row("vit_attn_f32s_kernel(T=%d)" % T, "mfma", 2.0 * B * 12 * 4.0 * T * T * 64, 1e12, 2500.0 / 3.0, "TFLOP/s", t,
            "fp32-equivalent flops against 1/3 of the bf16 dense peak (3 bf16 MFMAs per product in both contractions)")
        del qs
    # PE, S=256, bf16 hi/lo-split matrix cores; 20864 flop per neighbour row.  The kernels skip the 32-row tiles that hold nothing but the
    # ball query's padding (copies of the first neighbour cannot change the max-pool: bit-identical output), so `achieved` counts the rows
    # they really push through the MLP -- from the neighbour counts of these clouds -- and the reference's S rows per centre are reported
    # beside it (`algorithmic_rows_frac`): a fraction above what the matrix cores did would be meaningless
    _, (_, cnt) = ops.pe_group_mlp_max(x, pe.r2, pe.ns2, pe.mlp2, bf16x3=True, want_cand=True)
    cnt = torch.where(cnt < 0, torch.full_like(cnt, pe.ns2), cnt).clamp(max=pe.ns2)
    rows_done = float((((cnt + 31) // 32) * 32).clamp(max=pe.ns2).sum().item())
    t = hip_event_time(lambda: ops.pe_group_mlp_max(x, pe.r2, pe.ns2, pe.mlp2, bf16x3=True), 10, stream)
    r = row("pe_group_mlp_max_bf16x3_kernel(S=%d)" % pe.ns2, "mfma", rows_done * 20864.0, 1e12, 2500.0,
            "TFLOP/s", t, "fp32-equivalent flops of the neighbour rows the kernel computes (tiles of 32; %.1f of %d rows per centre on these clouds: the "
                          "rest is ball-query padding, skipped bit-identically); 3 bf16 MFMAs per product" % (rows_done / (B * N), pe.ns2))
    r["algorithmic_rows_frac"] = B * N * pe.ns2 * 20864.0 / t / 1e12 / 2500.0
    r["traffic"] = _pmc_traffic("pe_group_mlp_max_bf16x3_kernel", B)
    t = hip_event_time(lambda: ops.pe_group_mlp_max(x, pe.r2, pe.ns2, pe.mlp2, bf16x3=False), 5, stream)
    r = row("pe_group_mlp_max_kernel(S=%d, exact fp32 MFMA)" % pe.ns2, "mfma", rows_done * 20864.0, 1e12, 157.3,
            "TFLOP/s", t, "same row count (padding tiles skipped)")
    r["algorithmic_rows_frac"] = B * N * pe.ns2 * 20864.0 / t / 1e12 / 157.3
    # geometric embedding: 8 n^2 256^2 flop per cloud
    n = model.coarse_npoint + 1
    gp = torch.cat([torch.ones(B, 1, 3, device=x.device), x[:, :n - 1]], 1).contiguous()
    t = hip_event_time(lambda: ops.geo_embedding(gp, model.geo_embedding, out_dtype=torch.bfloat16), 10, stream)
    row("geo_embed_table_kernel<bf16>(4-point table interpolation) + geo_knn_kernel", "hbm", B * 512.0 * n * n, 1e9, 8000.0, "GB/s", t,
        "what the step runs: the (B, n, n, 256) bf16 result is written once (the algorithmic bytes); the kernel itself is VALU-issue-bound "
        "(16 ds_read_b128 + 64 v_fma_f32 + 20 v_readlane per pair and lane), not HBM-bound")
    ops.GEO_TABLE = False
    try:
        t = hip_event_time(lambda: ops.geo_embedding(gp, model.geo_embedding, out_dtype=torch.bfloat16), 10, stream)
    finally:
        ops.GEO_TABLE = True
    row("geo_embed_kernel<bf16>(matrix-core form, not on the step's path)", "mfma", B * 8.0 * n * n * 256 * 256, 1e12, 2500.0, "TFLOP/s", t)
    # correspondence-transformer attention on the 197-token coarse sequences, both clouds stacked (2B): the RPE self-attention streams the
    # geometric embedding E (B2,197,197,256) bf16 once -- its only HBM-sized operand; the cross-attention (no E) is a 39.7 MFLOP problem
    # per cloud and layer: launch / latency-sized
    import ctypes

    from unopose_amd._lib import call, ptr, stream_ptr

    B2 = 2 * B
    yq = torch.randn(B2, n, 1280, device=x.device).bfloat16()   # [q | q W_p (4 x 256)] rows as the model lays them out
    ykv = torch.randn(B2, n, 512, device=x.device).bfloat16()
    vt = torch.randn(B2, 256, 224, device=x.device).bfloat16()
    Eb = torch.randn(B2, n, n, 256, device=x.device).bfloat16()
    oa = torch.empty(B2, n, 256, device=x.device, dtype=torch.bfloat16)

    def attn(rpe):
        call("unopose_token_attention", ptr(yq), 1280, ptr(ykv), 512, ptr(vt), ctypes.c_void_p(yq.data_ptr() + 256 * 2) if rpe else None, 1280,
             ptr(Eb) if rpe else None, B2, n, n, 0.125, ptr(oa), stream_ptr())

    t = hip_event_time(lambda: attn(True), 10, stream)
    row("token_attn_kernel<rpe>(self-attention, %d clouds x %d tokens)" % (B2, n), "hbm", 2.0 * B2 * n * n * 256 + 2.0 * B2 * n * (1280 + 512 + 256), 1e9,
        8000.0, "GB/s", t, "streams the geometric embedding once; 6 launches per step")
    t = hip_event_time(lambda: attn(False), 10, stream)
    rows.append(dict(kernel="token_attn_kernel<cross>(%d clouds x %d tokens)" % (B2, n), bound="latency", us=t * 1e6,
                     tflops=B2 * 4.0 * n * n * 256 / t / 1e12, launches_per_step=12, ms_per_step=12 * t * 1e3,
                     note="the query<->reference cross-attention of the correspondence transformer: 4 n^2 256 flop per cloud = 2.5 GFLOP per launch, "
                          "latency-sized (no MFMA fraction quoted: a 197-token problem cannot fill the matrix pipe); 12 launches per step"))
    del yq, ykv, vt, Eb, oa
    # BASELINE configs[4]: 256 pose hypotheses per pair, weighted SVD / Procrustes over N correspondences each (model_utils.py:667-743; the
    # batched one-sided Jacobi on 3 x 3 covariances of csrc/geom.hip).  Problems = B x 256 per launch.
    for npts in (3, 196, 2048):
        Mh = B * 256
        src = torch.randn(Mh, npts, 3, device=x.device)
        ref = torch.randn(Mh, npts, 3, device=x.device)
        wgt = torch.rand(Mh, npts, device=x.device)
        t = hip_event_time(lambda: ops.weighted_procrustes(src, ref, wgt), 10, stream)
        rows.append(dict(kernel="weighted_procrustes(%d hypotheses x %d points)" % (Mh, npts), bound="latency" if npts < 512 else "hbm", us=t * 1e6,
                         problems_per_s=Mh / t, us_per_problem=t * 1e6 / Mh, gbps=Mh * npts * 28.0 / t / 1e9,
                         note="BASELINE configs[4] (256 hypotheses per pair, %d pairs): one 3x3 covariance + Jacobi SVD per hypothesis; 28 B per correspondence read once" % B))
        del src, ref, wgt
    # group_points (the reference's `_ext` gather): HBM-write-bound
    idx = _ext.ball_query(x, x, 0.2, 256)
    xt = x.transpose(1, 2).contiguous()
    t = hip_event_time(lambda: _ext.group_points(xt, idx), 20, stream)
    row("group_points_lds_kernel(S=256)", "hbm", 4.0 * B * (N * 256 + 3 * N * 256 + 3 * N), 1e9, 8000.0, "GB/s", t)
    t = hip_event_time(lambda: _ext.ball_query(x, x, 0.2, 256), 20, stream)
    rows.append(dict(kernel="ball_query_kernel(S=256)", bound="latency", us=t * 1e6, ns_per_centre=t * 1e9 / (B * N),
                     note="one wave per centre scans the cloud's LDS tile (ballot + mbcnt compaction): LDS / latency-bound, not an HBM stream"))
    tem = batch["tem1_pts"].float().contiguous()
    t = hip_event_time(lambda: _ext.furthest_point_sampling(tem, 2048), 3, stream)
    rows.append(dict(kernel="fps_kernel(5000->2048)", bound="latency", us=t * 1e6, us_per_iteration=t * 1e6 / 2047))
    return dict(roofline=gemm, roofline_other=rows)


def _pmc_summary():
    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", rnd + "_pmc_summary.json")
        if os.path.exists(path):
            return json.load(open(path)), "profiles/%s_pmc_summary.json" % rnd
    return None, None


def _pmc_traffic_gemm(B, T, alg_bytes):
    """HBM bytes of the four ViT linears (one launch each) from the committed PMC passes: per shape and summed, beside the
    algorithmic bytes; null when the batch / token count differ from the profiled ones (B=32, T=1374)."""
    doc, src = _pmc_summary()
    if doc is None or B != 32 or T != 1374 or "gemm_shapes" not in doc:
        return None
    per = {k: v["hbm_bytes_per_launch"] for k, v in doc["gemm_shapes"].items()}
    if sorted(per) != ["fc1", "fc2", "proj", "qkv"]:
        return None
    tot = sum(per.values())
    return dict(hbm_bytes=tot, launches=4, hbm_bytes_per_launch=tot / 4.0, per_shape=per, algorithmic_bytes=alg_bytes,
                ratio_to_algorithmic=tot / alg_bytes, source=src)


def _pmc_traffic(kernel, B):
    """HBM bytes per launch from the committed PMC passes (profiles/r0N_pmc_summary.json: FETCH_SIZE and
    WRITE_SIZE collected in separate rocprofv3 runs at B=32, FETCH_SIZE doubled per the gfx950 note);
    null when the batch differs from the profiled one."""
    doc, src = _pmc_summary()
    if doc is None or B != 32:
        return None
    k = doc["kernels"].get(kernel)
    return None if k is None else dict(hbm_bytes_per_launch=k["hbm_bytes_per_launch"], source=src)


def _cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_leg(img, batches=(1, 4), timed=2, all_cores=False):
    # (bounded to ~35 s of a default run: B = 1 gets one warm-up + `timed` forwards, the larger batch one warm-up + ONE timed forward --
    #  every number that can become `value` is a warmed one, first-touch allocations of a new batch size are not in it)
    """The oracle (torch-CPU port of the reference forward + the C `_ext` port; kind "port": the reference's
    Python cannot travel) timed on this host's cores on a bounded sample of the same workload
    (SURVEY.md 8(d)): per batch size one warm-up forward, then `timed` timed forwards; value = the best
    batch size's median rate.  The reported value uses torch's intra-op pool at min(32, cores) threads: more threads
    only add contention at these op sizes -- with all 256 hardware threads of the GPU box's host the same B=8 forward takes
    292 s instead of 26 s (0.027 vs 0.30 pairs/s; `--cpu-all-cores` repeats that measurement, the committed run is
    profiles/r03_bench_n1_bf16_s518_first_with_all_cores.json)."""
    import torch

    from oracle import unopose_ref as R
    from oracle.pointnet2_oracle import ext as oext
    from unopose_amd.synthetic import make_batch

    host = os.cpu_count() or 1
    threads = min(32, host)
    cfg = R.default_cfg()
    sd = R.random_state_dict(cfg, seed=0, img_size=img, tame=0.1)

    def run(b, n_threads, n_timed, warm=1):
        torch.set_num_threads(n_threads)
        ep, _, _ = make_batch(b, 2048, 5000, img, seed=1)
        rand = torch.rand(b, 18000, generator=torch.Generator().manual_seed(2))
        times = []
        with torch.no_grad():
            for i in range(warm + n_timed):
                t0 = time.perf_counter()
                R.unopose_forward(ep, sd, cfg, rand, oext)
                log("cpu_baseline B=%d threads=%d forward %d: %.2f s" % (b, n_threads, i, time.perf_counter() - t0))
                if i >= warm:
                    times.append(time.perf_counter() - t0)
        times.sort()
        med = times[len(times) // 2]
        return dict(pairs_per_s=b / med, median_s=med, min_s=times[0], max_s=times[-1], timed=n_timed, warmup=warm)

    by_batch = {str(b): run(b, threads, timed if b == min(batches) else 1) for b in sorted(batches)}
    best = max(by_batch.values(), key=lambda r: r["pairs_per_s"])
    out = dict(value=best["pairs_per_s"], unit="pairs/s", cores=threads, host_cores=host, kind="port",
               cpu_model=_cpu_model_name(), by_batch=by_batch,
               sample=f"batches of {list(batches)} pairs (2048 query / 5000 reference points, {img}x{img} crops): B={min(batches)} 1 warm-up + "
                      f"{timed} timed forwards, larger batches 1 warm-up + 1 timed forward; fp32, torch {threads} threads + C `_ext` port; value = best rate")
    if all_cores and host > threads:
        b = max(batches)
        r = run(b, host, 1)
        out["all_cores"] = dict(threads=host, batch=b, **r)
    return out


def _timed_steps(pipe, make_inputs, steps, warm, sync):
    """`warm` untimed + `steps` timed submits through a PipelinedForward, bracketed by synchronisations; -> (seconds, last output)."""
    for _ in range(warm):
        get = pipe.submit(make_inputs()).result
    pipe.drain()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        get = pipe.submit(make_inputs()).result
    out = get()
    pipe.drain()
    sync()
    return time.perf_counter() - t0, out


def contract_224_leg(dev, sync, steps=20, warm=3):
    """The reference's OWN test contract (configs/main_cfg.py:87-92,130-131 + save_unopose.sh): 224 x 224 crops, instance batch 16,
    2048 query / 5000 -> 2048 reference points, fp32 (`test.amp.enabled=False`, the default) and bf16 autocast.  A step is one
    forward over one instance batch, inputs resident; bounded (~1 s)."""
    import torch

    from unopose_amd.model import UNOPose, default_model_cfg
    from unopose_amd.pipeline import PipelinedForward
    from unopose_amd.synthetic import make_batch, trained_like_

    B = 16
    model = trained_like_(UNOPose(default_model_cfg(feature_extraction=dict(img_size=224)))).to(dev).eval()
    batch, R_gt, _ = make_batch(B, 2048, 5000, 224, seed=700, device=dev)
    batch["coarse_rand"] = torch.rand(B, 18000, device=dev)
    out = {"workload": "UNOPose.forward at the reference's test contract: instance batch 16, 224x224 crops, 2048 query pts, 5000->2048 "
                       "reference pts (configs/main_cfg.py:87-92,130-131)", "batch": B, "steps": steps, "warmup": warm}
    for name, amp, depth in (("fp32", None, 2), ("bf16", torch.bfloat16, 2)):
        pipe = PipelinedForward(model, depth=depth, autocast_dtype=amp)
        dt, o = _timed_steps(pipe, lambda: dict(batch), steps, warm, sync)
        pipe.close()
        err = (o["pred_R"] - R_gt).abs().amax(dim=(1, 2))
        out[name] = {"value": B * steps / dt, "unit": "pairs/s", "ms_per_step": dt / steps * 1e3, "forwards_in_flight": pipe.depth,
                     "median_rot_err_vs_gt": err.median().item()}
    del model, batch
    torch.cuda.empty_cache()
    return out


def ref_cached_leg(model, args, dev, sync, steps=20, warm=3, per_ref=8):
    """The step with the reference side served by `runner.ReferenceCache` (the reference's precomputed
    `dense_po` / `dense_fo` shortcut, oneref_feature_extraction.py:252-263): one reference view per `per_ref` queries, the views
    encoded once before the timed region (`UNOPose.encode_reference` through the pipeline), every timed step = cache lookup (hits) +
    forward over B pairs whose ViT sees the B query crops only.  Same model, batch size and crop side as the headline."""
    import torch

    from unopose_amd.pipeline import PipelinedForward
    from unopose_amd.runner import ReferenceCache
    from unopose_amd.synthetic import make_shared_reference_batch

    B = args.batch
    batch, keys, R_gt, _ = make_shared_reference_batch(B, per_ref, 2048, 5000, args.img, seed=800, device=dev)
    batch["coarse_rand"] = torch.rand(B, 18000, device=dev)
    pipe = PipelinedForward(model, depth=args.inflight, autocast_dtype=torch.bfloat16)
    cache = ReferenceCache(model)
    t0 = time.perf_counter()
    cache.lookup(keys, batch["tem1_rgb"], batch["tem1_choose"], batch["tem1_pts"], pipe.encode_reference)
    sync()
    encode_ms = (time.perf_counter() - t0) * 1e3
    query_side = {k: batch[k] for k in ("pts", "rgb", "rgb_choose", "coarse_rand")}

    def inputs():
        ep = dict(query_side)
        ep.update(cache.lookup(keys, batch["tem1_rgb"], batch["tem1_choose"], batch["tem1_pts"], pipe.encode_reference))
        return ep

    dt, o = _timed_steps(pipe, inputs, steps, warm, sync)
    pipe.close()
    err = (o["pred_R"] - R_gt).abs().amax(dim=(1, 2))
    return {"value": B * steps / dt, "unit": "pairs/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warm, "dtype": "bf16",
            "queries_per_reference": per_ref, "distinct_references": len(set(keys)), "cache_hits": cache.hits, "cache_misses": cache.misses,
            "encode_references_ms_once": encode_ms, "forwards_in_flight": pipe.depth,
            "median_rot_err_vs_gt": err.median().item(), "frac_pairs_solved(<0.05)": (err < 0.05).float().mean().item(),
            "workload": f"UNOPose.forward, batch {B}, {args.img}x{args.img} crops, reference features from runner.ReferenceCache "
                        f"(one view per {per_ref} queries; lookup inside the timed step, encoding outside)"}


def train_leg(dev, sync, steps=3, warm=2):
    """The training step of BASELINE configs[3] on ONE GPU, bounded (2 warm-up + 3 steps): 8 pairs, 4096 points per cloud, 224 x 224
    crops, frozen backbone, fp32, Adam -- forward in train mode + process_loss + backward + gradient hygiene + optimiser step
    (`bench.py --train` is the full-length form and the N > 1 DDP form)."""
    import torch

    from unopose_amd.model import UNOPose, default_model_cfg
    from unopose_amd.synthetic import make_train_batch, trained_like_
    from unopose_amd.train import build_optimizer, freeze_backbone, train_step

    B, npts, img = 8, 4096, 224
    nt = npts + npts // 2
    rng = torch.get_rng_state(), torch.cuda.get_rng_state(dev)
    torch.manual_seed(0)
    with torch.enable_grad():
        model = freeze_backbone(trained_like_(UNOPose(default_model_cfg(fine_npoint=npts, feature_extraction=dict(img_size=img)))).to(dev))
        batch = make_train_batch(B, npts, nt, img, seed=300, device=dev)
        opt, sched = build_optimizer(model, lr=1e-4, total_iters=188340)
        losses = []
        for _ in range(warm):
            losses.append(float(train_step(model, batch, opt, sched)["loss"]))
        sync()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        t0 = time.perf_counter()
        for a, b in evs:
            a.record()
            losses.append(float(train_step(model, batch, opt, sched)["loss"]))
            b.record()
        sync()
        dt = time.perf_counter() - t0
    per = sorted(a.elapsed_time(b) for a, b in evs)
    n_train = sum(p.numel() for p in model.parameters() if p.requires_grad)
    del model, batch, opt, sched
    torch.cuda.empty_cache()
    torch.set_rng_state(rng[0])
    torch.cuda.set_rng_state(rng[1], dev)
    return {"value": B * steps / dt, "unit": "pairs/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warm, "dtype": "fp32",
            "step_ms_hip_events": {"median": per[len(per) // 2], "min": per[0], "max": per[-1]}, "loss_first_last": [losses[0], losses[-1]],
            "workload": f"UNOPose training step (BASELINE configs[3], one rank): {B} pairs, {npts} query pts, {nt}->{npts} reference pts, "
                        f"{img}x{img} crops, frozen DINOv2 ViT-B/14, {n_train / 1e6:.1f} M trainable parameters, Adam"}


def train_main(args, world, rank, dev, sync):
    """bench.py --train: the training step of BASELINE configs[3]; one JSON line like the forward bench."""
    import torch
    import torch.distributed as dist

    from unopose_amd.model import UNOPose, default_model_cfg
    from unopose_amd.synthetic import make_train_batch, trained_like_
    from unopose_amd.train import build_optimizer, freeze_backbone, train_step, wrap_ddp

    if os.environ.get("UNOPOSE_TRAIN_NO_MIOPEN") == "1":  # A/B: torch's native BatchNorm / convolution kernels instead of MIOpen's
        torch.backends.cudnn.enabled = False
    B = args.batch if args.batch != 32 else 8  # --batch's default is the forward bench's; configs[3] is 8 pairs per rank
    img = args.img if args.img != 518 else 224  # the reference trains on 224 x 224 crops (configs/main_cfg.py:190)
    npts = args.train_pts
    nt = 5000 if npts <= 2048 else npts + npts // 2
    torch.manual_seed(0)
    cfg = default_model_cfg(fine_npoint=npts, feature_extraction=dict(img_size=img))
    model = freeze_backbone(trained_like_(UNOPose(cfg)).to(dev))
    batch = make_train_batch(B, npts, nt, img, seed=300 + rank, device=dev)
    net = wrap_ddp(model, dev) if world > 1 else model
    opt, sched = build_optimizer(model, lr=1e-4, total_iters=188340)
    amp = torch.bfloat16 if args.train_dtype == "bf16" else None
    losses = []

    def step():
        info = train_step(net, batch, opt, sched, amp_dtype=amp)
        losses.append(info["loss"])

    log("model + batch ready (rank %d of %d)" % (rank, world))
    for _ in range(args.warmup):
        step()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt_local = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    per_rank = None
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
        mine = torch.tensor([dt_local / args.steps * 1e3], device=dev, dtype=torch.float64)
        every = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [float(e.item()) for e in every]
    n_train = sum(p.numel() for p in model.parameters() if p.requires_grad)
    res = {
        "metric": "(query,ref) pairs/sec training step",
        "value": world * B * args.steps / dt,
        "unit": "pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.train_dtype,
        "data": "synthetic",
        "config": {"workload": f"UNOPose training step (BASELINE configs[3]), {B} pairs/GPU, {npts} query pts, {nt}->{npts} reference pts, "
                               f"196 coarse pts, {img}x{img} crops, frozen DINOv2 ViT-B/14 reg4 (fused inference kernels, no_grad), "
                               f"{n_train / 1e6:.1f} M trainable parameters, Adam(1e-4, (0.5, 0.999), 1e-6), flat-and-anneal schedule",
                   "sharding": f"dp{world} (replicas, bucketed gradient all-reduce over RCCL overlapped with backward)" if world > 1 else "dp1"},
        "loss_first_last": [float(losses[0]), float(losses[-1])],
    }
    if per_rank is not None:
        res["per_rank_ms_per_step"] = per_rank
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))  # nothing above has touched the GPU
    world = int(env_world or "1")
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a {world}-rank run as {args.gpus} GPUs")

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # UNOPOSE_BENCH_BACKEND=gloo is a TEST hook: it lets the N>1 code path run with several ranks sharing one
    # GPU (collectives staged through host memory); the driver's runs use the default, RCCL over xGMI.
    # Keep such shared-GPU runs short and at --img 224: concurrent stream-K library GEMMs of two processes can
    # starve each other's workgroups.
    backend = "gloo" if args.dry_run else os.environ.get("UNOPOSE_BENCH_BACKEND", "nccl")
    if args.dry_run:
        dev = comm_dev = torch.device("cpu")
    else:
        local = local if backend == "nccl" else local % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        comm_dev = dev if backend == "nccl" else torch.device("cpu")
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI
        else:
            dist.init_process_group(backend)

    def sync():
        if not args.dry_run:
            torch.cuda.synchronize()

    if args.train:
        return train_main(args, world, rank, dev, sync)
    torch.set_grad_enabled(False)
    torch.manual_seed(0)
    B = args.batch
    amp = args.dtype == "bf16"
    if args.dry_run:
        model = torch.nn.Linear(9, 9)
        R_gt = torch.eye(3).expand(B, 3, 3).contiguous()
        batch = {"pts": torch.zeros(B, 8, 3)}

        def forward(ep, use_amp):
            ep.update(pred_R=R_gt + 0 * model.weight.sum(), pred_t=torch.zeros(B, 3), pred_pose_score=torch.ones(B))
            return ep
    else:
        from unopose_amd.model import UNOPose, default_model_cfg
        from unopose_amd.synthetic import make_batch, trained_like_

        cfg = default_model_cfg(feature_extraction=dict(img_size=args.img))
        model = trained_like_(UNOPose(cfg)).to(dev).eval()
        batch, R_gt, t_gt = make_batch(B, 2048, 5000, args.img, seed=100 + rank, device=dev)
        batch["coarse_rand"] = torch.rand(B, 18000, device=dev)

        def forward(ep, use_amp):
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=use_amp):
                return model(ep)
    if world > 1:  # one flat broadcast of the weights from rank 0 over RCCL / xGMI
        from unopose_amd.runner import broadcast_module_

        broadcast_module_(model, 0)  # first call: communicator set-up included
        sync()
        dist.barrier()
        sync()
        tb = time.perf_counter()
        broadcast_module_(model, 0)
        sync()
        bcast_ms = (time.perf_counter() - tb) * 1e3

    graphed = None
    if args.graph and not args.dry_run:
        from unopose_amd.graph import GraphedForward

        graphed = GraphedForward(model, batch, torch.bfloat16 if amp else None)

    # consecutive steps go through unopose_amd.pipeline.PipelinedForward: with --inflight 2 (default on the bf16 path) the
    # matcher of step i runs underneath the ViT of step i+1 on a second HIP stream.  A step is still one whole forward
    # over one batch, all K of them start and finish inside the timed region.
    pipe = None
    if not args.dry_run and graphed is None:
        from unopose_amd.pipeline import PipelinedForward

        pipe = PipelinedForward(model, depth=args.inflight, autocast_dtype=torch.bfloat16 if amp else None, timing=True,
                                stages="auto" if args.stages == "auto" else bool(int(args.stages)))
        args.inflight = pipe.depth

    def step(use_amp=amp):
        """-> a callable returning the end_points of this step (host-visible after the final synchronize)."""
        ep = dict(batch)
        if graphed is not None:  # inputs copied into the graph's static buffers, one hipGraphLaunch
            o = graphed(ep)
            return lambda: o
        if pipe is not None and use_amp == amp:
            return pipe.submit(ep).result
        o = forward(ep, use_amp)
        return lambda: o

    log("model + batch ready (rank %d of %d)" % (rank, world))
    for _ in range(args.warmup):
        out = step()
    if pipe is not None:
        pipe.drain()
    sync()
    log("warm-up done")
    if world > 1:
        dist.barrier()
    sync()
    tickets_from = len(pipe.history) if pipe is not None else 0
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step()
    out = out()  # joins the stream of the last step to the current one
    if pipe is not None:
        pipe.drain()
    poses = torch.cat([out["pred_R"].reshape(B, 9), out["pred_t"], out["pred_pose_score"].reshape(B, 1)], 1)
    sync()
    dt_local = time.perf_counter() - t0  # this rank's own K steps, before it meets the others
    if world > 1:  # gather of poses to rank 0 (the reference lacks it: every rank writes the same file)
        poses = poses.to(comm_dev)
        gathered = [torch.empty_like(poses) for _ in range(world)] if rank == 0 else None
        dist.gather(poses, gathered, 0)
    sync()
    gather_ms = (time.perf_counter() - t0 - dt_local) * 1e3  # includes waiting for the slowest rank
    if world > 1:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=comm_dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
        mine = torch.tensor([dt_local / args.steps * 1e3, gather_ms, bcast_ms], device=comm_dev, dtype=torch.float64)
        every = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [[float(v) for v in e.tolist()] for e in every]

    log("timed region done: %.2f ms/step" % (dt / args.steps * 1e3))
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
        "launch": "hipGraph replay" if graphed is not None else "eager",
        "forwards_in_flight": args.inflight,
        "pipeline": None if pipe is None or pipe.depth == 1 else ("ViT half of step i+1 beside the matcher half of step i (two streams)" if pipe.last_mode == "stages" else "two whole forwards side by side (two streams)"),
        "sanity": {"median_rot_err_vs_gt": rot_err.median().item(),
                   "frac_pairs_solved(<0.05)": (rot_err < 0.05).float().mean().item()},
    }
    if world > 1:
        res["per_rank_ms_per_step"] = [r[0] for r in per_rank]
        res["collectives"] = {"backend": "RCCL (nccl)" if backend == "nccl" else backend,
                              "weights_broadcast_ms": max(r[2] for r in per_rank),
                              "poses_gather_ms_per_rank": [r[1] for r in per_rank],
                              "what": "broadcast: second flat broadcast of all parameters + buffers from rank 0 (communicator already up), outside "
                                      "the timed region; gather: the one pose gather that closes the timed region (includes the wait for the "
                                      "slowest rank); no collective on the data path"}
    if args.dry_run:
        res["dry_run"] = True
    if pipe is not None:
        # HIP events of every step: start / end on the stream it ran on.  `step_ms_hip_events` = time between the completions of
        # steps i and i + depth (two consecutive steps of ONE stream), divided by depth: the per-step period of the pipeline
        # (with one forward in flight: the step time).  `forward_latency_ms` = start -> end of one forward, which is longer than
        # a step when two overlap.
        hist = pipe.history[tickets_from:]
        q = lambda v, f: v[min(len(v) - 1, int(round(f * (len(v) - 1))))]  # noqa: E731
        d = pipe.depth
        gaps = sorted(hist[i][1].elapsed_time(hist[i + d][1]) / d for i in range(len(hist) - d))
        lat = sorted(a.elapsed_time(b) for a, b in hist)
        if gaps:
            res["step_ms_hip_events"] = {"median": q(gaps, 0.5), "p10": q(gaps, 0.1), "p90": q(gaps, 0.9), "min": gaps[0], "max": gaps[-1],
                                         "pairs_per_s_at_median": world * B / q(gaps, 0.5) * 1e3,
                                         "what": "(completion of step i + depth) - (completion of step i), / depth"}
        res["forward_latency_ms"] = {"median": q(lat, 0.5), "p10": q(lat, 0.1), "p90": q(lat, 0.9)}
    if rank == 0 and world == 1 and not args.dry_run:
        if amp and not args.no_fp32 and graphed is None:
            # the reference's default precision (configs/main_cfg.py:87-89: test.amp.enabled=False) through the same runner object as
            # `--dtype fp32` (PipelinedForward without autocast -- pipelined like the bf16 path since round 6: every fp32 GEMM is an own
            # kernel): 3 warm-ups, 20 steps, the HIP-event step period (median / p10 / p90) beside the wall-clock rate
            k = 20 if args.steps >= 20 else max(3, args.steps)
            if pipe is not None:
                pipe.close()
            from unopose_amd.pipeline import PipelinedForward

            pipe32 = PipelinedForward(model, depth=args.inflight, autocast_dtype=None, timing=True)
            d32, o32 = _timed_steps(pipe32, lambda: dict(batch), k, 3, sync)
            e32 = (o32["pred_R"] - R_gt).abs().amax(dim=(1, 2))
            h32, d_ = pipe32.history[-k:], pipe32.depth
            per = sorted(h32[i][1].elapsed_time(h32[i + d_][1]) / d_ for i in range(len(h32) - d_))  # completion-to-completion period, as for the headline
            q32 = lambda f: per[min(len(per) - 1, int(round(f * (len(per) - 1))))]  # noqa: E731
            pipe32.close()
            log("fp32 leg done")
            res["fp32"] = {"value": B * k / d32, "unit": "pairs/s", "ms_per_step": d32 / k * 1e3, "steps": k, "warmup": 3,
                           "step_ms_hip_events": {"median": q32(0.5), "p10": q32(0.1), "p90": q32(0.9)},
                           "forwards_in_flight": d_, "median_rot_err_vs_gt": e32.median().item(),
                           "arithmetic": "bf16x3 hi/lo split on the matrix cores with fp32 accumulation (hi.hi + hi.lo + lo.hi: ~2^-17 relative per "
                                         "product, not IEEE fp32's 2^-24) in every linear layer and attention contraction; geometry (FPS, ball query, "
                                         "frames, Procrustes, assignment statistics) and the small contractions of bmm_f32 in exact fp32; the "
                                         "geometric embedding's two projections of sinusoids by 6-point Lagrange interpolation (fp32 arithmetic) on "
                                         "fp32 tables of the projected functions (interpolation error ~1e-6 of the weights' scale)"}
        if not args.no_extra and graphed is None and amp:
            if pipe is not None:
                pipe.close()
            # bounded, driver-visible legs at the reference's own contract (each ~1-2 s): see the three functions
            res["ref_cached"] = ref_cached_leg(model, args, dev, sync)
            log("ref_cached leg done")
            res["contract_224"] = contract_224_leg(dev, sync)
            log("contract_224 leg done")
            res["train"] = train_leg(dev, sync)
            torch.set_grad_enabled(False)
            log("train leg done")
        if not args.no_roofline:
            res.update(roofline_leg(model, batch, args.img))
            log("roofline leg done")
        if not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline_leg(args.img, all_cores=args.cpu_all_cores)
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
