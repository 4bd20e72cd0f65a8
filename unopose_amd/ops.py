"""Fused geometry / attention / pose-head operators (C ABI part 2) on torch tensors.

Each function cites the reference Python it replaces.  Inputs must be CUDA
float32 tensors; there is no CPU path (RuntimeError).
"""
import ctypes
import itertools
import os

import torch

from ._lib import call, check_f32, lib, ptr, stream_ptr


def _c(x):
    return x if x.is_contiguous() else x.contiguous()


# ---- differentiable mode (training) -------------------------------------------------------------
# The fused HIP operators are inference kernels (no autograd).  Inside `with ops.differentiable():` every dispatcher
# below takes its op-by-op torch composite instead -- the same formulas on the GPU, recorded by autograd -- while
# index-producing kernels (FPS, ball query, LRF frames), whose outputs carry no gradient in the reference either,
# keep running on HIP.  UNOPose.forward enters this mode when `model.training`.
_DIFF = False


class differentiable:
    def __init__(self, on=True):
        self.on = on

    def __enter__(self):
        global _DIFF
        self.prev, _DIFF = _DIFF, bool(self.on)
        return self

    def __exit__(self, *a):
        global _DIFF
        _DIFF = self.prev
        return False


def is_differentiable():
    return _DIFF


def lrf_global(pts, use_ref_rad=False):
    """get_batch_lrf (oneref_grf_predator_pose_estimation_model.py:78-93). (B,N,3)->(B,N,3)."""
    pts = _c(pts.float())
    check_f32(pts, "pts")
    B, N, _ = pts.shape
    out = torch.empty_like(pts)
    with torch.cuda.device(pts.device):
        call("unopose_lrf_global", ptr(pts), B, N, int(bool(use_ref_rad)), ptr(out), stream_ptr())
    return out


def query_lrf_group(xyz, radius, nsample):
    """QueryAndLRFGroup(radius, nsample, use_xyz=True)(xyz, xyz, feats) (pointnet2_utils.py:522-584).
    (B,N,3) -> (B,6,N,nsample)."""
    xyz = _c(xyz.float())
    check_f32(xyz, "xyz")
    B, N, _ = xyz.shape
    out = torch.empty(B, 6, N, int(nsample), dtype=torch.float32, device=xyz.device)
    with torch.cuda.device(xyz.device):
        call("unopose_query_lrf_group", ptr(xyz), B, N, float(radius), int(nsample), ptr(out), stream_ptr())
    return out


def lrf_group_idx(xyz, new_xyz, idx, radius):
    """The general form of QueryAndLRFGroup.forward (pointnet2_utils.py:548-565): neighbour lists `idx` (B,N,S) int32 given
    (ball_query around `new_xyz`, possibly re-drawn by sample_uniformly) -> (B,6,N,S) with channels 0-2 relative to
    new_xyz and the frame / channels 3-5 relative to xyz, as the reference computes them."""
    xyz, new_xyz = _c(xyz.float()), _c(new_xyz.float())
    check_f32(xyz, "xyz")
    check_f32(new_xyz, "new_xyz")
    if new_xyz.shape != xyz.shape:
        raise ValueError(f"QueryAndLRFGroup: LRF_batch(xyz, grouped) needs npoint == N, got xyz {tuple(xyz.shape)} "
                         f"new_xyz {tuple(new_xyz.shape)} (pointnet2_utils.py:432-436)")
    idx = _c(idx.to(torch.int32))
    B, N, _ = xyz.shape
    if idx.shape[:2] != (B, N) or not idx.is_cuda:
        raise ValueError(f"idx must be a (B,N,S) device tensor, got {tuple(idx.shape)}")
    S = idx.shape[2]
    out = torch.empty(B, 6, N, S, dtype=torch.float32, device=xyz.device)
    with torch.cuda.device(xyz.device):
        call("unopose_lrf_group_idx", ptr(xyz), ptr(new_xyz), ptr(idx), B, N, float(radius), S, ptr(out), stream_ptr())
    return out


def weighted_procrustes(src, ref, weights=None, weight_thresh=0.0, eps=1e-5):
    """weighted_procrustes (utils/model_utils.py:667-743): R (M,3,3), t (M,3), ref ~ R src + t."""
    src, ref = _c(src.float()), _c(ref.float())
    check_f32(src, "src_points")
    check_f32(ref, "ref_points")
    M, N, _ = src.shape
    if weights is not None:
        weights = _c(weights.float())
        check_f32(weights, "weights")
    R = torch.empty(M, 3, 3, dtype=torch.float32, device=src.device)
    t = torch.empty(M, 3, dtype=torch.float32, device=src.device)
    with torch.cuda.device(src.device):
        call("unopose_weighted_procrustes", ptr(src), ptr(ref), ptr(weights) if weights is not None else None, M, N,
             float(weight_thresh), float(eps), ptr(R), ptr(t), stream_ptr())
    return R, t


# =============================================================================
# Dense-math operators.  Every GEMM of the eval path -- bf16 under autocast (csrc/gemm.hip, gemm_small.hip), fp32-class
# without (csrc/gemm_f32.hip, bmm_f32.hip) -- and everything with structure (attention, sampling, assignment, pose hypotheses) is a
# hand-written HIP kernel behind the C ABI; the `*_torch` composites below are the training path's differentiable forms and the
# fall-backs for shapes the kernels refuse (each warns once through note_fallback).
# =============================================================================

import torch.nn.functional as F

from .pointnet2 import _ext


# Module switches below are plain attributes (no environment routes in the product module): same-box A/B scripts under scripts/ set
# them on the imported module (e.g. `ops.USE_HIP_GEMM = False` sends the linears to the library).
# Own bf16 GEMM (csrc/gemm.hip + gemm_small.hip) for the linears.
USE_HIP_GEMM = True
# every ViT-sized linear on csrc/gemm.hip (no library stream-K kernels, i.e. no kernel with inter-workgroup waits, on the path)
# (measured 0.6 % behind the library at one forward in flight; it is what makes two forwards in flight safe: pipeline.py)
HIP_GEMM_ALL = True


_fallbacks_seen = set()


def note_fallback(site, why):
    """A shape / configuration the hand-written kernels do not take runs the torch composite instead -- still on the GPU, never
    silently: one RuntimeWarning per (site, reason) names it."""
    key = (site, why)
    if key not in _fallbacks_seen:
        _fallbacks_seen.add(key)
        import warnings

        warnings.warn(f"unopose_amd.ops.{site}: {why} -> torch composite (not a hand-written kernel)", RuntimeWarning, stacklevel=3)


def linear_backend():
    """Which GEMM runs the large bf16 linears (reported by bench.py next to the measured rate)."""
    if USE_HIP_GEMM and HIP_GEMM_ALL:
        return "csrc/gemm.hip (256x256x64 LDS-DMA tiles, persistent; bias / bias + erf-GELU epilogue) for every ViT linear"
    if USE_HIP_GEMM:
        return "hipBLASLt (through torch); fc1 + GELU: csrc/gemm.hip (256x256x64 LDS-DMA tiles, fused bias + erf-GELU epilogue)"
    return "hipBLASLt (through torch)"


def _params_key(mod, *extra):
    """Cache key of state derived from a module's tensors: version and address of EVERY parameter and buffer of the module, so that an
    in-place edit of any of them -- a bias alone included -- rebuilds the derived state."""
    return tuple((t._version, t.data_ptr()) for t in itertools.chain(mod.parameters(), mod.buffers())) + extra


def _bf16_weights(lin):
    key = _params_key(lin)
    cache = getattr(lin, "_bf16_cache", None)
    if cache is None or cache[0] != key:
        with torch.no_grad():
            b32 = torch.zeros(lin.weight.shape[0], device=lin.weight.device) if lin.bias is None else lin.bias.detach().float().contiguous()
            cache = (key, lin.weight.detach().to(torch.bfloat16).contiguous(),
                     None if lin.bias is None else lin.bias.detach().to(torch.bfloat16).contiguous(), b32)
        lin._bf16_cache = cache
    return cache


def linear_bf16_hip(x2, w, bias_f32, gelu=False, relu=False):
    """C-ABI unopose_linear_bf16: (M,K) bf16 @ (N,K)^T bf16 + bias fp32 [-> exact GELU | ReLU] -> (M,N) bf16."""
    M, K = x2.shape
    N = w.shape[0]
    out = torch.empty(M, N, dtype=torch.bfloat16, device=x2.device)
    with torch.cuda.device(x2.device):
        call("unopose_linear_bf16", ptr(x2), ptr(w), ptr(bias_f32), ptr(out), M, N, K, 1 if gelu else (2 if relu else 0), stream_ptr())
    return out


def own_gemm_ok(rows, N, K):
    """Does csrc/gemm.hip take this bf16 linear?  With HIP_GEMM_ALL: every shape its tiling admits (N % 256 == 0,
    K % 64 == 0, any row count) -- NO hipBLASLt bf16 kernel is left on the autocast path.  That matters beyond speed: the
    library's stream-K kernels spin on partner workgroups and hang when forwards overlap.  (Round 2 also
    blamed them for wrong sums in other kernels; round 3 traced those to packed-fp32 instructions in the VICTIM kernels beside any
    MFMA kernel -- the library is built without them now, see build.py.)"""
    if not USE_HIP_GEMM or N % 256 != 0 or K % 64 != 0 or rows < 1:
        return False
    return HIP_GEMM_ALL or rows >= 4096


def bf16_linear_2d(x2, w, bias_f32, bias_bf16=None, relu=False):
    """(rows,K) bf16 @ (N,K)^T + bias on csrc/gemm.hip when `own_gemm_ok`, else the library."""
    rows, K = x2.shape
    N = w.shape[0]
    if own_gemm_ok(rows, N, K):
        return linear_bf16_hip(_c(x2), w, bias_f32, relu=relu)
    y = F.linear(x2, w, bias_bf16 if bias_bf16 is not None else bias_f32.to(torch.bfloat16))
    return F.relu(y) if relu else y


# fp32 linears (the reference's default precision) on csrc/gemm_f32.hip: hi / lo-split bf16 operands, 3 MFMAs per product
# (fp32-class accuracy); `ops.USE_F32X3 = False` routes them back to the library SGEMM (A/B attribute).
USE_F32X3 = True


def f32x3_ok(rows, N, K):
    # (operands AND the output: the kernel addresses all three through 32-bit buffer offsets)
    return USE_F32X3 and N % 256 == 0 and K % 32 == 0 and rows >= 1 and rows * K * 4 < 2 ** 32 and N * K * 4 < 2 ** 32 and rows * N * 4 < 2 ** 32


_SPLIT_MEMO = []  # [(key, source tensor, split tensor)], newest first
_MUTATION_EPOCH = [0]  # bumped by every wrapper that writes a tensor through its raw pointer (in place, `out=`): see split_f32


def note_mutation():
    """A kernel is about to write an existing tensor through `ptr()` (torch's version counter does not see that): results remembered
    for tensors of an earlier epoch are dropped."""
    _MUTATION_EPOCH[0] += 1
    _SPLIT_MEMO.clear()


def clear_split_memo():
    """end of a forward: the remembered operands (and the source tensors they pin) are released"""
    _SPLIT_MEMO.clear()


def split_f32(x2, memo=False):
    """(M,K) fp32 -> the split layout of csrc/gemm_f32.hip (per row and 32-k block one 128-byte line [hi | lo] of bf16):
    returned as an (M, 2K) bf16 tensor (same bytes as the fp32 matrix).
    `memo=True` (the token projections of the matcher: the same tensor is projected two or three times, q / kv, k / v -- 49 of the 117
    splits of a forward, scripts/split_census.py): the last two SMALL results are remembered, keyed on storage, shape, strides, stream,
    torch's version counter (inference tensors have none and are not remembered) AND the mutation epoch -- every wrapper that writes a tensor through a raw
    pointer calls note_mutation(), which empties the memo; the model empties it at the end of each forward half as well.  The entry holds
    the source tensor, so its address cannot be handed to another tensor while the entry lives."""
    M, K = x2.shape
    memo = memo and not torch.is_grad_enabled() and M * K <= (8 << 20) and not x2.is_inference()  # (no version counter on inference tensors)
    if memo:
        ver = x2._version
        key = (x2.data_ptr(), M, K, x2.stride(), ver, _MUTATION_EPOCH[0], torch.cuda.current_stream(x2.device).cuda_stream)
        for e in _SPLIT_MEMO:
            if e[0] == key:
                return e[2]
    out = torch.empty(M, 2 * K, dtype=torch.bfloat16, device=x2.device)
    with torch.cuda.device(x2.device):
        call("unopose_split_bf16x2", ptr(x2), M, K, ptr(out), stream_ptr())
    if memo:
        _SPLIT_MEMO.insert(0, (key, x2, out))
        del _SPLIT_MEMO[2:]
    return out


def _f32x3_weights(lin):
    key = (lin.weight._version, lin.weight.data_ptr(), lin.weight.device, None if lin.bias is None else lin.bias._version)
    cache = getattr(lin, "_f32x3_cache", None)
    if cache is None or cache[0] != key:
        with torch.no_grad():
            w = lin.weight.detach().float().contiguous()
            b32 = torch.zeros(w.shape[0], device=w.device) if lin.bias is None else lin.bias.detach().float().contiguous()
            cache = (key, split_f32(w), b32)
        lin._f32x3_cache = cache
    return cache


def linear_f32x3(xs, ws, bias_f32, M, N, K, gelu=False, relu=False, out="f32"):
    """C-ABI unopose_linear_f32x3 on split operands; `out`: "f32" -> (M,N) fp32, "split" -> (M,2N) split layout, "both"."""
    C = torch.empty(M, N, dtype=torch.float32, device=xs.device) if out in ("f32", "both") else None
    Cs = torch.empty(M, 2 * N, dtype=torch.bfloat16, device=xs.device) if out in ("split", "both") else None
    with torch.cuda.device(xs.device):
        call("unopose_linear_f32x3", ptr(xs), ptr(ws), ptr(bias_f32), None if C is None else ptr(C), None if Cs is None else ptr(Cs),
             M, N, K, 1 if gelu else (2 if relu else 0), stream_ptr())
    return C if out == "f32" else (Cs if out == "split" else (C, Cs))


def linear_f32x3_bf16(xs, ws, bias_f32, M, N, K, resid=None):
    """C-ABI unopose_linear_f32x3_bf16: bf16( resid + bf16(X W^T + b) ) with X, W in the split layout."""
    out = torch.empty(M, N, dtype=torch.bfloat16, device=xs.device)
    with torch.cuda.device(xs.device):
        call("unopose_linear_f32x3_bf16", ptr(xs), ptr(ws), ptr(bias_f32), None if resid is None else ptr(resid), ptr(out), M, N, K, stream_ptr())
    return out


def linear_f32_raw(x, w, b, owner, tag):
    """x (...,K) fp32 @ w (N,K)^T + b with cached split weights on `owner` (the fp32 token / linear attention projections, whose
    fused weight matrices are built by their callers); csrc/gemm_f32.hip when the shape fits, else the library."""
    N, K = w.shape
    rows = x.numel() // K
    if not (x.is_cuda and not _DIFF and f32x3_ok(rows, N, K)):
        return F.linear(x, w, b)
    key = (w.data_ptr(), w._version, None if b is None else (b.data_ptr(), b._version), tag)
    caches = owner.__dict__.setdefault("_f32x3_raw", {})
    c = caches.get(tag)
    if c is None or c[0] != key:
        with torch.no_grad():
            c = (key, split_f32(w.detach().float().contiguous()), torch.zeros(N, device=w.device) if b is None else b.detach().float().contiguous())
        caches[tag] = c
    return linear_f32x3(split_f32(_c(x.float()).reshape(rows, K), memo=True), c[1], c[2], rows, N, K).reshape(*x.shape[:-1], N)


# ---- two-way InfoNCE loss of the matchers (loss_utils.py:181-187) on the streaming softmax statistics ---------------------------
USE_FUSED_INFONCE = True  # A/B attribute: False = two F.cross_entropy calls


class _InfoNCEFn(torch.autograd.Function):
    """atten (B,R,C) fp32, label1 (B,R-1), label2 (B,C-1) int64 -> (B,) loss
    0.5 (mean_i CE(row i >= 1 over all columns, label1) + mean_j CE(column j >= 1 over all rows, label2)).
    Forward: two statistics passes over the matrix (csrc/posehead.hip, the eval path's kernels) + gathers of the labelled
    entries; backward: ONE pass writing the gradient.  torch's log_softmax over a non-last dimension of the 4097 x 4097 fine
    similarity ran at 0.36 TB/s and was 19 % of the training step."""

    @staticmethod
    def forward(ctx, atten, label1, label2):
        B, R, C = atten.shape
        x = _c(atten.float())
        ws = torch.empty(2 * B * (R + C), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            call("unopose_softmax_stats", ptr(x), B, R, C, ptr(ws), stream_ptr())
        rmax, rinv = ws[:B * R].reshape(B, R), ws[B * R:2 * B * R].reshape(B, R)
        cmax, cinv = ws[2 * B * R:2 * B * R + B * C].reshape(B, C), ws[2 * B * R + B * C:].reshape(B, C)
        lse_r = rmax[:, 1:] - torch.log(rinv[:, 1:])
        lse_c = cmax[:, 1:] - torch.log(cinv[:, 1:])
        picked_r = torch.gather(x[:, 1:, :], 2, label1.unsqueeze(2)).squeeze(2)          # x[i, label1[i-1]], i >= 1
        picked_c = torch.gather(x[:, :, 1:], 1, label2.unsqueeze(1)).squeeze(1)          # x[label2[j-1], j], j >= 1
        ctx.save_for_backward(x, ws, label1, label2)
        return 0.5 * ((lse_r - picked_r).mean(1) + (lse_c - picked_c).mean(1))

    @staticmethod
    def backward(ctx, g):
        x, ws, label1, label2 = ctx.saved_tensors
        B, R, C = x.shape
        grad = torch.empty_like(x)
        with torch.cuda.device(x.device):
            call("unopose_infonce_grad", ptr(x), B, R, C, ptr(ws), ptr(_c(label1)), ptr(_c(label2)), ptr(_c(g.float())), ptr(grad), stream_ptr())
        return grad, None, None


def infonce_two_way(atten, label1, label2):
    """The "atten" loss of compute_overlap_loss for one transformer block -> (B,)."""
    if USE_FUSED_INFONCE and atten.is_cuda and atten.shape[1] <= 65535:
        return _InfoNCEFn.apply(atten, label1, label2)
    a = atten.float()
    l1 = F.cross_entropy(a.transpose(1, 2)[:, :, 1:], label1, reduction="none").mean(1)  # classes = columns, per query row
    l2 = F.cross_entropy(a[:, :, 1:], label2, reduction="none").mean(1)
    return 0.5 * (l1 + l2)


# ---- trainable linears: forward and input gradient on the hand-written GEMMs, recorded by autograd ------
USE_FUSED_BN_RELU = True  # A/B attribute: False = nn.BatchNorm2d (MIOpen) + F.relu in the PE's SharedMLP under train()


class _BNReLUTrain(torch.autograd.Function):
    """relu(batch_norm(x)) with BATCH statistics (nn.BatchNorm2d in train mode + ReLU, pytorch_utils.py:25-132) on csrc/bn_train.hip:
    forward = statistics pass + apply pass (running statistics updated in the statistics kernel), backward = reduction pass + apply
    pass with the ReLU mask recomputed from x; saved for backward: x, mean, rstd (not y, not the mask)."""

    @staticmethod
    def forward(ctx, x, weight, bias, bn):
        B, C = x.shape[:2]
        L = x.numel() // (B * C)
        x = _c(x)
        chunk = lib().unopose_bn_train_chunk()
        ws = torch.empty(2 * B * C * ((L + chunk - 1) // chunk), dtype=torch.float32, device=x.device)
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        y = torch.empty_like(x)
        track = bn.track_running_stats and bn.running_mean is not None
        if track:
            bn.num_batches_tracked.add_(1)
            momentum = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked.item())
        else:
            momentum = 0.0
        w, b_ = _c(weight.detach().float()), _c(bias.detach().float())
        with torch.cuda.device(x.device):
            call("unopose_bn_relu_train_forward", ptr(x), B, C, L, ptr(w), ptr(b_), float(bn.eps), float(momentum),
                 ptr(bn.running_mean) if track else None, ptr(bn.running_var) if track else None, ptr(ws), ptr(mean), ptr(rstd), ptr(y), stream_ptr())
        ctx.save_for_backward(x, w, b_, mean, rstd)
        ctx.dims = (B, C, L)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, b_, mean, rstd = ctx.saved_tensors
        B, C, L = ctx.dims
        dy = _c(dy.float())
        chunk = lib().unopose_bn_train_chunk()
        ws = torch.empty(2 * B * C * ((L + chunk - 1) // chunk), dtype=torch.float32, device=x.device)
        dgamma, dbeta, dx = torch.empty_like(mean), torch.empty_like(mean), torch.empty_like(x)
        with torch.cuda.device(x.device):
            call("unopose_bn_relu_train_backward", ptr(x), ptr(dy), B, C, L, ptr(w), ptr(b_), ptr(mean), ptr(rstd), ptr(ws), ptr(dgamma), ptr(dbeta),
                 ptr(dx), stream_ptr())
        return dx, dgamma, dbeta, None


def _aligned16(x):
    """16-byte aligned storage once contiguous (the streaming kernels move float4 / float2 per lane)."""
    return (x.data_ptr() % 16 == 0) if x.is_contiguous() else True  # (_c() copies a non-contiguous view into a fresh, aligned buffer)


def bn_relu(x, bn):
    """F.relu(bn(x)) for an nn.BatchNorm2d: the fused training form (csrc/bn_train.hip) when `bn` is in train mode on fp32 CUDA data
    with affine parameters, else the modules themselves (eval statistics, CPU, other dtypes)."""
    L = x[0, 0].numel() if x.dim() >= 3 else 0
    if (USE_FUSED_BN_RELU and bn.training and type(bn) is torch.nn.BatchNorm2d and x.is_cuda and x.dtype == torch.float32 and x.dim() >= 3
            and L % 4 == 0 and L >= 4 and bn.weight is not None and bn.bias is not None and x.shape[0] <= 65535 and x.shape[1] <= 65535
            and _aligned16(x)):  # (exactly nn.BatchNorm2d: a SyncBatchNorm's statistics are not per rank; 16-byte loads)
        return _BNReLUTrain.apply(x, bn.weight, bn.bias, bn)
    return F.relu(bn(x))


class _BNReLUMaxPoolTrain(torch.autograd.Function):
    """max over the last axis of relu(batch_norm(x)) with BATCH statistics, x (B, C, N, S): the last SharedMLP layer and the pooling
    after it (Fi:167-174 under train()) as one op on csrc/bn_train.hip.  Saved for backward: x, mean, rstd, the arg max (B, C, N)."""

    @staticmethod
    def forward(ctx, x, weight, bias, bn):
        B, C, N, S = x.shape
        x = _c(x)
        chunk = lib().unopose_bn_train_chunk()
        ws = torch.empty(2 * B * C * ((N * S + chunk - 1) // chunk), dtype=torch.float32, device=x.device)
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        out = torch.empty(B, C, N, dtype=torch.float32, device=x.device)
        idx = torch.empty(B, C, N, dtype=torch.int32, device=x.device)
        track = bn.track_running_stats and bn.running_mean is not None
        if track:
            bn.num_batches_tracked.add_(1)
            momentum = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked.item())
        else:
            momentum = 0.0
        w, b_ = _c(weight.detach().float()), _c(bias.detach().float())
        with torch.cuda.device(x.device):
            call("unopose_bn_relu_maxpool_train_forward", ptr(x), B, C, N, S, ptr(w), ptr(b_), float(bn.eps), float(momentum),
                 ptr(bn.running_mean) if track else None, ptr(bn.running_var) if track else None, ptr(ws), ptr(mean), ptr(rstd), ptr(out), ptr(idx),
                 stream_ptr())
        ctx.save_for_backward(x, w, b_, mean, rstd, idx)
        ctx.mark_non_differentiable(idx)
        return out, idx

    @staticmethod
    def backward(ctx, g, _gidx):
        x, w, b_, mean, rstd, idx = ctx.saved_tensors
        B, C, N, S = x.shape
        g = _c(g.float())
        ws = torch.empty(2 * B * C, dtype=torch.float32, device=x.device)
        dgamma, dbeta, dx = torch.empty_like(mean), torch.empty_like(mean), torch.empty_like(x)
        with torch.cuda.device(x.device):
            call("unopose_bn_relu_maxpool_train_backward", ptr(x), ptr(g), ptr(idx), B, C, N, S, ptr(w), ptr(b_), ptr(mean), ptr(rstd), ptr(ws),
                 ptr(dgamma), ptr(dbeta), ptr(dx), stream_ptr())
        return dx, dgamma, dbeta, None


def bn_relu_maxpool(x, bn):
    """F.relu(bn(x)).max(dim=3)[0] for x (B, C, N, S): fused (csrc/bn_train.hip) when `bn` is in train mode on fp32 CUDA data with
    S in {32, 64, 128, 256}, else `bn_relu` followed by torch's max."""
    if (USE_FUSED_BN_RELU and bn.training and type(bn) is torch.nn.BatchNorm2d and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
            and x.shape[3] in (32, 64, 128, 256) and bn.weight is not None and bn.bias is not None and x.shape[0] <= 65535
            and x.shape[1] <= 65535 and _aligned16(x)):
        return _BNReLUMaxPoolTrain.apply(x, bn.weight, bn.bias, bn)[0]
    return bn_relu(x, bn).max(dim=3)[0]


TRAIN_FUSED_SALIENCY = True  # A/B attribute: False = the two softmax + matmul pairs of the reference through torch


class _SaliencyFn(torch.autograd.Function):
    """(softmax(inner, 2) @ s2, softmax(inner^T, 2) @ s1) for inner = atten[:, 1:, 1:] (C:68-76 / Fi:91-99 under train()) on
    csrc/saliency_train.hip: a row pass and a column pass forward (statistics + weighted sums, no softmax-sized tensor), a pass that writes
    the similarity's gradient and a column pass backward."""

    @staticmethod
    def forward(ctx, atten, s1, s2):
        B, n1, n2 = atten.shape[0], atten.shape[1] - 1, atten.shape[2] - 1
        a, v1, v2 = _c(atten.float()), _c(s1.float().reshape(B, n1)), _c(s2.float().reshape(B, n2))
        dev = a.device
        m1, rmax, rsum = (torch.empty(B, n1, dtype=torch.float32, device=dev) for _ in range(3))
        m2, cmax, csum = (torch.empty(B, n2, dtype=torch.float32, device=dev) for _ in range(3))
        with torch.cuda.device(dev):
            call("unopose_saliency_train_forward", ptr(a), ptr(v1), ptr(v2), B, n1, n2, ptr(m1), ptr(m2), ptr(rmax), ptr(rsum), ptr(cmax), ptr(csum),
                 stream_ptr())
        ctx.save_for_backward(a, v1, v2, m1, m2, rmax, rsum, cmax, csum)
        ctx.meta = (atten.dtype, s1.dtype, s2.dtype, s1.shape, s2.shape)
        return m1.reshape(s1.shape).to(s1.dtype), m2.reshape(s2.shape).to(s2.dtype)

    @staticmethod
    def backward(ctx, g1, g2):
        a, v1, v2, m1, m2, rmax, rsum, cmax, csum = ctx.saved_tensors
        B, n1, n2 = a.shape[0], a.shape[1] - 1, a.shape[2] - 1
        g1, g2 = _c(g1.float().reshape(B, n1)), _c(g2.float().reshape(B, n2))
        da = torch.empty_like(a)
        ds1, ds2 = torch.empty_like(v1), torch.empty_like(v2)
        with torch.cuda.device(a.device):
            call("unopose_saliency_train_backward", ptr(a), ptr(v1), ptr(v2), ptr(m1), ptr(m2), ptr(rmax), ptr(rsum), ptr(cmax), ptr(csum), ptr(g1),
                 ptr(g2), B, n1, n2, ptr(da), ptr(ds1), ptr(ds2), stream_ptr())
        ad, d1, d2, sh1, sh2 = ctx.meta
        return da.to(ad), ds1.reshape(sh1).to(d1), ds2.reshape(sh2).to(d2)


def saliency_pair(atten, s1, s2):
    """m1 = softmax(atten[:, 1:, 1:], dim=2) @ s2 and m2 = softmax(atten[:, 1:, 1:].transpose(1, 2), dim=2) @ s1 (s1 (B, n1, 1),
    s2 (B, n2, 1)): fused with its backward on fp32 CUDA data, the reference's expression otherwise."""
    if TRAIN_FUSED_SALIENCY and atten.is_cuda and atten.dtype == torch.float32 and s1.shape[-1] == 1 and s2.shape[-1] == 1 and atten.dim() == 3:
        return _SaliencyFn.apply(atten, s1, s2)
    inner = atten[:, 1:, 1:]
    return torch.matmul(F.softmax(inner, dim=2), s2), torch.matmul(F.softmax(inner.transpose(1, 2), dim=2), s1)


def nearest_partner(a, b, thr, over_b=True):
    """(min distance, arg min, any partner within thr) of every point of `a` (B, n, 3) over the points of `b` (B, m, 3) -- or, with
    over_b=False, of every point of b over a -- as the training labels need them (loss_utils.py:150-176), on csrc/glue.hip: one launch,
    no (B, n, m) matrix.  No gradient (labels)."""
    a, b = _c(a.detach().float()), _c(b.detach().float())
    B, n, m = a.shape[0], a.shape[1], b.shape[1]
    nout = n if over_b else m
    d = torch.empty(B, nout, dtype=torch.float32, device=a.device)
    idx = torch.empty(B, nout, dtype=torch.int32, device=a.device)
    anyc = torch.empty(B, nout, dtype=torch.uint8, device=a.device)
    with torch.cuda.device(a.device):
        call("unopose_nearest_partner", ptr(a), ptr(b), B, n, m, int(over_b), float(thr), ptr(d), ptr(idx), ptr(anyc), stream_ptr())
    return d, idx.long(), anyc.bool()


TRAIN_OWN_CONV = True  # A/B attribute: False = nn.Conv2d (MIOpen) for the PE's 1 x 1 convolutions under train()
_CONV_FWD_PAIRS = ((8, (32,)), (32, (32, 64)), (64, (32, 64, 128)), (128, (64, 128)))  # (cin up to, couts): csrc/conv_train.hip


_CONV_WGRAD_PAIRS = {32: 32, 64: 64, 128: 128}  # cout -> largest cin unopose_conv1x1_train_wgrad builds (csrc/conv_train.hip)


def _conv1x1_pair_ok(cin, cout):
    return any(cin <= k and cout in ms for k, ms in _CONV_FWD_PAIRS)


def _conv1x1_wgrad_ok(cin, cout):
    return cin <= _CONV_WGRAD_PAIRS.get(cout, 0)


class _Conv1x1Fn(torch.autograd.Function):
    """nn.Conv2d(cin, cout, 1, bias=False) on (B, C, N, S) fp32 (pytorch_utils.py:25-132 in train mode) on csrc/conv_train.hip:
    forward y = W x and input gradient dx = W^T dy are one kernel (v_mfma_f32_32x32x2_f32 along the contiguous positions, weights in
    LDS), the weight gradient dW = sum dy x^T stages both tensors through LDS (lanes along channels) with per-workgroup partials
    reduced in double.  No NCHW <-> NHWC transposes, no library call."""

    @staticmethod
    def forward(ctx, x, weight):
        B, C = x.shape[:2]
        L = x.numel() // (B * C)
        M = weight.shape[0]
        x = _c(x)
        w2 = _c(weight.detach().reshape(M, C).float())
        y = torch.empty((B, M) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            call("unopose_conv1x1_train_forward", ptr(x), B, C, L, ptr(w2), M, ptr(y), stream_ptr())
        ctx.save_for_backward(x, w2)
        ctx.wshape = tuple(weight.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w2 = ctx.saved_tensors
        M, C = w2.shape
        B = x.shape[0]
        L = x.numel() // (B * C)
        dy = _c(dy.float())
        dx = dw = None
        with torch.cuda.device(x.device):
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                call("unopose_conv1x1_train_forward", ptr(dy), B, M, L, ptr(w2.t().contiguous()), C, ptr(dx), stream_ptr())
            if ctx.needs_input_grad[1]:
                ws = torch.empty(lib().unopose_conv1x1_train_wgrad_blocks() * M * 128, dtype=torch.float32, device=x.device)
                dw = torch.empty(M, C, dtype=torch.float32, device=x.device)
                call("unopose_conv1x1_train_wgrad", ptr(dy), ptr(x), B, M, C, L, ptr(ws), ptr(dw), stream_ptr())
                dw = dw.reshape(ctx.wshape)
        return dx, dw


def conv1x1(x, conv):
    """`conv(x)` for a bias-free 1 x 1 nn.Conv2d: under autograd on fp32 CUDA data with a supported channel pair the own kernels
    (forward, and both gradients in backward), else the module itself."""
    w = conv.weight
    cout, cin = w.shape[0], w.shape[1]
    L = x[0, 0].numel() if x.dim() >= 3 else 0
    if (TRAIN_OWN_CONV and x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and conv.bias is None
            and w.shape[2:] == (1, 1) and tuple(conv.stride) == (1, 1) and tuple(conv.padding) == (0, 0) and tuple(conv.dilation) == (1, 1)
            and conv.groups == 1 and x.dim() == 4 and L % 64 == 0 and 64 <= L < (1 << 28) and _conv1x1_pair_ok(cin, cout)
            and (not x.requires_grad or _conv1x1_pair_ok(cout, cin)) and (not w.requires_grad or _conv1x1_wgrad_ok(cin, cout))
            and cout in (32, 64, 128) and _aligned16(x)):
        return _Conv1x1Fn.apply(x, w)
    return conv(x)


TRAIN_OWN_WGRAD = True  # A/B attribute: False = the linears' weight gradients through the library (dY^T @ X)
TRAIN_OWN_WGRAD_MIN_ROWS = 16384
TRAIN_OWN_GEMM = True  # A/B attribute: False = nn.Linear through the library
# The persistent 256 x 256-tile kernels pay off from a few tens of GFLOP per launch (measured at the training shapes: a
# 32 776 x 256 x 256 linear takes 38 us on csrc/gemm_f32.hip and 17 us on the library, the 4096 x 3072 x 4096 up-projection
# 0.31 vs 0.86 ms): below this many flops the training step keeps nn.Linear.
TRAIN_OWN_GEMM_MIN_FLOP = 4e9  # (round 6 same-box A/B of the training step: 2e10 121-124 ms, 1e10 119-121, 4e9 118.2, 1e9 117-120: scripts/ubench/train_ab.py)


def _transposed_weights(lin, kind):
    """W^T (K,N) in the operand form of the input-gradient GEMM dX = dY W, cached on the module per weight version:
    kind "f32": split layout of csrc/gemm_f32.hip, "bf16": bf16 rows."""
    key = (lin.weight._version, lin.weight.data_ptr(), kind)
    cache = lin.__dict__.setdefault("_wt_cache", {})
    c = cache.get(kind)
    if c is None or c[0] != key:
        with torch.no_grad():
            wt = lin.weight.detach().float().t().contiguous()
            c = (key, split_f32(wt) if kind == "f32" else wt.to(torch.bfloat16))
        cache[kind] = c
    return c[1]


class _LinearFn(torch.autograd.Function):
    """y = act(x W^T + b) for the trainable layers of the matcher, act in {none, ReLU}.  Forward: csrc/gemm_f32.hip (fp32: bf16 x 3
    matrix-core products, fp32 accumulation) or csrc/gemm.hip (autocast: bf16 operands) with bias / ReLU in the epilogue -- the
    kernels of the eval path.  Backward: dX = (dY * act') W on the same kernels against a cached W^T; dW = dY^T X and db = sum dY
    are reductions over the rows and go through torch (fp32 accumulate).  Gradients follow torch.nn.functional.linear's to
    the rounding of the products (tests/test_train_gpu.py)."""

    @staticmethod
    def forward(ctx, x, weight, bias, lin, relu, bf16):
        K = x.shape[-1]
        N = weight.shape[0]
        x2 = _c(x).reshape(-1, K)
        rows = x2.shape[0]
        if bf16:
            c = _bf16_weights(lin)
            xb = x2 if x2.dtype == torch.bfloat16 else x2.to(torch.bfloat16)
            y = linear_bf16_hip(xb, c[1], c[3], False, relu)
            ctx.save_for_backward(xb, y if relu else None)
        else:
            c = _f32x3_weights(lin)
            x2 = x2.float()
            y = linear_f32x3(split_f32(x2), c[1], c[2], rows, N, K, False, relu)
            ctx.save_for_backward(x2, y if relu else None)
        ctx.lin, ctx.relu, ctx.bf16, ctx.shape, ctx.has_bias, ctx.in_dtype = lin, relu, bf16, x.shape, bias is not None, x.dtype
        return y.reshape(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, gy):
        x2, y = ctx.saved_tensors
        lin = ctx.lin
        N, K = lin.weight.shape
        g = _c(gy).reshape(-1, N)
        if ctx.relu:
            g = g * (y > 0).to(g.dtype)
        rows = g.shape[0]
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            if ctx.bf16 and own_gemm_ok(rows, K, N):
                gb16 = g if g.dtype == torch.bfloat16 else g.to(torch.bfloat16)
                gx = linear_bf16_hip(_c(gb16), _transposed_weights(lin, "bf16"), _zero_bias(K, g.device), False, False)
            elif not ctx.bf16 and f32x3_ok(rows, K, N):
                gx = linear_f32x3(split_f32(_c(g.float())), _transposed_weights(lin, "f32"), _zero_bias(K, g.device), rows, K, N)
            else:
                gx = g.to(lin.weight.dtype) @ lin.weight.detach()
            gx = gx.reshape(ctx.shape).to(ctx.in_dtype)
        if ctx.needs_input_grad[1]:
            if (TRAIN_OWN_WGRAD and not ctx.bf16 and g.dtype == torch.float32 and x2.dtype == torch.float32 and g.is_cuda and N % 128 == 0
                    and K % 128 == 0 and rows >= TRAIN_OWN_WGRAD_MIN_ROWS and _aligned16(g) and _aligned16(x2)):
                # dW = dY^T X on csrc/conv_train.hip::linear_wgrad_f32_kernel (fp32 matrix instruction straight from the row-major operands)
                gc = _c(g)
                splits = lib().unopose_linear_wgrad_f32_splits(rows, N, K)
                ws = torch.empty(splits * N * K, dtype=torch.float32, device=g.device)
                gw = torch.empty(N, K, dtype=torch.float32, device=g.device)
                with torch.cuda.device(g.device):
                    call("unopose_linear_wgrad_f32", ptr(gc), ptr(x2), rows, N, K, ptr(ws), ptr(gw), stream_ptr())
                gw = gw.to(lin.weight.dtype)
            else:
                gw = (g.t() @ x2.to(g.dtype)).to(lin.weight.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g.float().sum(0).to(lin.bias.dtype)
        return gx, gw, gb, None, None, None


_ZERO_BIAS = {}


def _zero_bias(n, device):
    key = (n, device)
    if key not in _ZERO_BIAS:
        _ZERO_BIAS[key] = torch.zeros(n, device=device)
    return _ZERO_BIAS[key]


def linear_train(x, lin, relu=False):
    """The training-mode `linear`: own GEMM forward + input gradient under autograd when the shape fits, else nn.Linear."""
    N, K = lin.weight.shape
    rows = x.numel() // K
    bf16 = torch.is_autocast_enabled()
    ok = TRAIN_OWN_GEMM and x.is_cuda and rows > 0 and 2.0 * rows * N * K >= TRAIN_OWN_GEMM_MIN_FLOP and \
        (own_gemm_ok(rows, N, K) if bf16 else (x.dtype == torch.float32 and f32x3_ok(rows, N, K)))
    if not ok:
        y = lin(x)
        return F.relu(y) if relu else y
    return _LinearFn.apply(x, lin.weight, lin.bias, lin, relu, bf16)


def _lin(x, lin):
    """Inside the op-by-op composites: the trainable form in differentiable mode, the plain module call otherwise (the
    composites stay library-only A/B references of the fused kernels)."""
    return linear_train(x, lin) if (_DIFF and torch.is_grad_enabled()) else lin(x)


def _no_autograd():
    """The hand-written fp32 fast paths return tensors WITHOUT a grad_fn: they are taken only where autograd is not recording
    (no_grad / inference mode -- every eval entry point of this package).  With gradients enabled the call falls through to the
    torch composite (or to `linear_train` in differentiable mode), so eval-mode gradient use (pose refinement, saliency) stays
    correct instead of silently losing its graph."""
    return not torch.is_grad_enabled()


def _f32_path(x):
    return x.is_cuda and x.dtype == torch.float32 and not _DIFF and not torch.is_autocast_enabled() and _no_autograd()


def mlp(x, fc1, fc2):
    """timm Mlp (fc1 -> exact GELU -> fc2).  fp32: both linears on csrc/gemm_f32.hip, the hidden activation handed over in the
    split layout fc2 reads (never materialised in fp32); autocast: two fused bf16 GEMMs."""
    N1, K1 = fc1.weight.shape
    N2, K2 = fc2.weight.shape
    rows = x.numel() // K1
    if _f32_path(x) and f32x3_ok(rows, N1, K1) and f32x3_ok(rows, N2, K2):
        c1, c2 = _f32x3_weights(fc1), _f32x3_weights(fc2)
        hs = linear_f32x3(split_f32(_c(x).reshape(rows, K1)), c1[1], c1[2], rows, N1, K1, gelu=True, out="split")
        return linear_f32x3(hs, c2[1], c2[2], rows, N2, K2).reshape(*x.shape[:-1], N2)
    return linear(linear(x, fc1, gelu=True), fc2)


def linear(x, lin, relu=False, gelu=False):
    """nn.Linear under autocast without the per-call weight cast: bf16 copies of (weight, bias) are cached
    on the module (keyed by the parameter version) and the GEMM is issued directly in bf16.  Outside
    autocast this is just `lin(x)`.  Large problems (>= 4096 rows, N % 256 == 0, K % 64 == 0: every ViT linear and
    the up-projection) run on the hand-written GEMM of csrc/gemm.hip with the bias -- and, for `gelu=True`, timm
    Mlp's exact-erf GELU -- fused into its epilogue; the rest goes to hipBLASLt, where `relu=True` rides in the
    library epilogue (RELU_BIAS through torch._addmm_activation)."""
    if _f32_path(x):
        N, K = lin.weight.shape
        rows = x.numel() // K
        if f32x3_ok(rows, N, K):
            c = _f32x3_weights(lin)
            return linear_f32x3(split_f32(_c(x).reshape(rows, K), memo=True), c[1], c[2], rows, N, K, gelu, relu).reshape(*x.shape[:-1], N)
    if _DIFF and not gelu and torch.is_grad_enabled():
        return linear_train(x, lin, relu)
    if _DIFF or not (torch.is_autocast_enabled() and x.is_cuda):
        y = lin(x)
        return F.relu(y) if relu else (F.gelu(y) if gelu else y)
    cache = _bf16_weights(lin)
    with torch.autocast("cuda", enabled=False):
        xb = x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16)
        N, K = cache[1].shape
        rows = xb.numel() // K
        # Measured on the ViT shapes (scripts/gemm_ab.py, M = 87 936): the fused bias + GELU epilogue beats library
        # GEMM + separate GELU pass by 18 % (537 vs 655 us); on the plain linears the K loop of both is bound by the
        # same L2 -> LDS stream (~10 TB/s chip-wide) and the library's deeper pipeline is 0-25 % ahead, so those stay there.
        if own_gemm_ok(rows, N, K) and (HIP_GEMM_ALL or gelu):
            return linear_bf16_hip(_c(xb).reshape(rows, K), cache[1], cache[3], gelu, relu).reshape(*xb.shape[:-1], N)
        if FORBID_LIBRARY_BF16_GEMM:
            raise RuntimeError(f"ops.linear: a {rows} x {K} -> {N} bf16 linear does not fit csrc/gemm.hip (N % 256, K % 64) and would go to "
                               "a library GEMM while several forwards are in flight (PipelinedForward, depth > 1): library stream-K "
                               "kernels of two streams can starve each other.  Use depth=1 for this model configuration")
        note_fallback("linear", f"{rows} x {K} -> {N} bf16 (own GEMM: N % 256 == 0, K % 64 == 0): library GEMM")
        if relu and cache[2] is not None:
            x2 = xb.reshape(-1, xb.shape[-1])
            return torch._addmm_activation(cache[2], x2, cache[1].t()).reshape(*xb.shape[:-1], cache[1].shape[0])
        y = F.linear(xb, cache[1], cache[2])
        return F.relu(y) if relu else (F.gelu(y) if gelu else y)


# set by pipeline.PipelinedForward around every forward it enqueues with more than one in flight: the bf16 library fallback of
# `linear` raises instead of dispatching silently (ops.py:202)
FORBID_LIBRARY_BF16_GEMM = False
USE_FUSED_LINEAR_LN = True  # A/B attribute


def ffn_add_layernorm(x, expand, squeeze, norm):
    """LayerNorm(x + squeeze(relu(expand(x)))): the transformer layers' output block (transformer.py:151-193 `AttentionOutput`).
    fp32: the hidden activation goes from one fp32-class GEMM to the next in the split layout (never materialised in fp32, no split
    pass); autocast: expand + ReLU, then squeeze + residual + LayerNorm in one GEMM epilogue."""
    N1, K1 = expand.weight.shape
    N2, K2 = squeeze.weight.shape
    rows = x.numel() // K1
    if _f32_path(x) and f32x3_ok(rows, N1, K1) and f32x3_ok(rows, N2, K2):
        c1, c2 = _f32x3_weights(expand), _f32x3_weights(squeeze)
        hs = linear_f32x3(split_f32(_c(x).reshape(rows, K1)), c1[1], c1[2], rows, N1, K1, relu=True, out="split")
        y = linear_f32x3(hs, c2[1], c2[2], rows, N2, K2).reshape(*x.shape[:-1], N2)
        return add_layernorm(y, x, norm)
    return linear_add_layernorm(linear(x, expand, relu=True), squeeze, x, norm)


def linear_add_layernorm(h, lin, x, norm):
    """LayerNorm(lin(h) + x): the post-LN glue after an attention output projection / FFN squeeze (transformer.py:151-193).
    256-wide layers under autocast run as ONE launch -- residual add and LayerNorm in the epilogue of csrc/gemm.hip, on the
    fp32 accumulators (the unfused form rounds lin(h) to bf16 first); everything else: add_layernorm(linear(h), x)."""
    N, K = lin.weight.shape
    rows = h.numel() // K
    if (USE_FUSED_LINEAR_LN and not _DIFF and h.is_cuda and torch.is_autocast_enabled() and HIP_GEMM_ALL and N == 256
            and own_gemm_ok(rows, N, K) and tuple(norm.normalized_shape) == (256,) and norm.weight is not None and norm.bias is not None and lin.bias is not None):
        cache = _bf16_weights(lin)
        with torch.autocast("cuda", enabled=False):
            hb = _c(h if h.dtype == torch.bfloat16 else h.to(torch.bfloat16)).reshape(rows, K)
            xb = _c(x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16)).reshape(rows, N)
            out = torch.empty(rows, N, dtype=torch.bfloat16, device=h.device)
            with torch.cuda.device(h.device):
                call("unopose_linear_add_layernorm_bf16", ptr(hb), ptr(cache[1]), ptr(cache[3]), ptr(xb), ptr(norm.weight.detach()),
                     ptr(norm.bias.detach()), float(norm.eps), ptr(out), rows, K, stream_ptr())
        return out.reshape(*h.shape[:-1], N)
    return add_layernorm(linear(h, lin), x, norm)


def patch_embed(patches, conv):
    """The ViT's 14x14/14 patch convolution as a GEMM over unfolded patches (B,P,3*14*14) fp32 -> (B,P,D).  On the
    autocast path with `HIP_GEMM_ALL` it runs on csrc/gemm.hip like every other ViT linear: K = 588 is zero-padded to 640
    (a multiple of the 64-wide K tile) in the bf16 copies of patches and weight."""
    w = conv.weight.reshape(conv.weight.shape[0], -1)
    D, K = w.shape
    rows = patches.numel() // K
    if _f32_path(patches) and f32x3_ok(rows, D, 32):
        # fp32: K = 588 zero-padded to 608 (a multiple of the 32-wide stage of csrc/gemm_f32.hip)
        Kp = (K + 31) // 32 * 32
        key = _params_key(conv, Kp, "f32")
        cache = getattr(conv, "_f32x3_pad_cache", None)
        if cache is None or cache[0] != key:
            with torch.no_grad():
                wp = torch.zeros(D, Kp, dtype=torch.float32, device=w.device)
                wp[:, :K] = w.detach()
                b = torch.zeros(D, device=w.device) if conv.bias is None else conv.bias.detach().float().contiguous()
                cache = (key, split_f32(wp), b)
            conv._f32x3_pad_cache = cache
        a = torch.zeros(rows, Kp, dtype=torch.float32, device=patches.device)
        a[:, :K] = patches.reshape(rows, K)
        return linear_f32x3(split_f32(a), cache[1], cache[2], rows, D, Kp).reshape(*patches.shape[:-1], D)
    if _DIFF or not (HIP_GEMM_ALL and patches.is_cuda and torch.is_autocast_enabled() and own_gemm_ok(rows, D, 64)):
        return F.linear(patches, w, conv.bias)
    Kp = (K + 63) // 64 * 64
    key = _params_key(conv, Kp)
    cache = getattr(conv, "_bf16_pad_cache", None)
    if cache is None or cache[0] != key:
        with torch.no_grad():
            wp = torch.zeros(D, Kp, dtype=torch.bfloat16, device=w.device)
            wp[:, :K] = w.detach()
            b = torch.zeros(D, device=w.device) if conv.bias is None else conv.bias.detach().float().contiguous()
        cache = (key, wp, b)
        conv._bf16_pad_cache = cache
    with torch.autocast("cuda", enabled=False):
        a = torch.zeros(rows, Kp, dtype=torch.bfloat16, device=patches.device)
        a[:, :K] = patches.reshape(rows, K)
        return linear_bf16_hip(a, cache[1], cache[2]).reshape(*patches.shape[:-1], D)


def vit_prologue_ok(xa, vit):
    """The fused ViT prologue (csrc/glue.hip) runs under autocast on the hand-written GEMM, for ViT-B (768 wide)."""
    return (not _DIFF and xa.is_cuda and torch.is_autocast_enabled() and HIP_GEMM_ALL and USE_HIP_GEMM and xa.dtype == torch.float32
            and vit.pos_embed.shape[-1] == 768 and xa.shape[-1] == xa.shape[-2] and xa.shape[-1] % 14 == 0
            and vit.pos_embed.shape[1] == (xa.shape[-1] // 14) ** 2)


def vit_prologue(xa, xb, vit, norm1):
    """Both image batches (xb may be None) -> (x fp32 (n,T,768) residual stream, n1 = norm1(x) bf16): patch unfolding straight into
    the zero-padded bf16 patch matrix, the patch-embedding GEMM (csrc/gemm.hip), then pos_embed / class + register tokens / first
    LayerNorm in ONE pass.  Replaces cat([rgb, tem_rgb]) + unfold copy + zeros + cast copy + add + cat + LayerNorm."""
    conv = vit.patch_embed.proj
    w = conv.weight.reshape(conv.weight.shape[0], -1)
    D, K = w.shape
    Kp = (K + 63) // 64 * 64
    key = _params_key(conv, Kp, vit.cls_token._version, vit.reg_token._version)
    cache = getattr(conv, "_prologue_cache", None)
    if cache is None or cache[0] != key:
        with torch.no_grad():
            wp = torch.zeros(D, Kp, dtype=torch.bfloat16, device=w.device)
            wp[:, :K] = w.detach()
            b = torch.zeros(D, device=w.device) if conv.bias is None else conv.bias.detach().float().contiguous()
            prefix = torch.cat([vit.cls_token.detach().float().reshape(-1, D), vit.reg_token.detach().float().reshape(-1, D)], 0).contiguous()
        cache = (key, wp, b, prefix)
        conv._prologue_cache = cache
    _, wp, b, prefix = cache
    na, nb = xa.shape[0], 0 if xb is None else xb.shape[0]
    S = xa.shape[-1]
    P = (S // 14) ** 2
    npre = prefix.shape[0]
    dev = xa.device
    with torch.autocast("cuda", enabled=False), torch.cuda.device(dev):
        a = torch.empty((na + nb) * P, Kp, dtype=torch.bfloat16, device=dev)
        call("unopose_patchify_bf16", ptr(_c(xa)), na, None if xb is None else ptr(_c(xb)), nb, S, Kp, ptr(a), stream_ptr())
        y = linear_bf16_hip(a, wp, b)
        x = torch.empty(na + nb, npre + P, D, dtype=torch.float32, device=dev)
        n1 = torch.empty(na + nb, npre + P, D, dtype=torch.bfloat16, device=dev)
        call("unopose_vit_tokens_layernorm", ptr(y), ptr(vit.pos_embed.detach().float().contiguous()), ptr(prefix), npre, P, na + nb, D,
             ptr(norm1.weight.detach()), ptr(norm1.bias.detach()), float(norm1.eps), ptr(x), ptr(n1), stream_ptr())
    return x, n1


def bmm_nt_f32(a, b, alpha=1.0):
    """C[..., i, j] = alpha * sum_k a[..., i, k] b[..., j, k] on csrc/bmm_f32.hip (exact-fp32 MFMA, any strides): a (Bo,[Bi,]n,K),
    b (Bo,[Bi,]m,K) fp32 views -> contiguous (Bo,[Bi,]n,m) fp32."""
    assert a.dtype == torch.float32 and b.dtype == torch.float32 and a.dim() == b.dim() and a.dim() in (3, 4)
    if a.dim() == 3:
        a4, b4 = a.unsqueeze(1), b.unsqueeze(1)
    else:
        a4, b4 = a, b
    Bo, Bi, n, K = a4.shape
    m = b4.shape[2]
    out = torch.empty(Bo, Bi, n, m, dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        call("unopose_bmm_f32", ptr(a4), a4.stride(0), a4.stride(1), a4.stride(2), a4.stride(3), ptr(b4), b4.stride(0), b4.stride(1),
             b4.stride(2), b4.stride(3), ptr(out), Bo, Bi, n, m, K, float(alpha), stream_ptr())
    return out if a.dim() == 4 else out[:, 0]


def _own_f32(x):
    return x.is_cuda and not _DIFF and USE_F32X3 and _no_autograd()


def _own_glue(x):
    """the small fp32 steps between the kernels on own kernels (csrc/glue.hip, round 5): eval on the GPU, nothing recorded by autograd"""
    return x.is_cuda and not _DIFF and _no_autograd()


def cloud_radius(pts):
    """max_i |p_i - mean(p)| per cloud, pts (B,N,3) -> (B,) (the normalisation radius of the forward)"""
    if not _own_glue(pts):
        return torch.norm(pts - pts.mean(1, keepdim=True), dim=2).max(1)[0]
    p = _c(pts.float())
    out = torch.empty(p.shape[0], dtype=torch.float32, device=p.device)
    with torch.cuda.device(p.device):
        call("unopose_cloud_radius", ptr(p), p.shape[0], p.shape[1], ptr(out), stream_ptr())
    return out


def scale_by_radius(x, radius, multiply=False, eps=1e-6):
    """x / (radius[b] + eps) (or x * (...)) for x (B, ...) fp32"""
    if not (_own_glue(x) and x.dtype == torch.float32):
        s = (radius + eps).reshape(-1, *([1] * (x.dim() - 1)))
        return x * s if multiply else x / s
    xc = _c(x)
    out = torch.empty_like(xc)
    B = xc.shape[0]
    if xc.numel() == 0:
        return out
    r = radius.float().reshape(-1)
    if r.numel() != B:  # (the torch expression broadcast a single radius over the batch: the kernel indexes radius[b])
        if r.numel() != 1:
            raise ValueError(f"scale_by_radius: {r.numel()} radii for a batch of {B}")
        r = r.expand(B)
    with torch.cuda.device(x.device):
        call("unopose_scale_by_radius", ptr(xc), B, xc.numel() // B, ptr(_c(r)), float(eps), int(multiply), ptr(out), stream_ptr())
    return out


def overlap_scores(scores, n1):
    """clamp(sigmoid(.)) of the score heads' outputs (B, n_tot, 1) without the two background tokens -> (B, n_tot - 2) fp32"""
    if not (_own_glue(scores) and scores.dtype in (torch.float32, torch.bfloat16)):
        s1, s2 = scores[:, 1:(n1 + 1)], scores[:, (n1 + 2):]
        return torch.clamp(torch.sigmoid(torch.cat((s1, s2), dim=1).squeeze(-1).float()), 0, 1)
    sc = _c(scores)
    B, n_tot = sc.shape[0], sc.shape[1]
    out = torch.empty(B, n_tot - 2, dtype=torch.float32, device=sc.device)
    with torch.cuda.device(sc.device):
        call("unopose_overlap_scores", ptr(sc), int(sc.dtype == torch.bfloat16), B, n_tot, n1, ptr(out), stream_ptr())
    return out


def pose_score(dis, w, thr):
    """sum [dis < thr] w / (sum w + 1e-8) * mean w per batch row (model_utils.py:559-566)"""
    if not _own_glue(dis):
        return ((dis < thr).float() * w).sum(1) / (w.sum(1) + 1e-8) * w.mean(1)
    d, ww = _c(dis.float()), _c(w.float())
    out = torch.empty(d.shape[0], dtype=torch.float32, device=d.device)
    with torch.cuda.device(d.device):
        call("unopose_pose_score", ptr(d), ptr(ww), d.shape[0], d.shape[1], float(thr), ptr(out), stream_ptr())
    return out


def rigid_rows(p, t, R):
    """(p - t) @ R for row-vector points p (B,N,3), t (B,3), R (B,3,3) (Fi:69).  Under autocast the reference's `@` is a
    bf16 bmm (operands rounded to bf16, fp32 accumulation, bf16 result); here the same arithmetic as three broadcast
    multiply-adds, so that no library bf16 GEMM kernel is on the path (`own_gemm_ok`)."""
    if HIP_GEMM_ALL and torch.is_autocast_enabled() and _own_glue(p) and p.dtype == torch.float32 and p.dim() == 3:
        pc, out = _c(p), torch.empty(p.shape, dtype=torch.bfloat16, device=p.device)
        with torch.cuda.device(p.device):
            call("unopose_rigid_rows_bf16", ptr(pc), pc.shape[0], pc.shape[1], ptr(_c(t.float())), ptr(_c(R.float())), ptr(out), stream_ptr())
        return out
    x = p - t.unsqueeze(1)
    if _own_f32(p) and not torch.is_autocast_enabled() and x.dtype == torch.float32:
        return bmm_nt_f32(x, R.float().transpose(1, 2))  # (x @ R)[n, j] = sum_k x[n, k] R[k, j]
    if _DIFF or not (HIP_GEMM_ALL and p.is_cuda and torch.is_autocast_enabled()):
        return x @ R
    with torch.autocast("cuda", enabled=False):
        bf = torch.bfloat16
        xb, Rb = x.to(bf).float(), R.to(bf).float()
        y = xb[..., 0:1] * Rb[:, None, 0, :] + xb[..., 1:2] * Rb[:, None, 1, :] + xb[..., 2:3] * Rb[:, None, 2, :]
        return y.to(bf)


def score_head(x, lin):
    """The overlap-score head nn.Linear(d, 1) (C:66, Fi:89).  One output channel is no GEMM shape for csrc/gemm.hip, and no
    library bf16 GEMM may be on the autocast path (`own_gemm_ok`): evaluated as a multiply + row sum in fp32 on the
    bf16-rounded weights, rounded to the dtype the autocast Linear would return."""
    if _f32_path(x) and USE_F32X3 and x.shape[-1] == 256:  # fp32: the same row dot on the unrounded weights
        key = (lin.weight._version, lin.weight.data_ptr(), None if lin.bias is None else lin.bias._version, "f32")
        cache = getattr(lin, "_rowdot_cache_f32", None)
        if cache is None or cache[0] != key:
            cache = (key, lin.weight.detach().float().reshape(-1).contiguous(), 0.0 if lin.bias is None else float(lin.bias.detach().float().item()))
            lin._rowdot_cache_f32 = cache
        xc = _c(x)
        out = torch.empty(*x.shape[:-1], 1, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            call("unopose_row_dot", ptr(xc), 0, ptr(cache[1]), cache[2], xc.numel() // 256, 256, ptr(out), 0, stream_ptr())
        return out
    if _DIFF or not (HIP_GEMM_ALL and x.is_cuda and torch.is_autocast_enabled()):
        return lin(x)
    with torch.autocast("cuda", enabled=False):
        key = (lin.weight._version, lin.weight.data_ptr(), None if lin.bias is None else lin.bias._version)
        cache = getattr(lin, "_rowdot_cache", None)
        if cache is None or cache[0] != key:
            cache = (key, lin.weight.detach().to(torch.bfloat16).float().reshape(-1).contiguous(),
                     0.0 if lin.bias is None else float(lin.bias.detach().float().item()))
            lin._rowdot_cache = cache
        if x.shape[-1] == 256 and x.dtype in (torch.bfloat16, torch.float32):
            xc = _c(x)
            out = torch.empty(*x.shape[:-1], 1, dtype=torch.bfloat16, device=x.device)
            with torch.cuda.device(x.device):
                call("unopose_row_dot", ptr(xc), int(x.dtype == torch.bfloat16), ptr(cache[1]), cache[2], xc.numel() // 256, 256, ptr(out), 1,
                     stream_ptr())
            return out
        return ((x.float() * cache[1]).sum(-1, keepdim=True) + cache[2]).to(torch.bfloat16)


def gather_rows(feats, idx, off=0, alt=None, prepend=False):
    """out[b,j,:] = feats[b, idx[b,j] - off, :]  (the (B,N,C)-layout twin of gather_operation; avoids the two transpose copies
    around every reference call, model_utils.py:146-149, transformer.py:658).  `alt` (B,1,C) / (B,C): the row taken where
    idx - off < 0, and -- with `prepend` -- an extra row 0 of the output: the background-token sampling of the sparse-to-dense
    block in one launch (csrc/glue.hip; index cast + clamp + gather + compare + where + cat otherwise)."""
    B, N, C = feats.shape
    J = idx.shape[1]
    if (feats.is_cuda and not _DIFF and not feats.requires_grad and feats.is_contiguous() and idx.is_contiguous() and idx.dtype in (torch.int32, torch.int64)
            and (C * feats.element_size()) % 4 == 0 and B <= 65535):
        if alt is not None:
            alt = _c(alt.reshape(B, C).to(feats.dtype))
        out = torch.empty(B, J + int(prepend), C, dtype=feats.dtype, device=feats.device)
        with torch.cuda.device(feats.device):
            call("unopose_gather_rows", ptr(feats), B, N, C * feats.element_size(), ptr(idx), int(idx.dtype == torch.int64), J, int(off),
                 None if alt is None else ptr(alt), int(prepend), ptr(out), stream_ptr())
        return out
    if off == 0 and alt is None:
        return torch.gather(feats, 1, idx.long().unsqueeze(2).expand(-1, -1, C))
    i = idx.long() - off
    g = torch.gather(feats, 1, i.clamp(min=0).unsqueeze(2).expand(-1, -1, C))
    if alt is not None:
        a = alt.reshape(B, 1, C).to(g.dtype)
        g = torch.where((i < 0).unsqueeze(-1), a, g)
        if prepend:
            g = torch.cat([a, g], 1)
    return g


def vit_attention(qkv, heads):
    """timm Attention core: qkv (B,T,3C) -> (B,T,C).  bf16 (autocast) inputs run the flash-style HIP
    kernel (csrc/vit_attn.hip, head dim 64); fp32 inputs the op-by-op composite."""
    if _DIFF:
        return vit_attention_torch(qkv, heads)
    if qkv.dtype == torch.bfloat16 and qkv.shape[-1] == 3 * heads * 64:
        B, T, C3 = qkv.shape
        qkv = _c(qkv)
        out = torch.empty(B, T, C3 // 3, dtype=torch.bfloat16, device=qkv.device)
        with torch.cuda.device(qkv.device):
            call("unopose_vit_attention", ptr(qkv), B, T, heads, ptr(out), stream_ptr())
        return out
    if qkv.dtype == torch.float32 and qkv.is_cuda and qkv.shape[-1] == 3 * heads * 64:
        B, T, C3 = qkv.shape
        qkv = _c(qkv)
        out = torch.empty(B, T, C3 // 3, dtype=torch.float32, device=qkv.device)
        with torch.cuda.device(qkv.device):
            call("unopose_vit_attention_f32", ptr(qkv), B, T, heads, ptr(out), stream_ptr())
        return out
    note_fallback("vit_attention", f"head dim {qkv.shape[-1] // (3 * heads)} / dtype {qkv.dtype} (kernels: head dim 64, bf16 or fp32)")
    return vit_attention_torch(qkv, heads)


def vit_attention_f32_split(qkv, heads):
    """fp32 qkv (B,T,3C) -> the attention output in the split layout of csrc/gemm_f32.hip, a (B*T, 2C) bf16 tensor: the operand of
    the projection GEMM, written by the attention kernel itself (no fp32 round trip, no split pass)."""
    B, T, C3 = qkv.shape
    assert qkv.dtype == torch.float32 and qkv.is_cuda and C3 == 3 * heads * 64
    qkv = _c(qkv)
    out = torch.empty(B * T, 2 * (C3 // 3), dtype=torch.bfloat16, device=qkv.device)
    with torch.cuda.device(qkv.device):
        call("unopose_vit_attention_f32_split", ptr(qkv), B, T, heads, ptr(out), stream_ptr())
    return out


def vit_attention_f32_ss(qkv_split, B, T, heads):
    """qkv in the split layout ((B*T, 2 * 3C) bf16, as `linear_f32x3(..., out="split")` writes it) -> the attention output in the
    split layout ((B*T, 2C) bf16): csrc/vit_attn_f32s.hip, the fp32 ViT block's attention core (no fp32 tensor in between)."""
    C3 = qkv_split.shape[-1] // 2
    assert qkv_split.dtype == torch.bfloat16 and qkv_split.is_cuda and qkv_split.is_contiguous() and C3 == 3 * heads * 64
    out = torch.empty(B * T, 2 * (C3 // 3), dtype=torch.bfloat16, device=qkv_split.device)
    with torch.cuda.device(qkv_split.device):
        call("unopose_vit_attention_f32_ss", ptr(qkv_split), B, T, heads, ptr(out), stream_ptr())
    return out


def vit_attention_torch(qkv, heads):
    """Op-by-op composite of the same function."""
    B, T, C3 = qkv.shape
    C = C3 // 3
    hd = C // heads
    q, k, v = qkv.reshape(B, T, 3, heads, hd).permute(2, 0, 3, 1, 4)
    a = torch.softmax((q * hd ** -0.5) @ k.transpose(-2, -1), dim=-1, dtype=torch.float32).to(v.dtype) @ v
    return a.transpose(1, 2).reshape(B, T, C)


def bilinear_sample_native(z, choose, H, W, out=None, tok_offset=0):
    """Fused HIP version of bilinear_sample_pixels on the up-projection output in its NATIVE order
    z (B, side, side, 4, 4, 256) (no permute copy): (B,Np) int64 pixel indices -> (B,Np,256) fp32
    (written into `out` when given: a contiguous fp32 (B,Np,256) view).  With `tok_offset` > 0, z is
    (B, tok_offset + side*side, 4, 4, 256): the token tensor with its prefix (class / register) tokens left
    in place."""
    z = _c(z)
    assert z.shape[-1] == 256 and z.shape[-2] == 4 and z.shape[-3] == 4 and z.dtype in (torch.float32, torch.bfloat16)
    if out is not None:
        note_mutation()
    B = z.shape[0]
    if z.dim() == 5:  # (B, tokens, 4, 4, 256)
        tok_stride = z.shape[1]
        side = int(round((tok_stride - tok_offset) ** 0.5))
        assert side * side + tok_offset == tok_stride
    else:
        assert z.dim() == 6 and tok_offset == 0
        side = z.shape[1]
        tok_stride = side * side
    choose = _c(choose.long())
    Np = choose.shape[1]
    if out is None:
        out = torch.empty(B, Np, 256, dtype=torch.float32, device=z.device)
    assert out.shape == (B, Np, 256) and out.dtype == torch.float32 and out.is_contiguous()
    with torch.cuda.device(z.device):
        call("unopose_bilinear_sample_tokens", ptr(z), int(z.dtype == torch.bfloat16), ptr(choose), B, side, Np, int(H),
             int(W), int(tok_offset), int(tok_stride), ptr(out), stream_ptr())
    return out


USE_SPARSE_UPPROJ = True  # only the map cells the chosen pixels' bilinear taps read are up-projected (csrc/upproj.hip)


def sparse_upproj_ok(x):
    """True when `upproj_plan` + `sparse_pixel_features` may stand in for the dense up-projection + pixel sampling:
    HIP device, autocast (bf16 operands, as the dense autocast GEMM), inference."""
    return USE_SPARSE_UPPROJ and x.is_cuda and torch.is_autocast_enabled() and not _DIFF


def upproj_plan(choose, H, W, side, tok_offset, tok_stride):
    """choose (B2,Np) int64 pixel indices of (H,W) crops -> the gather plan of csrc/upproj.hip (device tensors only;
    needs nothing from the ViT, so it can be built on a side stream while the ViT runs)."""
    choose = _c(choose.long())
    B2, Np = choose.shape
    cells = 16 * side * side
    cap_rows = (B2 * min(4 * Np, cells) + 16 * 256 + 255) // 256 * 256
    dev = choose.device
    i32 = dict(dtype=torch.int32, device=dev)
    plan = dict(choose=choose, H=int(H), W=int(W), side=int(side), tok_offset=int(tok_offset), tok_stride=int(tok_stride),
                cap_rows=cap_rows, ws=torch.empty(B2 * (cells + 32), **i32), row_list=torch.empty(cap_rows, **i32),
                cellmap=torch.empty(B2 * cells, **i32), tile_info=torch.empty(18, **i32))
    with torch.cuda.device(dev):
        call("unopose_upproj_plan", ptr(choose), B2, Np, int(H), int(W), int(side), int(tok_offset), int(tok_stride), cap_rows,
             ptr(plan["ws"]), ptr(plan["row_list"]), ptr(plan["cellmap"]), ptr(plan["tile_info"]), stream_ptr())
    return plan


def sparse_pixel_features(acts, lin, plan, out=None):
    """ViT_AE's Linear 3072->4096 + pixel shuffle + bilinear upsampling + pixel gather (oneref_feature_extraction.py:
    200-236, model_utils.py:215-227) evaluated only where the chosen pixels look: acts (B2, tok_stride, K) bf16 token
    activations (prefix tokens in place), lin the up-projection -> (B2, Np, 256) fp32.  Same bf16 operands, fp32
    accumulation and bf16 rounding of the cell values as the dense path (`linear` + `bilinear_sample_native`)."""
    acts = _c(acts)
    if out is not None:
        note_mutation()
    B2, ts, K = acts.shape
    assert acts.dtype == torch.bfloat16 and ts == plan["tok_stride"] and B2 == plan["choose"].shape[0]
    cache = _bf16_weights(lin)
    w, bias = cache[1], cache[3]
    N = w.shape[0]
    assert N == 16 * 256 and w.shape[1] == K
    Np = plan["choose"].shape[1]
    dev = acts.device
    if out is None:
        out = torch.empty(B2, Np, 256, dtype=torch.float32, device=dev)
    assert out.shape == (B2, Np, 256) and out.dtype == torch.float32 and out.is_contiguous()
    with torch.cuda.device(dev):
        cells = torch.empty(plan["cap_rows"], 256, dtype=torch.bfloat16, device=dev)
        call("unopose_linear_bf16_gather", ptr(acts), B2 * ts, K, ptr(w), N, ptr(bias), ptr(plan["row_list"]),
             ptr(plan["tile_info"]), plan["cap_rows"] // 256, ptr(cells), stream_ptr())
        call("unopose_bilinear_sample_compact", ptr(cells), ptr(plan["cellmap"]), ptr(plan["choose"]), B2, plan["side"], Np,
             plan["H"], plan["W"], ptr(out), stream_ptr())
    return out


def bilinear_sample_pixels(low, choose, H, W):
    """F.interpolate(map, (H,W), bilinear, align_corners=False) followed by the pixel gather of
    get_chosen_pixel_feats (oneref_feature_extraction.py:229, model_utils.py:215-227), fused: only the
    chosen pixels are ever interpolated.  low (B,h,w,C) channels-last, choose (B,Np) int64 -> (B,Np,C)."""
    B, h, w, C = low.shape
    ys = torch.div(choose, W, rounding_mode="floor")
    xs = choose - ys * W

    def src(dst, n_in, n_out):
        s = (dst.float() + 0.5) * (n_in / n_out) - 0.5
        s = s.clamp(min=0.0)
        i0 = s.floor().long().clamp(max=n_in - 1)
        i1 = torch.where(i0 < n_in - 1, i0 + 1, i0)
        l1 = s - i0.float()
        return i0, i1, l1

    y0, y1, ly = src(ys, h, H)
    x0, x1, lx = src(xs, w, W)
    flat = low.reshape(B, h * w, C)

    def g(yy, xx):
        return torch.gather(flat, 1, (yy * w + xx).unsqueeze(2).expand(-1, -1, C)).float()

    lx, ly = lx.unsqueeze(2), ly.unsqueeze(2)
    top = (1 - lx) * g(y0, x0) + lx * g(y0, x1)
    bot = (1 - lx) * g(y1, x0) + lx * g(y1, x1)
    return (1 - ly) * top + ly * bot


def pairwise_distance(x, y):
    """model_utils.py:230-257."""
    xy = x @ y.transpose(-1, -2)
    return ((x ** 2).sum(-1).unsqueeze(-1) - 2 * xy + (y ** 2).sum(-1).unsqueeze(-2)).clamp(min=0.0)


def _bf16_split(w):
    """bf16 hi/lo parts of an fp32 tensor: w ~ hi + lo with ~2^-16 relative error."""
    hi = w.float().to(torch.bfloat16)
    lo = (w.float() - hi.float()).to(torch.bfloat16)
    return hi.contiguous(), lo.contiguous()


def _mfma_fragment_order(w):
    """(256 out, 256 in) -> [k/16][out/32][(k%16)/8][out%32][k%8], the B-operand order of
    v_mfma_f32_32x32x16_bf16 when wave w owns output channels [32w, 32w+32)."""
    return w.reshape(8, 32, 16, 2, 8).permute(2, 0, 3, 1, 4).contiguous()


GEO_TABLE = True     # bf16 (autocast) result through the table-interpolated kernel (4-point); False: the matrix-core kernel
GEO_TABLE_F32 = True  # fp32 result through the 6-point table kernel (fp32-class interpolation error); False: the split-operand matrix-core kernel
_GEO_HINV = 4        # table nodes per unit index (csrc/embed.hip: GT_HINV)
_GEO_D_RANGE = 64    # distance indices the table covers (the kernel evaluates larger ones from the defining sum)


_GEO_TABLE_UNAVAILABLE = {}  # device -> True once the table kernel's LDS opt-in has failed there


def _geo_tables(m, key, npoint=4):
    """proj_d(sinus(x)) and proj_a(sinus(x)) without their biases on the grid x = (r - (npoint / 2 - 1)) / 4 (fp64 sum, stored fp32),
    cached per weight version and interpolation order: what `unopose_geo_embedding_table` interpolates.  None when the angle table
    would not fit the kernel's LDS."""
    cache = m.__dict__.setdefault("_hip_tables", {})
    c = cache.get(npoint)
    if c is not None and c[0] == key:
        return c[1]
    import math
    lo = npoint // 2 - 1
    rows_a = int(math.floor(math.pi * float(m.factor_a) * _GEO_HINV)) + npoint + 1
    rows_d = _GEO_D_RANGE * _GEO_HINV + npoint
    tabs = None
    if rows_a <= 80 - npoint:
        div = m.embedding.div_term.detach().double()

        def table(rows, lin):
            x = (torch.arange(rows, device=div.device, dtype=torch.float64) - float(lo)) / _GEO_HINV
            om = x[:, None] * div[None, :]
            s = torch.stack([torch.sin(om), torch.cos(om)], dim=-1).reshape(rows, -1)
            return (s @ lin.weight.detach().double().t()).float().contiguous()

        tabs = (table(rows_d, m.proj_d), table(rows_a, m.proj_a), m.proj_d.weight.detach().float().contiguous())
    cache[npoint] = (key, tabs)
    return tabs


def geo_embedding(points, m, out_dtype=None):
    """GeometricStructureEmbedding.forward (transformer.py:303-350) as ONE fused HIP kernel
    (sinusoid generation -> MFMA -> max-over-k epilogue; csrc/embed.hip).  Under autocast(bf16) the
    result is bf16 with plain bf16 operands (what proj_d / proj_a produce under autocast in the
    reference); otherwise fp32 with hi/lo-split operands (fp32-class accuracy).  Round 5: both results come
    from the table-interpolated kernel (`unopose_geo_embedding_table`: fp32 Lagrange interpolation on tables of the
    two projections -- 4-point for the bf16 result, no bf16 operand rounding at all, 6-point (error ~1e-6) for the
    fp32 result; 3x / 5x faster than the contractions); `GEO_TABLE` / `GEO_TABLE_F32` switch back."""
    if _DIFF and torch.is_grad_enabled() and geo_embedding_train_ok(points, m):
        return _GeoEmbedFn.apply(_c(points.detach().float()), m.proj_d.weight, m.proj_d.bias, m.proj_a.weight, m.proj_a.bias, m)
    if _DIFF or m.proj_d.weight.shape != (256, 256) or m.angle_k != 3 or points.shape[1] < 4:
        if not _DIFF:
            note_fallback("geo_embedding", f"hidden_dim {tuple(m.proj_d.weight.shape)} / angle_k {m.angle_k} / {points.shape[1]} points (kernel: 256, 3, >= 4)")
        return geo_embedding_torch(points, m)  # other widths / k: op-by-op GPU composite
    points = _c(points.float())
    check_f32(points, "points")
    B, n, _ = points.shape
    if out_dtype is None:
        out_dtype = torch.bfloat16 if torch.is_autocast_enabled() else torch.float32
    bf16_out = out_dtype == torch.bfloat16
    cache = getattr(m, "_hip_cache", None)
    # (every tensor / number the derived state is computed from: an in-place edit of a bias alone, or a changed factor_a, rebuilds it)
    key = (m.proj_d.weight._version, m.proj_a.weight._version, m.proj_d.bias._version, m.proj_a.bias._version, m.proj_d.weight.data_ptr(),
           m.proj_d.weight.device, float(m.factor_a), float(m.sigma_d))
    if cache is None or cache[0] != key:
        assert m.proj_d.weight.shape == (256, 256) and m.angle_k == 3, "kernel is built for hidden_dim=256, k=3"
        wdh, wdl = (_mfma_fragment_order(t) for t in _bf16_split(m.proj_d.weight.detach()))
        wah, wal = (_mfma_fragment_order(t) for t in _bf16_split(m.proj_a.weight.detach()))
        bias = (m.proj_d.bias.detach().float() + m.proj_a.bias.detach().float()).contiguous()
        cache = (key, wdh, wdl, wah, wal, bias, m.embedding.div_term.detach().float().contiguous())
        m._hip_cache = cache
    _, wdh, wdl, wah, wal, bias, div = cache
    out = torch.empty(B, n, n, 256, dtype=out_dtype, device=points.device)
    knn = torch.empty(B, n, 3, dtype=torch.int32, device=points.device)
    if GEO_TABLE if bf16_out else GEO_TABLE_F32:
        npoint = 4 if bf16_out else 6
        tab = _geo_tables(m, key, npoint)
        if tab is not None and not _GEO_TABLE_UNAVAILABLE.get(points.device, False):
            try:
                with torch.cuda.device(points.device):
                    call("unopose_geo_embedding_table", ptr(points), B, n, ptr(tab[0]), tab[0].shape[0], ptr(tab[1]), tab[1].shape[0],
                         ptr(bias), ptr(tab[2]), ptr(div), _GEO_HINV, npoint, float(m.sigma_d), float(m.factor_a),
                         int(m.reduction_a == "mean"), int(bf16_out), ptr(knn), ptr(out), stream_ptr())
                return out
            except RuntimeError as e:
                if "cannot reserve" not in str(e):
                    raise
                # a device that cannot give the table kernel its ~145 KiB of LDS: the matrix-core kernel below takes over, loudly, for good
                _GEO_TABLE_UNAVAILABLE[points.device] = True
                note_fallback("geo_embedding", f"table kernel unavailable on {points.device} ({e}): matrix-core kernel")
    with torch.cuda.device(points.device):
        call("unopose_geo_embedding", ptr(points), B, n, ptr(wdh), ptr(wdl), ptr(wah), ptr(wal), ptr(bias), ptr(div),
             float(m.sigma_d), float(m.factor_a), int(m.reduction_a == "mean"), int(not bf16_out), int(bf16_out),
             ptr(knn), ptr(out), stream_ptr())
    return out


def geo_embedding_torch(points, m):
    """Op-by-op torch composite of the same function (A/B reference for tests and profiling)."""
    points = points.float()
    B, N, _ = points.shape
    with torch.autocast("cuda", enabled=False):
        dist = torch.sqrt(pairwise_distance(points, points))
        k = m.angle_k
        knn = dist.topk(k=k + 1, dim=2, largest=False)[1][:, :, 1:]
        knn_pts = torch.gather(points.unsqueeze(1).expand(B, N, N, 3), 2, knn.unsqueeze(3).expand(B, N, k, 3))
        ref = (knn_pts - points.unsqueeze(2)).unsqueeze(2).expand(B, N, N, k, 3)
        anc = (points.unsqueeze(1) - points.unsqueeze(2)).unsqueeze(3).expand(B, N, N, k, 3)
        sin_v = torch.linalg.norm(torch.cross(ref, anc, dim=-1), dim=-1)
        cos_v = (ref * anc).sum(-1)
        a_idx = torch.atan2(sin_v, cos_v) * m.factor_a
        d_idx = dist / m.sigma_d
        div = m.embedding.div_term

        def sinus(idx):
            om = idx.unsqueeze(-1) * div
            return torch.stack([torch.sin(om), torch.cos(om)], dim=-1).reshape(*idx.shape, -1)

        sd, sa = sinus(d_idx), sinus(a_idx)
    d_emb = _lin(sd, m.proj_d)
    a_emb = _lin(sa, m.proj_a)
    a_emb = a_emb.max(dim=3)[0] if m.reduction_a == "max" else a_emb.mean(dim=3)
    return d_emb + a_emb


_KEY_PAD = None


def token_attention(x, mem, att, heads, embed=None):
    """MultiHeadAttention / RPEMultiHeadAttention core (transformer.py:130-148, 386-405): returns the
    concatenated heads (B,n,C) before the output Linear.  The RPE term q.proj_p(E) is folded:
    q.(W_p e + b_p) = (q W_p).e + q.b_p  (SURVEY.md App-F), so no (B,4,n,m,64) tensor exists.
    Under autocast(bf16) the whole core (q k^T, folded RPE term, softmax, P v) is ONE HIP kernel on the
    bf16 matrix cores (csrc/attn.hip) that streams E once; in fp32 the op-by-op composite below runs."""
    global _KEY_PAD
    if not _DIFF and x.is_cuda and heads == 4 and x.shape[-1] == 256:
        if _KEY_PAD is None:
            from ._lib import lib
            _KEY_PAD = lib().unopose_token_attention_key_pad()
        if mem.shape[1] <= _KEY_PAD:
            if torch.is_autocast_enabled():
                return _token_attention_hip(x, mem, att, embed)
            if x.dtype == torch.float32:
                return _token_attention_hip_f32(x, mem, att, embed)
    if not _DIFF and x.is_cuda:
        note_fallback("token_attention", f"heads {heads} x width {x.shape[-1]}, {mem.shape[1]} keys, dtype {x.dtype} (kernels: 4 x 64, up to the padded key count)")
    return token_attention_torch(x, mem, att, heads, embed)


def _token_attention_hip_f32(x, mem, att, embed):
    """fp32 path: same kernel scheme with hi/lo-split bf16 MFMAs (csrc/attn_f32.hip); projections in
    fp32 with the RPE fold baked into the weights (exact algebra, fp32 rounding)."""

    B, n, C = x.shape
    m = mem.shape[1]
    rpe = embed is not None
    key = _params_key(att, rpe, "f32")
    cache = getattr(att, "_hip_cache_f32", None)
    if cache is None or cache[0] != key:
        with torch.no_grad():
            wq, bq = att.proj_q.weight.float(), att.proj_q.bias.float()
            pw, pb = [wq], [bq]
            if rpe:
                wp = att.proj_p.weight.float().reshape(4, 64, 256)
                pw.append(torch.einsum("hcd,hci->hdi", wp, wq.reshape(4, 64, 256)).reshape(1024, 256))
                pb.append(torch.einsum("hcd,hc->hd", wp, bq.reshape(4, 64)).reshape(1024))
            cache = (key, torch.cat(pw, 0).contiguous(), torch.cat(pb, 0).contiguous(),
                     torch.cat([att.proj_k.weight.float(), att.proj_v.weight.float()], 0).contiguous(),
                     torch.cat([att.proj_k.bias.float(), att.proj_v.bias.float()], 0).contiguous())
        att._hip_cache_f32 = cache
    _, w_q, b_q, w_kv, b_kv = cache
    yq = linear_f32_raw(x, w_q, b_q, att, "q")
    ykv = linear_f32_raw(mem, w_kv, b_kv, att, "kv")
    vt = torch.zeros(B, C, _KEY_PAD, dtype=torch.float32, device=x.device)
    vt[:, :, :m] = ykv[..., C:].transpose(1, 2)
    E = _c(embed.float()) if rpe else None
    out = torch.empty(B, n, C, dtype=torch.float32, device=x.device)
    qptr, kptr = yq.data_ptr(), ykv.data_ptr()
    with torch.cuda.device(x.device):
        call("unopose_token_attention_f32", ctypes.c_void_p(qptr), yq.stride(1), ctypes.c_void_p(kptr), ykv.stride(1),
             ptr(vt), ctypes.c_void_p(qptr + C * 4) if rpe else None, yq.stride(1), ptr(E) if rpe else None, B, n, m,
             0.125, ptr(out), stream_ptr())
    return out


def _attn_weights(att, rpe):
    """bf16 projection weights of one attention module, concatenated so that q | k | v (| the folded
    RPE query q W_p, 4 x 256) come out of as few GEMMs as possible.  Folding in fp32:
    (x Wq_h^T + bq_h) Wp_h = x (Wp_h^T Wq_h)^T + bq_h Wp_h."""
    key = _params_key(att, rpe)
    cache = getattr(att, "_hip_cache", None)
    if cache is not None and cache[0] == key:
        return cache[1]
    with torch.no_grad():
        bf = torch.bfloat16
        wq, bq = att.proj_q.weight.float(), att.proj_q.bias.float()
        parts_w, parts_b = [wq], [bq]
        if rpe:
            wp = att.proj_p.weight.float().reshape(4, 64, 256)  # [h][c][d]
            wqh = wq.reshape(4, 64, 256)                         # [h][c][in]
            parts_w.append(torch.einsum("hcd,hci->hdi", wp, wqh).reshape(1024, 256))
            parts_b.append(torch.einsum("hcd,hc->hd", wp, bq.reshape(4, 64)).reshape(1024))
        w_q = torch.cat(parts_w, 0).to(bf).contiguous()
        b_q = torch.cat(parts_b, 0).to(bf).contiguous()
        w_kv = torch.cat([att.proj_k.weight.float(), att.proj_v.weight.float()], 0).to(bf).contiguous()
        b_kv = torch.cat([att.proj_k.bias.float(), att.proj_v.bias.float()], 0).to(bf).contiguous()
        w_all = torch.cat([w_q, w_kv], 0).contiguous()
        b_all = torch.cat([b_q, b_kv], 0).contiguous()
    val = (w_q, b_q, w_kv, b_kv, w_all, b_all, b_q.float(), b_kv.float(), b_all.float())
    att._hip_cache = (key, val)
    return val


def _token_attention_hip(x, mem, att, embed):
    B, n, C = x.shape
    m = mem.shape[1]
    bf = torch.bfloat16
    rpe = embed is not None
    w_q, b_q, w_kv, b_kv, w_all, b_all, bq32, bkv32, ball32 = _attn_weights(att, rpe)
    nq = w_q.shape[0]
    xb = x.to(bf)
    with torch.autocast("cuda", enabled=False):
        if mem is x:  # self-attention: one GEMM for q | qp | k | v
            y = bf16_linear_2d(xb.reshape(B * n, C), w_all, ball32, b_all).reshape(B, n, -1)
            yq, ykv = y[..., :nq], y[..., nq:]
        else:
            yq = bf16_linear_2d(xb.reshape(B * n, C), w_q, bq32, b_q).reshape(B, n, -1)
            ykv = bf16_linear_2d(mem.to(bf).reshape(B * m, C), w_kv, bkv32, b_kv).reshape(B, m, -1)
    # q | qp and k | v are consumed in place from the projection outputs (row strides passed to the kernel)
    vt = torch.empty(B, C, _KEY_PAD, dtype=bf, device=x.device)
    with torch.cuda.device(x.device):
        call("unopose_transpose_pad_bf16", ctypes.c_void_p(ykv.data_ptr() + C * 2), ykv.stride(1), B, m, C, _KEY_PAD, ptr(vt), stream_ptr())
    E = _c(embed.to(bf)) if rpe else None
    out = torch.empty(B, n, C, dtype=bf, device=x.device)
    esz = 2
    q_ptr = yq.data_ptr()
    k_ptr = ykv.data_ptr()
    with torch.cuda.device(x.device):
        call("unopose_token_attention", ctypes.c_void_p(q_ptr), yq.stride(1), ctypes.c_void_p(k_ptr), ykv.stride(1),
             ptr(vt), ctypes.c_void_p(q_ptr + C * esz) if rpe else None, yq.stride(1),
             ptr(E) if E is not None else None, B, n, m, 0.125, ptr(out), stream_ptr())
    return out


def token_attention_torch(x, mem, att, heads, embed=None):
    """Op-by-op composite of the same function (fp32 path; A/B reference for the HIP kernel)."""
    B, n, C = x.shape
    hd = C // heads
    q = _lin(x, att.proj_q).reshape(B, n, heads, hd)
    k = _lin(mem, att.proj_k).reshape(B, -1, heads, hd)
    v = _lin(mem, att.proj_v).reshape(B, -1, heads, hd)
    s = torch.einsum("bnhc,bmhc->bhnm", q, k)
    if embed is not None:
        wp = att.proj_p.weight.reshape(heads, hd, C)  # rows of W_p grouped by head
        qp = torch.einsum("bnhc,hcd->bnhd", q, wp.to(q.dtype))  # (B,n,h,C)
        s = s + torch.einsum("bnhd,bnmd->bhnm", qp, embed.to(q.dtype))
        s = s + torch.einsum("bnhc,hc->bhn", q, att.proj_p.bias.reshape(heads, hd).to(q.dtype)).unsqueeze(-1)
    p = torch.softmax(s.float() / hd ** 0.5, dim=-1).to(v.dtype)
    return torch.einsum("bhnm,bmhc->bnhc", p, v).reshape(B, n, C)


def focused_linear_attention(xq, xkv, att, heads, focusing):
    """LinearAttention.forward (transformer.py:533-568).  With 4 heads x 64 the focusing + per-head
    contraction + z scaling run in ONE HIP kernel per side (csrc/linattn.hip): bf16 MFMAs under autocast,
    hi/lo-split (fp32-class) MFMAs on fp32 data; other shapes take the op-by-op composite."""
    if not _DIFF and heads == 4 and xq.shape[-1] == 256 and xq.is_cuda and float(focusing) == 3.0:
        if torch.is_autocast_enabled():
            return _focused_linear_attention_hip(xq, xkv, att, int(focusing))
        if xq.dtype == torch.float32 and xkv.dtype == torch.float32:
            return _focused_linear_attention_hip_f32(xq, xkv, att, int(focusing))
    if not _DIFF and xq.is_cuda:
        note_fallback("focused_linear_attention", f"heads {heads} x width {xq.shape[-1]}, focusing {focusing} (kernels: 4 x 64, 3)")
    return focused_linear_attention_torch(xq, xkv, att, heads, focusing)


def _focused_linear_attention_hip_f32(xq, xkv, att, focusing):
    """fp32 configuration: same two launches on fp32 data (csrc/linattn.hip linear_attn_f32_kernel: hi/lo-split
    MFMAs); projections are plain fp32 GEMMs, kv / k-sum tiny fp32 contractions."""
    B, N, C = xq.shape
    j = xkv.shape[1]
    key = (att.scale._version, att.scale.data_ptr(), "f32")
    cache = getattr(att, "_hip_cache_f32", None)
    if cache is None or cache[0] != key:
        with torch.no_grad():
            cache = (key, (1.0 / F.softplus(att.scale.float())).reshape(-1).contiguous())
        att._hip_cache_f32 = cache
    inv_sp = cache[1]
    q = _c(linear(xq, att.proj_q))
    kproj = _c(linear(xkv, att.proj_k))
    v = linear(xkv, att.proj_v)
    kf = torch.empty(B, j, C, dtype=torch.float32, device=xq.device)
    out = torch.empty(B, N, C, dtype=torch.float32, device=xq.device)
    with torch.cuda.device(xq.device):
        call("unopose_linear_attention_f32", ptr(kproj), ptr(inv_sp), None, None, B, j, focusing, 1, ptr(kf), stream_ptr())
        ksum = _c(kf.sum(dim=1))
        if _own_f32(v) and v.dtype == torch.float32:
            # kv_h^T[d][c] = sum_j v[j,h,d] k[j,h,c]: (pair, head) batches, both operands read in place (j strided)
            kvt = bmm_nt_f32(v.reshape(B, j, 4, 64).permute(0, 2, 3, 1), kf.reshape(B, j, 4, 64).permute(0, 2, 3, 1))
        else:
            kvt = _c(torch.einsum("bjhd,bjhc->bhdc", v.reshape(B, j, 4, 64), kf.reshape(B, j, 4, 64)))
        call("unopose_linear_attention_f32", ptr(q), ptr(inv_sp), ptr(kvt), ptr(ksum), B, N, focusing, 0, ptr(out),
             stream_ptr())
    return out


def _focused_linear_attention_hip(xq, xkv, att, focusing):
    bf = torch.bfloat16
    B, N, C = xq.shape
    j = xkv.shape[1]
    key = _params_key(att)
    cache = getattr(att, "_hip_cache", None)
    if cache is None or cache[0] != key:
        with torch.no_grad():
            w_kv = torch.cat([att.proj_k.weight.float(), att.proj_v.weight.float()], 0).to(bf).contiguous()
            b_kv = torch.cat([att.proj_k.bias.float(), att.proj_v.bias.float()], 0).to(bf).contiguous()
            inv_sp = (1.0 / F.softplus(att.scale.float())).reshape(-1).contiguous()
        cache = (key, w_kv, b_kv, inv_sp, b_kv.float())
        att._hip_cache = cache
    _, w_kv, b_kv, inv_sp, bkv32 = cache
    q = _c(linear(xq, att.proj_q))
    with torch.autocast("cuda", enabled=False):
        ykv = bf16_linear_2d(xkv.to(bf).reshape(B * j, C), w_kv, bkv32, b_kv).reshape(B, j, -1)
    kproj, v = _c(ykv[..., :C]), ykv[..., C:]
    kf = torch.empty(B, j, C, dtype=bf, device=xq.device)
    out = torch.empty(B, N, C, dtype=bf, device=xq.device)
    with torch.cuda.device(xq.device):
        call("unopose_linear_attention", ptr(kproj), ptr(inv_sp), None, None, B, j, focusing, 1, ptr(kf),
             stream_ptr())
        kf32 = kf.float()
        if _own_glue(kf):
            ksum = torch.empty(B, C, dtype=torch.float32, device=kf.device)  # (B,256): sum over the tokens
            call("unopose_token_sum_bf16", ptr(kf), B, j, C, ptr(ksum), stream_ptr())
        else:
            ksum = _c(kf32.sum(dim=1))
        # kv_h^T[d][c] = sum_j v[j,h,d] k[j,h,c]   (fp32 contraction: autocast would turn it into a library bf16 GEMM)
        with torch.autocast("cuda", enabled=False):
            if _own_f32(v):
                kvt = bmm_nt_f32(v.float().reshape(B, j, 4, 64).permute(0, 2, 3, 1), kf32.reshape(B, j, 4, 64).permute(0, 2, 3, 1)).to(bf)
            else:
                kvt = _c(torch.einsum("bjhd,bjhc->bhdc", v.float().reshape(B, j, 4, 64), kf32.reshape(B, j, 4, 64)).to(bf))
        call("unopose_linear_attention", ptr(q), ptr(inv_sp), ptr(kvt), ptr(ksum), B, N, focusing, 0, ptr(out),
             stream_ptr())
    return out


def focused_linear_attention_torch(xq, xkv, att, heads, focusing):
    """Op-by-op composite, kv branch (the shape test at transformer.py:560 is static)."""
    q, k, v = _lin(xq, att.proj_q), _lin(xkv, att.proj_k), _lin(xkv, att.proj_v)
    dt = v.dtype
    q, k = q.float(), k.float()
    scale = F.softplus(att.scale.float())
    q = (F.relu(q) + 1e-6) / scale
    k = (F.relu(k) + 1e-6) / scale
    qn, kn = q.norm(dim=-1, keepdim=True), k.norm(dim=-1, keepdim=True)
    q, k = q ** focusing, k ** focusing
    q = q / q.norm(dim=-1, keepdim=True) * qn
    k = k / k.norm(dim=-1, keepdim=True) * kn
    B, i, C = q.shape
    j = k.shape[1]
    hd = C // heads
    q = q.reshape(B, i, heads, hd)
    k = k.reshape(B, j, heads, hd)
    v = v.reshape(B, j, heads, hd)
    z = 1 / (torch.einsum("bihc,bhc->bih", q, k.sum(dim=1)) + 1e-6)
    if i * j * (hd + hd) > hd * hd * (i + j):
        kv = torch.einsum("bjhc,bjhd->bhcd", k.to(dt), v)
        x = torch.einsum("bihc,bhcd->bihd", q.to(dt), kv).float() * z.unsqueeze(-1)
    else:
        qk = torch.einsum("bihc,bjhc->bhij", q.to(dt), k.to(dt))
        x = torch.einsum("bhij,bjhd->bihd", qk, v).float() * z.unsqueeze(-1)
    return x.reshape(B, i, C).to(dt)


def pe_group_mlp_max(pts, radius, nsample, mlp, bf16x3=None, cand_in=None, want_cand=False, out_split=None):
    """QueryAndLRFGroup -> SharedMLP[6,32,64,128] -> max over neighbours (fine matcher PE, Fi:167-174)
    as ONE HIP kernel (csrc/pe.hip): neighbour lists, frames and all MLP activations stay on chip;
    (B,N,3) -> (B,N,128) fp32.  Matrix-core precision: exact fp32 MFMA by default; under autocast(bf16)
    (or bf16x3=True) bf16 MFMA with hi/lo-split operands (~2^-16 relative error, ~5x the fp32 MFMA rate).
    Neighbour-list hand-off (bf16x3 kernel only): `want_cand=True` also returns (lists (B,N,nsample) int32,
    counts (B,N) int32) of this pass; passing such a pair from a LARGER-radius pass over the same points as
    `cand_in` lets this pass test those candidates instead of scanning the cloud (same result).
    `out_split` = (buf (Btot,N,2W) bf16 viewed as the split layout of a W-wide fp32 row, first cloud b0, first channel c0):
    the 128 channels go straight into that operand of csrc/gemm_f32.hip (bf16x3 kernel only); returns buf."""
    if bf16x3 is None:  # the hi/lo-split matrix-core form is the fp32-class arithmetic of every other contraction of the fp32 path too
        bf16x3 = torch.is_autocast_enabled() or USE_F32X3
    if [tuple(l.conv.weight.shape[:2]) for l in mlp.layers()] != [(32, 6), (64, 32), (128, 64)] or nsample % 32:
        note_fallback("pe_group_mlp_max", f"MLP widths / nsample {nsample} (kernel: 6-32-64-128, nsample % 32 == 0)")
        return pe_group_mlp_max_unfused(pts, radius, nsample, mlp)  # other widths: grouping kernel + GEMMs
    pts = _c(pts.float())
    check_f32(pts, "pts")
    B, N, _ = pts.shape
    cache = getattr(mlp, "_hip_cache", None)
    key = _params_key(mlp)  # (conv weights, BatchNorm affine AND running statistics)
    if cache is None or cache[0] != key:
        with torch.no_grad():
            flat = []
            for l in mlp.layers():
                w, b = l.folded()
                flat += [w.float().contiguous(), b.float().contiguous()]
        assert [tuple(t.shape) for t in flat[::2]] == [(32, 6), (64, 32), (128, 64)], "kernel is built for [6,32,64,128]"
        from ._lib import lib
        image = torch.empty(lib().unopose_pe_image_bytes(), dtype=torch.uint8, device=pts.device)
        with torch.cuda.device(pts.device):
            call("unopose_pe_pack_weights", *(ptr(t) for t in flat), ptr(image), stream_ptr())
        cache = (key, flat, image)
        mlp._hip_cache = cache
    w1, b1, w2, b2, w3, b3 = cache[1]
    cand_out = None
    if out_split is not None:
        buf, b0, c0 = out_split
        assert bf16x3 and buf.dtype == torch.bfloat16 and buf.is_contiguous() and buf.shape[1] == N and c0 % 32 == 0 and b0 + B <= buf.shape[0]
        ld = buf.shape[2] // 2  # row width in 4-byte units
        with torch.cuda.device(pts.device):
            if want_cand:
                cand_out = (torch.empty(B, N, int(nsample), dtype=torch.int32, device=pts.device),
                            torch.empty(B, N, dtype=torch.int32, device=pts.device))
            ci = cand_in if cand_in is not None else (None, None)
            dst = ctypes.c_void_p(buf.data_ptr() + (b0 * N * ld + c0) * 4)
            call("unopose_pe_group_mlp_max_packed_out", ptr(pts), B, N, float(radius), int(nsample), ptr(cache[2]),
                 None if ci[0] is None else ptr(ci[0]), None if ci[0] is None else ptr(ci[1]),
                 0 if ci[0] is None else int(ci[0].shape[2]), None if cand_out is None else ptr(cand_out[0]),
                 None if cand_out is None else ptr(cand_out[1]), dst, ld, 1, stream_ptr())
        return (buf, cand_out) if want_cand else buf
    out = torch.empty(B, N, 128, dtype=torch.float32, device=pts.device)
    with torch.cuda.device(pts.device):
        if bf16x3:
            if want_cand:
                cand_out = (torch.empty(B, N, int(nsample), dtype=torch.int32, device=pts.device),
                            torch.empty(B, N, dtype=torch.int32, device=pts.device))
            ci = cand_in if cand_in is not None else (None, None)
            assert ci[0] is None or (ci[0].is_contiguous() and ci[0].shape[:2] == (B, N) and ci[1].shape == (B, N))
            call("unopose_pe_group_mlp_max_packed_cand", ptr(pts), B, N, float(radius), int(nsample), ptr(cache[2]),
                 None if ci[0] is None else ptr(ci[0]), None if ci[0] is None else ptr(ci[1]),
                 0 if ci[0] is None else int(ci[0].shape[2]), None if cand_out is None else ptr(cand_out[0]),
                 None if cand_out is None else ptr(cand_out[1]), ptr(out), stream_ptr())
        else:
            call("unopose_pe_group_mlp_max", ptr(pts), B, N, float(radius), int(nsample), ptr(w1), ptr(b1), ptr(w2),
                 ptr(b2), ptr(w3), ptr(b3), 0, ptr(out), stream_ptr())
    return (out, cand_out) if want_cand else out


def pe_group_mlp_max_unfused(pts, radius, nsample, mlp, chunk=4):
    """Same function as pe_group_mlp_max through the materialised (B,6,N,S) features (HIP fused
    ball-query+group+LRF, then torch GEMMs with BN folded): A/B reference for tests and profiling."""
    outs = []
    folded = [l.folded() for l in mlp.layers()]
    with torch.autocast("cuda", enabled=False):
        for b0 in range(0, pts.shape[0], chunk):
            x = query_lrf_group(pts[b0:b0 + chunk], radius, nsample)  # (b,6,N,S)
            x = x.permute(0, 2, 3, 1)
            for w, b in folded:
                x = F.relu(F.linear(x, w.float(), b.float()))
            outs.append(x.max(dim=2)[0])
    return torch.cat(outs, 0)


def furthest_point_sample(pts, npoint):
    return _ext.furthest_point_sampling(_c(pts.float()), npoint)


def feature_similarity(f1, f2, temp):
    """compute_feature_similarity, cosine + normalize (model_utils.py:260-282).
    Under autocast on the GPU the bf16 GEMM writes its fp32 accumulators straight out (`out_dtype`) with
    1/temp folded into the (small) left operand: one 4-byte write of the (B,N1,N2) matrix instead of a
    bf16 write, a division pass and the fp32 cast the pose heads ask for (1.6 GB per step at B=32)."""
    if _DIFF:  # the reference's expression, dtype and all (model_utils.py:260-282)
        return F.normalize(f1, p=2, dim=2) @ F.normalize(f2, p=2, dim=2).transpose(1, 2) / temp
    if f1.is_cuda and torch.is_autocast_enabled() and HIP_GEMM_ALL and _own_f32(f1):
        # no library bf16 GEMM on the path (own_gemm_ok): the bf16-rounded normalised operands, multiplied by the exact-fp32 MFMA
        # kernel (exact products, fp32 sums -- what the bf16 bmm with fp32 output computes up to summation order)
        with torch.autocast("cuda", enabled=False):
            return bmm_nt_f32(normalize_rows_bf16(f1, temp).float(), normalize_rows_bf16(f2, 1.0).float())
    a, b = F.normalize(f1.float(), p=2, dim=2), F.normalize(f2.float(), p=2, dim=2)
    if f1.is_cuda and torch.is_autocast_enabled() and not HIP_GEMM_ALL:
        with torch.autocast("cuda", enabled=False):
            return torch.bmm((a / temp).to(torch.bfloat16), b.to(torch.bfloat16).transpose(1, 2), out_dtype=torch.float32)
    if f1.is_cuda and torch.is_autocast_enabled():
        with torch.autocast("cuda", enabled=False):
            return torch.bmm((a / temp).to(torch.bfloat16).float(), b.to(torch.bfloat16).float().transpose(1, 2))
    if _own_f32(f1) and a.dtype == torch.float32:
        return bmm_nt_f32(_c(a), _c(b)) / temp
    return (a @ b.transpose(1, 2)) / temp


def soft_assignment(atten, score1, score2):
    """Mutual softmax x overlap scores + bg-aware labels (model_utils.py:434-446, 538-547)."""
    B = atten.shape[0]
    one = torch.ones(B, 1, device=atten.device)
    s1 = torch.cat((one, score1), 1)[:, :, None]
    s2 = torch.cat((one, score2), 1)[:, None, :]
    a = torch.softmax(atten, dim=2) * torch.softmax(atten, dim=1) * s1 * s2
    label1 = a[:, 1:, :].max(dim=2)[1]
    label2 = a[:, :, 1:].max(dim=1)[1]
    return a, label1, label2


def coarse_pose_torch(atten, score, pts1, pts2, rand, n1p=6000, n2p=300):
    """compute_coarse_Rt_overlap (model_utils.py:411-490); `rand` (B,3*n1p) is the uniform draw the
    reference makes inside forward (:462).  [torch composite + HIP 3-point Procrustes]"""
    B, N1, _ = pts1.shape
    N2 = pts2.shape[1]
    atten, pts1, pts2 = atten.float(), pts1.float(), pts2.float()
    a, l1, l2 = soft_assignment(atten, score[:, :N1].float(), score[:, N2:].float())
    w1, w2 = (l1 > 0).float(), (l2 > 0).float()
    ps = (a[:, 1:, 1:] * w1.unsqueeze(2) * w2.unsqueeze(1)).reshape(B, N1 * N2) ** 1.5
    # torch's CPU cumsum accumulates float32 input in double; mirror that so searchsorted agrees
    cs = torch.cumsum(ps.double(), dim=1).float()
    cs = cs / (cs[:, -1].unsqueeze(1) + 1e-8)
    idx = torch.searchsorted(cs, rand.contiguous())
    i1 = torch.clamp(idx.div(N2, rounding_mode="floor"), max=N1 - 1)
    i2 = torch.clamp(idx % N2, max=N2 - 1)
    p1 = torch.gather(pts1, 1, i1.unsqueeze(2).expand(-1, -1, 3)).reshape(B * n1p, 3, 3)
    p2 = torch.gather(pts2, 1, i2.unsqueeze(2).expand(-1, -1, 3)).reshape(B * n1p, 3, 3)
    rs, ts = weighted_procrustes(p2, p1, None, 0.5)
    rs, ts = rs.reshape(B, n1p, 3, 3), ts.reshape(B, n1p, 1, 3)
    p1, p2 = p1.reshape(B, n1p, 3, 3), p2.reshape(B, n1p, 3, 3)
    dis = torch.norm((p1 - ts) @ rs - p2, dim=3).mean(2)
    top = torch.topk(dis, n2p, dim=1, largest=False)[1]
    rs2 = torch.gather(rs, 1, top.reshape(B, n2p, 1, 1).expand(-1, -1, 3, 3))
    ts2 = torch.gather(ts, 1, top.reshape(B, n2p, 1, 1).expand(-1, -1, 1, 3))
    tp = (pts1.unsqueeze(1) - ts2) @ rs2  # (B,n2p,N1,3)
    d = torch.sqrt(pairwise_distance(tp, pts2.unsqueeze(1))).min(3)[0]  # (B,n2p,N1)
    sc = w1.unsqueeze(1).sum(2) / ((d * w1.unsqueeze(1)).sum(2) + 1e-8)
    pose_score, best = sc.max(1)
    R = torch.gather(rs2, 1, best.reshape(B, 1, 1, 1).expand(-1, -1, 3, 3)).squeeze(1)
    t = torch.gather(ts2, 1, best.reshape(B, 1, 1, 1).expand(-1, -1, 1, 3)).squeeze(2).squeeze(1)
    return R, t, pose_score


def fine_pose_torch(atten, score, pts1, pts2, dis_thres=0.15):
    """compute_fine_Rt_overlap (model_utils.py:527-566).  [torch composite + HIP weighted Procrustes]"""
    atten, pts1, pts2 = atten.float(), pts1.float(), pts2.float()
    N1 = pts1.shape[1]
    a, l1, l2 = soft_assignment(atten, score[:, :N1].float(), score[:, N1:].float())
    a = a[:, 1:, 1:] * (l1 > 0).float().unsqueeze(2) * (l2 > 0).float().unsqueeze(1)
    rows = a.sum(2)
    pred = (a / (rows.unsqueeze(2) + 1e-6)) @ pts2
    R, t = weighted_procrustes(pred, pts1, rows, 0.001)
    pp = (pts1 - t.unsqueeze(1)) @ R
    dis = torch.sqrt(pairwise_distance(pp, pts2)).min(2)[0]
    mask = (l1 > 0).float()
    ps = ((dis < dis_thres).float() * mask).sum(1) / (mask.sum(1) + 1e-8)
    return R, t, ps * mask.mean(1)


def _assign_labels(atten, score1, score2):
    B, R, C = atten.shape
    dev = atten.device
    stats = torch.empty(2 * B * (R + C), dtype=torch.float32, device=dev)
    w1 = torch.empty(B, R - 1, dtype=torch.float32, device=dev)
    w2 = torch.empty(B, C - 1, dtype=torch.float32, device=dev)
    call("unopose_assign_labels", ptr(atten), B, R, C, ptr(score1), ptr(score2), ptr(stats), ptr(w1), ptr(w2),
         stream_ptr())
    return stats, w1, w2


def coarse_pose(atten, score, pts1, pts2, rand, n1p=6000, n2p=300):
    """compute_coarse_Rt_overlap (model_utils.py:411-490) on HIP kernels (csrc/posehead.hip): streaming
    assignment statistics, CDF + searchsorted + 3-point Procrustes + residual per hypothesis, candidate
    scoring; torch only picks the top-k / argmax of tiny (B,6000) / (B,300) arrays."""
    B, N1, _ = pts1.shape
    N2 = pts2.shape[1]
    atten, pts1, pts2 = _c(atten.float()), _c(pts1.float()), _c(pts2.float())
    check_f32(atten, "atten")
    score1, score2 = _c(score[:, :N1].float()), _c(score[:, N2:].float())  # NB `N2:` (model_utils.py:440)
    rand = _c(rand.float())
    dev = atten.device
    with torch.cuda.device(dev):
        stats, w1, w2 = _assign_labels(atten, score1, score2)
        cdf = torch.empty(B, N1 * N2, dtype=torch.float32, device=dev)
        rs = torch.empty(B, n1p, 3, 3, dtype=torch.float32, device=dev)
        ts = torch.empty(B, n1p, 3, dtype=torch.float32, device=dev)
        dis = torch.empty(B, n1p, dtype=torch.float32, device=dev)
        call("unopose_coarse_hypotheses", ptr(atten), B, N1 + 1, N2 + 1, ptr(score1), ptr(score2), ptr(stats),
             ptr(w1), ptr(w2), ptr(rand), n1p, ptr(pts1), ptr(pts2), ptr(cdf), ptr(rs), ptr(ts), ptr(dis),
             stream_ptr())
        top = torch.topk(dis, n2p, dim=1, largest=False)[1].contiguous()
        sc = torch.empty(B, n2p, dtype=torch.float32, device=dev)
        call("unopose_coarse_scores", ptr(pts1), ptr(pts2), B, N1, N2, ptr(rs), ptr(ts), n1p, ptr(top), n2p, ptr(w1),
             ptr(sc), stream_ptr())
    pose_score, best = sc.max(1)
    hyp = torch.gather(top, 1, best.unsqueeze(1))  # (B,1)
    R = torch.gather(rs, 1, hyp.reshape(B, 1, 1, 1).expand(-1, -1, 3, 3)).squeeze(1)
    t = torch.gather(ts, 1, hyp.reshape(B, 1, 1).expand(-1, -1, 3)).squeeze(1)
    return R, t, pose_score


def fine_pose(atten, score, pts1, pts2, dis_thres=0.15):
    """compute_fine_Rt_overlap (model_utils.py:527-566) on HIP kernels: five streaming passes over the
    (B,N1+1,N2+1) similarity that write only O(N) statistics, weighted Procrustes (Jacobi), min-distance
    verification."""
    B, N1, _ = pts1.shape
    N2 = pts2.shape[1]
    atten, pts1, pts2 = _c(atten.float()), _c(pts1.float()), _c(pts2.float())
    check_f32(atten, "atten")
    score1, score2 = _c(score[:, :N1].float()), _c(score[:, N1:].float())
    dev = atten.device
    with torch.cuda.device(dev):
        stats, w1, w2 = _assign_labels(atten, score1, score2)
        weight = torch.empty(B, N1, dtype=torch.float32, device=dev)
        pred = torch.empty(B, N1, 3, dtype=torch.float32, device=dev)
        call("unopose_fine_correspondences", ptr(atten), B, N1 + 1, N2 + 1, ptr(score1), ptr(score2), ptr(stats),
             ptr(w1), ptr(w2), ptr(pts2), ptr(weight), ptr(pred), stream_ptr())
        R, t = weighted_procrustes(pred, pts1, weight, 0.001)
        dis = torch.empty(B, N1, dtype=torch.float32, device=dev)
        call("unopose_min_dist", ptr(pts1), ptr(pts2), B, N1, N2, ptr(R), ptr(t), 1, ptr(dis), stream_ptr())
    return R, t, pose_score(dis, w1, dis_thres)


USE_FUSED_FINE = True  # bf16 fine stage without the (B,N1+1,N2+1) similarity (csrc/fineassign.hip)


def fine_pose_fused_ok(f1, f2):
    """True when `fine_pose_from_features` may stand in for feature_similarity + fine_pose: HIP device, autocast (the
    bf16 product the reference's autocast matmul makes), inference, 256-wide features."""
    return (USE_FUSED_FINE and f1.is_cuda and torch.is_autocast_enabled() and not _DIFF and f1.shape[-1] == 256
            and f2.shape[-1] == 256)


def normalize_rows_bf16(f, temp):
    """bf16(F.normalize(f.float(), dim=-1) / temp) in one pass (csrc/glue.hip); f (...,256) bf16 or fp32."""
    if f.shape[-1] != 256 or f.dtype not in (torch.bfloat16, torch.float32):
        return _c((F.normalize(f.float(), p=2, dim=-1) / temp).to(torch.bfloat16))
    fc = _c(f)
    out = torch.empty(f.shape, dtype=torch.bfloat16, device=f.device)
    with torch.cuda.device(f.device):
        call("unopose_normalize_rows_bf16", ptr(fc), int(f.dtype == torch.bfloat16), fc.numel() // 256, 256, float(temp), ptr(out), stream_ptr())
    return out


def fine_pose_from_features(f1, f2, temp, score, pts1, pts2, dis_thres=0.15):
    """compute_feature_similarity (cosine, /temp; model_utils.py:260-282) + compute_fine_Rt_overlap (:527-566) with the
    similarity recomputed tile by tile inside the three reduction passes instead of stored: f1 (B,N1+1,256), f2
    (B,N2+1,256) are the out_proj features (row 0 = background token).  Same bf16 operands / fp32 accumulation as the
    autocast bmm of `feature_similarity`."""
    B, N1, _ = pts1.shape
    N2 = pts2.shape[1]
    assert f1.shape == (B, N1 + 1, 256) and f2.shape == (B, N2 + 1, 256)
    a, b = normalize_rows_bf16(f1, temp), normalize_rows_bf16(f2, 1.0)
    pts1, pts2 = _c(pts1.float()), _c(pts2.float())
    score1, score2 = _c(score[:, :N1].float()), _c(score[:, N1:].float())
    dev = pts1.device
    R, C = N1 + 1, N2 + 1
    with torch.cuda.device(dev):
        ws = torch.empty(B * (R + C) + B * (-(-N1 // 256) - (-N2 // 256)), dtype=torch.float32, device=dev)
        w1 = torch.empty(B, N1, dtype=torch.float32, device=dev)
        w2 = torch.empty(B, N2, dtype=torch.float32, device=dev)
        weight = torch.empty(B, N1, dtype=torch.float32, device=dev)
        pred = torch.empty(B, N1, 3, dtype=torch.float32, device=dev)
        call("unopose_fine_assign", ptr(a), ptr(b), B, R, C, 256, 1.0 / temp, ptr(score1), ptr(score2), ptr(pts2), ptr(ws),
             ptr(w1), ptr(w2), ptr(weight), ptr(pred), stream_ptr())
        Rm, t = weighted_procrustes(pred, pts1, weight, 0.001)
        dis = torch.empty(B, N1, dtype=torch.float32, device=dev)
        call("unopose_min_dist", ptr(pts1), ptr(pts2), B, N1, N2, ptr(Rm), ptr(t), 1, ptr(dis), stream_ptr())
    return Rm, t, pose_score(dis, w1, dis_thres)


def add_layernorm(a, b, norm, out_dtype=None, out=None):
    """LayerNorm(a + b) in one HIP pass (b may be None); a/b fp32 or bf16, output `out_dtype`
    (default: bf16 under autocast, else a.dtype).  norm: nn.LayerNorm.  `out`: optional destination, a view
    of shape a.shape whose rows are `ld` elements apart in one row-major buffer (last dim contiguous) --
    several LayerNorms can then fill column blocks of one wider tensor without a concatenation."""
    a = _c(a)
    C = a.shape[-1]
    rows = a.numel() // C
    if b is not None:
        b = _c(b)
        assert b.shape == a.shape
    if out is not None:
        note_mutation()
    if out is None:
        if out_dtype is None:
            out_dtype = torch.bfloat16 if torch.is_autocast_enabled() else a.dtype
        out = torch.empty(a.shape, dtype=out_dtype, device=a.device)
        ld = C
    else:
        out_dtype = out.dtype
        ld = out.stride(-2)
        assert out.shape == a.shape and out.stride(-1) == 1 and ld >= C
        exp = ld
        for d in range(out.dim() - 2, -1, -1):  # every leading dim must continue the same row pitch
            assert out.stride(d) == exp or out.shape[d] == 1, "out rows must be uniformly strided"
            exp *= out.shape[d]
    ok = (torch.float32, torch.bfloat16)
    assert a.dtype in ok and out_dtype in ok and (b is None or b.dtype in ok) and a.is_cuda
    with torch.cuda.device(a.device):
        call("unopose_add_layernorm_strided", ptr(a), int(a.dtype == torch.bfloat16), ptr(b) if b is not None else None,
             int(b is not None and b.dtype == torch.bfloat16), ptr(norm.weight), ptr(norm.bias), rows, C,
             float(norm.eps), ptr(out), int(out_dtype == torch.bfloat16), int(ld), stream_ptr())
    return out


def scale_residual_(x, y, gamma):
    """x (fp32, contiguous) += gamma * y (bf16) in place (ViT LayerScale residual)."""
    note_mutation()
    assert x.dtype == torch.float32 and x.is_contiguous() and y.dtype == torch.bfloat16
    y = _c(y)
    C = x.shape[-1]
    with torch.cuda.device(x.device):
        call("unopose_scale_residual", ptr(x), ptr(y), ptr(gamma), x.numel() // C, C, stream_ptr())
    return x


def scale_residual_layernorm_f32_(x, y, gamma, norm):
    """fp32 twin for the no-autocast path: x (fp32, contiguous) += gamma * y (fp32) in place (y None: no update); returns
    LayerNorm(x) in the split layout of csrc/gemm_f32.hip as a (rows, 2C) bf16 tensor (norm None: residual update only, returns x)."""
    note_mutation()
    assert x.dtype == torch.float32 and x.is_contiguous() and (y is None or y.dtype == torch.float32)
    C = x.shape[-1]
    rows = x.numel() // C
    out = None if norm is None else torch.empty(rows, 2 * C, dtype=torch.bfloat16, device=x.device)
    with torch.cuda.device(x.device):
        call("unopose_scale_residual_layernorm_f32", ptr(x), None if y is None else ptr(_c(y)), None if y is None else ptr(gamma),
             None if norm is None else ptr(norm.weight), None if norm is None else ptr(norm.bias), rows, C,
             0.0 if norm is None else float(norm.eps), None if out is None else ptr(out), stream_ptr())
    return x if norm is None else out


def vit_f32_fused_ok(x, vit):
    """The no-autocast ViT on the fused fp32-class path (split-layout LayerNorm outputs feeding csrc/gemm_f32.hip directly)?"""
    C = x.shape[-1]
    return (x.is_cuda and x.dtype == torch.float32 and not _DIFF and _no_autograd() and not torch.is_autocast_enabled() and USE_F32X3 and C % 32 == 0
            and C <= 1024 and C % 64 == 0 and f32x3_ok(x.numel() // C, 3 * C, C) and f32x3_ok(x.numel() // C, C, 4 * C))


def scale_residual_layernorm_(x, y, gamma, norm):
    """x (fp32) += gamma * y (bf16) in place; returns LayerNorm(x) in bf16 -- one pass over the residual stream."""
    note_mutation()
    assert x.dtype == torch.float32 and x.is_contiguous() and y.dtype == torch.bfloat16
    y = _c(y)
    C = x.shape[-1]
    out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    with torch.cuda.device(x.device):
        call("unopose_scale_residual_layernorm", ptr(x), ptr(y), ptr(gamma), ptr(norm.weight), ptr(norm.bias),
             x.numel() // C, C, float(norm.eps), ptr(out), stream_ptr())
    return out


# ---- round 6: the ViT's residual + LayerNorm passes folded into the GEMMs around them (csrc/gemm_kernel.h EPI 5 / 6 / 7) -------------
# timm Block under autocast (oneref_feature_extraction.py:24-42):  x = x + ls1(attn(norm1(x)));  x = x + ls2(mlp(norm2(x))).
# Producer (proj / fc2, LayerScale folded into its weights): the epilogue updates the fp32 residual stream in place and emits the updated
# rows in bf16 + per-row partial sums.  Consumer (qkv / fc1 on W' = norm.weight (.) W): reads the UN-normalised bf16 rows and applies
# LayerNorm algebraically in its epilogue.  No separate residual / LayerNorm pass: 23 launches and 270 MB of the 1080 MB per (GEMM, LN,
# GEMM) link are gone (same-box A/B: scripts/ubench/lnfold_ab.py, profiles/r06_ln_fold_ab.txt).
USE_LN_FOLD = True  # A/B attribute: False = scale_residual_layernorm_ between the GEMMs (round 5's path)


def ln_fold_ok(rows, C):
    """The fold runs on the 256 x 256-tile kernel only: shapes whose proj / fc2 grid the small-tile kernel would take keep the separate pass."""
    if not (USE_LN_FOLD and USE_HIP_GEMM and HIP_GEMM_ALL) or C % 256 != 0 or C // 256 > 4 or rows * C * 4 >= 2 ** 31:
        return False
    n_cu = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count & ~7
    return ((rows + 255) // 256) * (C // 256) >= n_cu * 5 // 8


def _fold_producer_weights(lin, gamma):
    """(gamma (.) W) bf16, gamma (.) b fp32 of a LayerScale-d linear, cached on the module (keyed on every tensor they are derived from)."""
    key = (lin.weight._version, lin.weight.data_ptr(), lin.weight.device, None if lin.bias is None else lin.bias._version, gamma._version, gamma.data_ptr())
    cache = getattr(lin, "_fold_prod_cache", None)
    if cache is None or cache[0] != key:
        with torch.no_grad():
            g = gamma.detach().float()
            w = (lin.weight.detach().float() * g[:, None]).to(torch.bfloat16).contiguous()
            b = (torch.zeros_like(g) if lin.bias is None else lin.bias.detach().float() * g).contiguous()
        cache = (key, w, b)
        lin._fold_prod_cache = cache
    return cache


def _fold_consumer_weights(lin, norm):
    """W' = W (.) norm.weight (bf16), c_n = sum_k W'[n][k] (of the ROUNDED W': what the matrix cores multiply), d_n = sum_k norm.bias[k] W[n][k] + b[n]."""
    key = (lin.weight._version, lin.weight.data_ptr(), lin.weight.device, None if lin.bias is None else lin.bias._version,
           norm.weight._version, norm.weight.data_ptr(), norm.bias._version)
    cache = getattr(lin, "_fold_cons_cache", None)
    if cache is None or cache[0] != key:
        with torch.no_grad():
            w32 = lin.weight.detach().float()
            w = (w32 * norm.weight.detach().float()[None, :]).to(torch.bfloat16).contiguous()
            c = w.float().sum(1).contiguous()
            d = (w32.double() @ norm.bias.detach().double()).float()
            if lin.bias is not None:
                d = d + lin.bias.detach().float()
            d = d.contiguous()
        cache = (key, w, c, d)
        lin._fold_cons_cache = cache
    return cache


def linear_residual_(x, a, lin, gamma):
    """x (rows, C) fp32 += gamma * lin(a) IN PLACE (a: bf16 (rows, K)); -> (bf16 copy of the updated rows, row partial sums (rows_padded, C/256, 2))."""
    note_mutation()
    _, w, b = _fold_producer_weights(lin, gamma)
    C, K = w.shape
    rows = x.numel() // C
    assert x.dtype == torch.float32 and x.is_contiguous() and a.dtype == torch.bfloat16 and a.numel() == rows * K
    a = _c(a)
    xb = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    stats = torch.empty((rows + 255) // 256 * 256, C // 256, 2, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        call("unopose_linear_bf16_residual", ptr(a), ptr(w), ptr(b), ptr(x), ptr(xb), ptr(stats), rows, C, K, stream_ptr())
    return xb, stats


def linear_lnfold(xb, stats, lin, norm, gelu=False):
    """lin(LayerNorm(x)) [-> GELU] from the un-normalised bf16 rows `xb` and the row partial sums of `linear_residual_`; bf16 out."""
    _, w, c, d = _fold_consumer_weights(lin, norm)
    N, K = w.shape
    rows = xb.numel() // K
    out = torch.empty(*xb.shape[:-1], N, dtype=torch.bfloat16, device=xb.device)
    with torch.cuda.device(xb.device):
        call("unopose_linear_bf16_lnfold", ptr(xb), ptr(w), ptr(d), ptr(c), ptr(stats), K // 256, float(norm.eps), ptr(out), rows, N, K, int(gelu), stream_ptr())
    return out


# ---- round 6: the geometric embedding under autograd on the table kernels (csrc/embed.hip) ---------------------------------------------
TRAIN_OWN_GEO = True  # A/B attribute: False = the op-by-op composite (geo_embedding_torch) under autograd


def _geo_grid(m, rows, npoint):
    """sinus(x_r) on the table grid x_r = (r - (npoint / 2 - 1)) / 4, (rows, 256) float64: T = S W^T is the table, dW = dT^T S its gradient."""
    div = m.embedding.div_term
    key = (rows, npoint, div._version, div.data_ptr())
    cache = m.__dict__.setdefault("_geo_grid_cache", {})
    if key not in cache:  # (depends on the frequencies only: built once, not per training step)
        d64 = div.detach().double()
        x = (torch.arange(rows, device=div.device, dtype=torch.float64) - float(npoint // 2 - 1)) / _GEO_HINV
        om = x[:, None] * d64[None, :]
        cache.clear() if len(cache) > 8 else None
        cache[key] = torch.stack([torch.sin(om), torch.cos(om)], dim=-1).reshape(rows, -1)
    return cache[key]


class _GeoEmbedFn(torch.autograd.Function):
    """GeometricStructureEmbedding.forward (transformer.py:303-350) with gradients for proj_d / proj_a (the points carry none: they are
    data).  Forward = the 6-point table kernel of the fp32 eval path (tables rebuilt from the current weights), which also records the
    arg-max of the three angle terms; backward = its mirror, scattering dE into a table-shaped gradient in LDS (csrc/embed.hip
    geo_embed_table_bwd_kernel), then dW = dT^T S_grid and db = sum_r dT[r] on the host side (two 256-wide matmuls)."""

    @staticmethod
    def forward(ctx, points, wd, bd, wa, ba, m):
        import math

        npoint = 6
        B, n, _ = points.shape
        rows_a = int(math.floor(math.pi * float(m.factor_a) * _GEO_HINV)) + npoint + 1
        rows_d = _GEO_D_RANGE * _GEO_HINV + npoint
        sd, sa = _geo_grid(m, rows_d, npoint), _geo_grid(m, rows_a, npoint)
        td = (sd @ wd.detach().double().t()).float().contiguous()
        ta = (sa @ wa.detach().double().t()).float().contiguous()
        bias = (bd.detach().float() + ba.detach().float()).contiguous()
        wdf = wd.detach().float().contiguous()
        div = m.embedding.div_term.detach().float().contiguous()
        out = torch.empty(B, n, n, 256, dtype=torch.float32, device=points.device)
        amax = torch.empty(B, n, n, 64, dtype=torch.int32, device=points.device)
        knn = torch.empty(B, n, 3, dtype=torch.int32, device=points.device)
        mean = int(m.reduction_a == "mean")
        with torch.cuda.device(points.device):
            call("unopose_geo_embedding_train_forward", ptr(points), B, n, ptr(td), rows_d, ptr(ta), rows_a, ptr(bias), ptr(wdf), ptr(div), _GEO_HINV,
                 float(m.sigma_d), float(m.factor_a), mean, ptr(knn), ptr(out), ptr(amax), stream_ptr())
        ctx.save_for_backward(points, knn, amax, sd, sa)
        ctx.meta = (rows_d, rows_a, float(m.sigma_d), float(m.factor_a), mean, npoint)
        return out

    @staticmethod
    def backward(ctx, dE):
        points, knn, amax, sd, sa = ctx.saved_tensors
        rows_d, rows_a, sigma_d, factor_a, mean, npoint = ctx.meta
        B, n, _ = points.shape
        rd_l = min(rows_d, 16 * _GEO_HINV + npoint - 1)
        dE = _c(dE.float())
        G = lib().unopose_geo_embedding_train_workgroups(B, n)
        ws = torch.empty(G, rd_l + rows_a, 256, dtype=torch.float32, device=dE.device)
        full = torch.zeros(rows_d, 256, dtype=torch.float32, device=dE.device)
        past = torch.zeros(1, dtype=torch.int32, device=dE.device)
        with torch.cuda.device(dE.device):
            call("unopose_geo_embedding_train_backward", ptr(points), ptr(knn), B, n, rows_d, rows_a, _GEO_HINV, sigma_d, factor_a, mean, ptr(dE), ptr(amax),
                 ptr(ws), ptr(full), ptr(past), stream_ptr())
        torch._assert_async(past == 0, "geo embedding backward: a distance index past the table (the clouds are not radius-normalised)")
        dT = ws.double().sum(0)
        dTd = full.double()
        dTd[:rd_l] += dT[:rd_l]
        dTa = dT[rd_l:]
        dwd, dwa = (dTd.t() @ sd).float(), (dTa.t() @ sa).float()
        dbd, dba = dTd.sum(0).float(), dTa.sum(0).float()
        return None, dwd, dbd, dwa, dba, None


def geo_embedding_train_ok(points, m):
    return (TRAIN_OWN_GEO and points.is_cuda and m.proj_d.weight.shape == (256, 256) and m.angle_k == 3 and points.shape[1] >= 4
            and m.proj_d.weight.dtype == torch.float32 and not torch.is_autocast_enabled()
            and int(__import__("math").floor(__import__("math").pi * float(m.factor_a) * _GEO_HINV)) + 7 <= 80 - 6)
