"""Autograd Functions of the training step on own kernels: InfoNCE, BatchNorm + ReLU (+ max-pool), saliency, nearest partner, the PE's 1 x 1
convolutions (csrc/bn_train.hip, conv_train.hip, saliency_train.hip)."""
import torch
import torch.nn.functional as F

from .._lib import call, lib, on_device, ptr, stream_ptr
from . import _state as st
from .common import _aligned16, _c


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
        with on_device(x.device):
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
        with on_device(x.device):
            call("unopose_infonce_grad", ptr(x), B, R, C, ptr(ws), ptr(_c(label1)), ptr(_c(label2)), ptr(_c(g.float())), ptr(grad), stream_ptr())
        return grad, None, None


def infonce_two_way(atten, label1, label2):
    """The "atten" loss of compute_overlap_loss for one transformer block -> (B,)."""
    if st.USE_FUSED_INFONCE and atten.is_cuda and atten.shape[1] <= 65535:
        return _InfoNCEFn.apply(atten, label1, label2)
    a = atten.float()
    l1 = F.cross_entropy(a.transpose(1, 2)[:, :, 1:], label1, reduction="none").mean(1)  # classes = columns, per query row
    l2 = F.cross_entropy(a[:, :, 1:], label2, reduction="none").mean(1)
    return 0.5 * (l1 + l2)


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
        with on_device(x.device):
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
        with on_device(x.device):
            call("unopose_bn_relu_train_backward", ptr(x), ptr(dy), B, C, L, ptr(w), ptr(b_), ptr(mean), ptr(rstd), ptr(ws), ptr(dgamma), ptr(dbeta),
                 ptr(dx), stream_ptr())
        return dx, dgamma, dbeta, None


def bn_relu(x, bn):
    """F.relu(bn(x)) for an nn.BatchNorm2d: the fused training form (csrc/bn_train.hip) when `bn` is in train mode on fp32 CUDA data
    with affine parameters, else the modules themselves (eval statistics, CPU, other dtypes)."""
    L = x[0, 0].numel() if x.dim() >= 3 else 0
    if (st.USE_FUSED_BN_RELU and bn.training and type(bn) is torch.nn.BatchNorm2d and x.is_cuda and x.dtype == torch.float32 and x.dim() >= 3
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
        with on_device(x.device):
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
        with on_device(x.device):
            call("unopose_bn_relu_maxpool_train_backward", ptr(x), ptr(g), ptr(idx), B, C, N, S, ptr(w), ptr(b_), ptr(mean), ptr(rstd), ptr(ws),
                 ptr(dgamma), ptr(dbeta), ptr(dx), stream_ptr())
        return dx, dgamma, dbeta, None


def bn_relu_maxpool(x, bn):
    """F.relu(bn(x)).max(dim=3)[0] for x (B, C, N, S): fused (csrc/bn_train.hip) when `bn` is in train mode on fp32 CUDA data with
    S in {32, 64, 128, 256}, else `bn_relu` followed by torch's max."""
    if (st.USE_FUSED_BN_RELU and bn.training and type(bn) is torch.nn.BatchNorm2d and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
            and x.shape[3] in (32, 64, 128, 256) and bn.weight is not None and bn.bias is not None and x.shape[0] <= 65535
            and x.shape[1] <= 65535 and _aligned16(x)):
        return _BNReLUMaxPoolTrain.apply(x, bn.weight, bn.bias, bn)[0]
    return bn_relu(x, bn).max(dim=3)[0]


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
        with on_device(dev):
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
        with on_device(a.device):
            call("unopose_saliency_train_backward", ptr(a), ptr(v1), ptr(v2), ptr(m1), ptr(m2), ptr(rmax), ptr(rsum), ptr(cmax), ptr(csum), ptr(g1),
                 ptr(g2), B, n1, n2, ptr(da), ptr(ds1), ptr(ds2), stream_ptr())
        ad, d1, d2, sh1, sh2 = ctx.meta
        return da.to(ad), ds1.reshape(sh1).to(d1), ds2.reshape(sh2).to(d2)


def saliency_pair(atten, s1, s2):
    """m1 = softmax(atten[:, 1:, 1:], dim=2) @ s2 and m2 = softmax(atten[:, 1:, 1:].transpose(1, 2), dim=2) @ s1 (s1 (B, n1, 1),
    s2 (B, n2, 1)): fused with its backward on fp32 CUDA data, the reference's expression otherwise."""
    if st.TRAIN_FUSED_SALIENCY and atten.is_cuda and atten.dtype == torch.float32 and s1.shape[-1] == 1 and s2.shape[-1] == 1 and atten.dim() == 3:
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
    with on_device(a.device):
        call("unopose_nearest_partner", ptr(a), ptr(b), B, n, m, int(over_b), float(thr), ptr(d), ptr(idx), ptr(anyc), stream_ptr())
    return d, idx.long(), anyc.bool()


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
        with on_device(x.device):
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
        with on_device(x.device):
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
    if (st.TRAIN_OWN_CONV and x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and conv.bias is None
            and w.shape[2:] == (1, 1) and tuple(conv.stride) == (1, 1) and tuple(conv.padding) == (0, 0) and tuple(conv.dilation) == (1, 1)
            and conv.groups == 1 and x.dim() == 4 and L % 64 == 0 and 64 <= L < (1 << 28) and _conv1x1_pair_ok(cin, cout)
            and (not x.requires_grad or _conv1x1_pair_ok(cout, cin)) and (not w.requires_grad or _conv1x1_wgrad_ok(cin, cout))
            and cout in (32, 64, 128) and _aligned16(x)):
        return _Conv1x1Fn.apply(x, w)
    return conv(x)
