"""Dense linear layers on the hand-written GEMMs (csrc/gemm.hip, gemm_small.hip, gemm_f32.hip, bmm_f32.hip) and the row-wise glue around them
(csrc/fused.hip, glue.hip): nn.Linear under autocast and in fp32 class, the ViT prologue, LayerNorm / LayerScale passes, the residual +
LayerNorm fold, and the trainable form under autograd.  Each function cites the reference Python it replaces."""
import ctypes

import torch
import torch.nn.functional as F

from .._lib import call, lib, on_device, ptr, stream_ptr
from . import _state as st
from .common import _MUTATION_EPOCH, _SPLIT_MEMO, _aligned16, _c, _f32_path, _no_autograd, _params_key, note_fallback, note_mutation


def linear_backend():
    """Which GEMM runs the large bf16 linears (reported by bench.py next to the measured rate)."""
    if st.USE_HIP_GEMM and st.HIP_GEMM_ALL:
        return "csrc/gemm.hip (256x256x64 LDS-DMA tiles, persistent; bias / bias + erf-GELU epilogue) for every ViT linear"
    if st.USE_HIP_GEMM:
        return "hipBLASLt (through torch); fc1 + GELU: csrc/gemm.hip (256x256x64 LDS-DMA tiles, fused bias + erf-GELU epilogue)"
    return "hipBLASLt (through torch)"


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
    with on_device(x2.device):
        call("unopose_linear_bf16", ptr(x2), ptr(w), ptr(bias_f32), ptr(out), M, N, K, 1 if gelu else (2 if relu else 0), stream_ptr())
    return out


def own_gemm_ok(rows, N, K):
    """Does csrc/gemm.hip take this bf16 linear?  With HIP_GEMM_ALL: every shape its tiling admits (N % 256 == 0,
    K % 64 == 0, any row count) -- NO hipBLASLt bf16 kernel is left on the autocast path.  That matters beyond speed: the
    library's stream-K kernels spin on partner workgroups and hang when forwards overlap.  (Round 2 also
    blamed them for wrong sums in other kernels; round 3 traced those to packed-fp32 instructions in the VICTIM kernels beside any
    MFMA kernel -- the library is built without them now, see build.py.)"""
    if not st.USE_HIP_GEMM or N % 256 != 0 or K % 64 != 0 or rows < 1:
        return False
    return st.HIP_GEMM_ALL or rows >= 4096


def bf16_linear_2d(x2, w, bias_f32, bias_bf16=None, relu=False):
    """(rows,K) bf16 @ (N,K)^T + bias on csrc/gemm.hip when `own_gemm_ok`, else the library."""
    rows, K = x2.shape
    N = w.shape[0]
    if own_gemm_ok(rows, N, K):
        return linear_bf16_hip(_c(x2), w, bias_f32, relu=relu)
    y = F.linear(x2, w, bias_bf16 if bias_bf16 is not None else bias_f32.to(torch.bfloat16))
    return F.relu(y) if relu else y


def f32x3_ok(rows, N, K):
    # (operands AND the output: the kernel addresses all three through 32-bit buffer offsets)
    return st.USE_F32X3 and N % 256 == 0 and K % 32 == 0 and rows >= 1 and rows * K * 4 < 2 ** 32 and N * K * 4 < 2 ** 32 and rows * N * 4 < 2 ** 32


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
    with on_device(x2.device):
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
    with on_device(xs.device):
        call("unopose_linear_f32x3", ptr(xs), ptr(ws), ptr(bias_f32), None if C is None else ptr(C), None if Cs is None else ptr(Cs),
             M, N, K, 1 if gelu else (2 if relu else 0), stream_ptr())
    return C if out == "f32" else (Cs if out == "split" else (C, Cs))


def linear_f32x3_bf16(xs, ws, bias_f32, M, N, K, resid=None):
    """C-ABI unopose_linear_f32x3_bf16: bf16( resid + bf16(X W^T + b) ) with X, W in the split layout."""
    out = torch.empty(M, N, dtype=torch.bfloat16, device=xs.device)
    with on_device(xs.device):
        call("unopose_linear_f32x3_bf16", ptr(xs), ptr(ws), ptr(bias_f32), None if resid is None else ptr(resid), ptr(out), M, N, K, stream_ptr())
    return out


def linear_f32_raw(x, w, b, owner, tag):
    """x (...,K) fp32 @ w (N,K)^T + b with cached split weights on `owner` (the fp32 token / linear attention projections, whose
    fused weight matrices are built by their callers); csrc/gemm_f32.hip when the shape fits, else the library."""
    N, K = w.shape
    rows = x.numel() // K
    if not (x.is_cuda and not st._DIFF and f32x3_ok(rows, N, K)):
        return F.linear(x, w, b)
    key = (w.data_ptr(), w._version, None if b is None else (b.data_ptr(), b._version), tag)
    caches = owner.__dict__.setdefault("_f32x3_raw", {})
    c = caches.get(tag)
    if c is None or c[0] != key:
        with torch.no_grad():
            c = (key, split_f32(w.detach().float().contiguous()), torch.zeros(N, device=w.device) if b is None else b.detach().float().contiguous())
        caches[tag] = c
    return linear_f32x3(split_f32(_c(x.float()).reshape(rows, K), memo=True), c[1], c[2], rows, N, K).reshape(*x.shape[:-1], N)


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
            if (st.TRAIN_OWN_WGRAD and not ctx.bf16 and g.dtype == torch.float32 and x2.dtype == torch.float32 and g.is_cuda and N % 128 == 0
                    and K % 128 == 0 and rows >= st.TRAIN_OWN_WGRAD_MIN_ROWS and _aligned16(g) and _aligned16(x2)):
                # dW = dY^T X on csrc/conv_train.hip::linear_wgrad_f32_kernel (fp32 matrix instruction straight from the row-major operands)
                gc = _c(g)
                splits = lib().unopose_linear_wgrad_f32_splits(rows, N, K)
                ws = torch.empty(splits * N * K, dtype=torch.float32, device=g.device)
                gw = torch.empty(N, K, dtype=torch.float32, device=g.device)
                with on_device(g.device):
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
    ok = st.TRAIN_OWN_GEMM and x.is_cuda and rows > 0 and 2.0 * rows * N * K >= st.TRAIN_OWN_GEMM_MIN_FLOP and \
        (own_gemm_ok(rows, N, K) if bf16 else (x.dtype == torch.float32 and f32x3_ok(rows, N, K)))
    if not ok:
        y = lin(x)
        return F.relu(y) if relu else y
    return _LinearFn.apply(x, lin.weight, lin.bias, lin, relu, bf16)


def _lin(x, lin):
    """Inside the op-by-op composites: the trainable form in differentiable mode, the plain module call otherwise (the
    composites stay library-only A/B references of the fused kernels)."""
    return linear_train(x, lin) if (st._DIFF and torch.is_grad_enabled()) else lin(x)


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
    if st._DIFF and not gelu and torch.is_grad_enabled():
        return linear_train(x, lin, relu)
    if st._DIFF or not (torch.is_autocast_enabled() and x.is_cuda):
        if st.FORBID_LIBRARY_BF16_GEMM and x.is_cuda and not st._DIFF:
            raise RuntimeError(f"ops.linear: a {tuple(x.shape)} {x.dtype} -> {lin.weight.shape[0]} linear does not fit csrc/gemm_f32.hip (N % 256, K % 32, fp32 "
                               "without autograd) and would go to a library GEMM while several forwards are in flight (PipelinedForward, depth > 1).  "
                               "Use depth=1 for this model configuration")
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
        if own_gemm_ok(rows, N, K) and (st.HIP_GEMM_ALL or gelu):
            return linear_bf16_hip(_c(xb).reshape(rows, K), cache[1], cache[3], gelu, relu).reshape(*xb.shape[:-1], N)
        if st.FORBID_LIBRARY_BF16_GEMM:
            raise RuntimeError(f"ops.linear: a {rows} x {K} -> {N} bf16 linear does not fit csrc/gemm.hip (N % 256, K % 64) and would go to "
                               "a library GEMM while several forwards are in flight (PipelinedForward, depth > 1): library stream-K "
                               "kernels of two streams can starve each other.  Use depth=1 for this model configuration")
        note_fallback("linear", f"{rows} x {K} -> {N} bf16 (own GEMM: N % 256 == 0, K % 64 == 0): library GEMM")
        if relu and cache[2] is not None:
            x2 = xb.reshape(-1, xb.shape[-1])
            return torch._addmm_activation(cache[2], x2, cache[1].t()).reshape(*xb.shape[:-1], cache[1].shape[0])
        y = F.linear(xb, cache[1], cache[2])
        return F.relu(y) if relu else (F.gelu(y) if gelu else y)


def ffn_add_layernorm(x, expand, squeeze, norm, out=None):
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
        if out is not None and out.dtype == x.dtype and out.shape == x.shape and out.stride(-1) == 1 and out.is_contiguous():
            return add_layernorm(y, x, norm, out=out)  # the LayerNorm kernel writes the caller's destination itself
        return _into(out, add_layernorm(y, x, norm))
    return linear_add_layernorm(linear(x, expand, relu=True), squeeze, x, norm, out=out)


def _into(out, y):
    """y, or `out` filled with y when the caller named a destination the producing kernel could not write itself."""
    if out is None:
        return y
    note_mutation()
    out.copy_(y)
    return out


def linear_add_layernorm(h, lin, x, norm, out=None):
    """LayerNorm(lin(h) + x): the post-LN glue after an attention output projection / FFN squeeze (transformer.py:151-193).
    256-wide layers under autocast run as ONE launch -- residual add and LayerNorm in the epilogue of csrc/gemm.hip, on the
    fp32 accumulators (the unfused form rounds lin(h) to bf16 first); everything else: add_layernorm(linear(h), x).
    `out`: a contiguous tensor of the result's shape the result is written INTO (the kernel's own output buffer when it is bf16: the
    cross layers of a matcher block fill the two halves of one stacked tensor instead of being concatenated afterwards)."""
    N, K = lin.weight.shape
    rows = h.numel() // K
    if (st.USE_FUSED_LINEAR_LN and not st._DIFF and h.is_cuda and torch.is_autocast_enabled() and st.HIP_GEMM_ALL and N == 256
            and own_gemm_ok(rows, N, K) and tuple(norm.normalized_shape) == (256,) and norm.weight is not None and norm.bias is not None and lin.bias is not None):
        cache = _bf16_weights(lin)
        with torch.autocast("cuda", enabled=False):
            hb = _c(h if h.dtype == torch.bfloat16 else h.to(torch.bfloat16)).reshape(rows, K)
            xb = _c(x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16)).reshape(rows, N)
            direct = out is not None and out.dtype == torch.bfloat16 and out.is_contiguous() and out.numel() == rows * N and out.device == h.device
            if direct:
                note_mutation()
            y = out if direct else torch.empty(rows, N, dtype=torch.bfloat16, device=h.device)
            with on_device(h.device):
                call("unopose_linear_add_layernorm_bf16", ptr(hb), ptr(cache[1]), ptr(cache[3]), ptr(xb), ptr(norm.weight.detach()),
                     ptr(norm.bias.detach()), float(norm.eps), ptr(y), rows, K, stream_ptr())
        return out if direct else _into(out, y.reshape(*h.shape[:-1], N))
    if out is not None and h.is_cuda and not st._DIFF and out.shape == x.shape and out.is_contiguous() and out.dtype in (torch.float32, torch.bfloat16):
        y = linear(h, lin)
        if y.dtype in (torch.float32, torch.bfloat16) and (out.dtype == (torch.bfloat16 if torch.is_autocast_enabled() else y.dtype)):
            return add_layernorm(y, x, norm, out=out)
        return _into(out, add_layernorm(y, x, norm))
    return _into(out, add_layernorm(linear(h, lin), x, norm))


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
    if st._DIFF or not (st.HIP_GEMM_ALL and patches.is_cuda and torch.is_autocast_enabled() and own_gemm_ok(rows, D, 64)):
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
    return (not st._DIFF and xa.is_cuda and torch.is_autocast_enabled() and st.HIP_GEMM_ALL and st.USE_HIP_GEMM and xa.dtype == torch.float32
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
    with torch.autocast("cuda", enabled=False), on_device(dev):
        a = torch.empty((na + nb) * P, Kp, dtype=torch.bfloat16, device=dev)
        call("unopose_patchify_bf16", ptr(_c(xa)), na, None if xb is None else ptr(_c(xb)), nb, S, Kp, ptr(a), stream_ptr())
        y = linear_bf16_hip(a, wp, b)
        x = torch.empty(na + nb, npre + P, D, dtype=torch.float32, device=dev)
        n1 = torch.empty(na + nb, npre + P, D, dtype=torch.bfloat16, device=dev)
        call("unopose_vit_tokens_layernorm", ptr(y), ptr(vit.pos_embed.detach().float().contiguous()), ptr(prefix), npre, P, na + nb, D,
             ptr(norm1.weight.detach()), ptr(norm1.bias.detach()), float(norm1.eps), ptr(x), ptr(n1), stream_ptr())
    return x, n1


def vit_prologue_f32_ok(xa, vit):
    """The fused prologue of the no-autocast ViT (fp32-class GEMM on split operands), for ViT-B (768 wide)."""
    if not (not st._DIFF and _no_autograd() and xa.is_cuda and not torch.is_autocast_enabled() and st.USE_F32X3 and xa.dtype == torch.float32
            and vit.pos_embed.shape[-1] == 768 and xa.shape[-1] == xa.shape[-2] and xa.shape[-1] % 14 == 0
            and vit.pos_embed.shape[1] == (xa.shape[-1] // 14) ** 2):
        return False
    rows = 2 * xa.shape[0] * (vit.pos_embed.shape[1] + 5)
    return f32x3_ok(rows, 768, 608) and f32x3_ok(rows, 3 * 768, 768) and f32x3_ok(rows, 768, 4 * 768)


def vit_prologue_f32(xa, xb, vit, norm1):
    """`vit_prologue` at the reference's default precision: (x fp32 (n,T,768) residual stream, norm1(x) in the split layout of
    csrc/gemm_f32.hip).  Patches straight into the split-layout patch matrix (K = 588 zero-padded to 608), the patch embedding on the
    fp32-class GEMM, pos_embed / class + register tokens / first LayerNorm in one pass.  Replaces cat([rgb, tem_rgb]) + unfold copy + zeros +
    copy + split pass + add + cat + LayerNorm (1.4 GB written per forward at 64 x 518 x 518)."""
    conv = vit.patch_embed.proj
    w = conv.weight.reshape(conv.weight.shape[0], -1)
    D, K = w.shape
    Kp = (K + 31) // 32 * 32
    key = _params_key(conv, Kp, "f32", vit.cls_token._version, vit.reg_token._version, vit.pos_embed._version)
    cache = getattr(conv, "_prologue_cache_f32", None)
    if cache is None or cache[0] != key:
        with torch.no_grad():
            wp = torch.zeros(D, Kp, dtype=torch.float32, device=w.device)
            wp[:, :K] = w.detach()
            b = torch.zeros(D, device=w.device) if conv.bias is None else conv.bias.detach().float().contiguous()
            prefix = torch.cat([vit.cls_token.detach().float().reshape(-1, D), vit.reg_token.detach().float().reshape(-1, D)], 0).contiguous()
            cache = (key, split_f32(wp), b, prefix, vit.pos_embed.detach().float().contiguous())
        conv._prologue_cache_f32 = cache
    _, ws, b, prefix, pos = cache
    na, nb = xa.shape[0], 0 if xb is None else xb.shape[0]
    S = xa.shape[-1]
    P = (S // 14) ** 2
    npre = prefix.shape[0]
    dev = xa.device
    note_mutation()
    with on_device(dev):
        a = torch.empty((na + nb) * P, 2 * Kp, dtype=torch.bfloat16, device=dev)
        call("unopose_patchify_split", ptr(_c(xa)), na, None if xb is None else ptr(_c(xb)), nb, S, Kp, ptr(a), stream_ptr())
        y = linear_f32x3(a, ws, b, (na + nb) * P, D, Kp)
        x = torch.empty(na + nb, npre + P, D, dtype=torch.float32, device=dev)
        n1 = torch.empty((na + nb) * (npre + P), 2 * D, dtype=torch.bfloat16, device=dev)
        call("unopose_vit_tokens_layernorm_f32", ptr(y), ptr(pos), ptr(prefix), npre, P, na + nb, D, ptr(norm1.weight.detach()),
             ptr(norm1.bias.detach()), float(norm1.eps), ptr(x), ptr(n1), stream_ptr())
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
    with on_device(a.device):
        call("unopose_bmm_f32", ptr(a4), a4.stride(0), a4.stride(1), a4.stride(2), a4.stride(3), ptr(b4), b4.stride(0), b4.stride(1),
             b4.stride(2), b4.stride(3), ptr(out), Bo, Bi, n, m, K, float(alpha), stream_ptr())
    return out if a.dim() == 4 else out[:, 0]


def score_head(x, lin):
    """The overlap-score head nn.Linear(d, 1) (C:66, Fi:89).  One output channel is no GEMM shape for csrc/gemm.hip, and no
    library bf16 GEMM may be on the autocast path (`own_gemm_ok`): evaluated as a multiply + row sum in fp32 on the
    bf16-rounded weights, rounded to the dtype the autocast Linear would return."""
    if _f32_path(x) and st.USE_F32X3 and x.shape[-1] == 256:  # fp32: the same row dot on the unrounded weights
        key = (lin.weight._version, lin.weight.data_ptr(), None if lin.bias is None else lin.bias._version, "f32")
        cache = getattr(lin, "_rowdot_cache_f32", None)
        if cache is None or cache[0] != key:
            cache = (key, lin.weight.detach().float().reshape(-1).contiguous(), 0.0 if lin.bias is None else float(lin.bias.detach().float().item()))
            lin._rowdot_cache_f32 = cache
        xc = _c(x)
        out = torch.empty(*x.shape[:-1], 1, dtype=torch.float32, device=x.device)
        with on_device(x.device):
            call("unopose_row_dot", ptr(xc), 0, ptr(cache[1]), cache[2], xc.numel() // 256, 256, ptr(out), 0, stream_ptr())
        return out
    if st._DIFF or not (st.HIP_GEMM_ALL and x.is_cuda and torch.is_autocast_enabled()):
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
            with on_device(x.device):
                call("unopose_row_dot", ptr(xc), int(x.dtype == torch.bfloat16), ptr(cache[1]), cache[2], xc.numel() // 256, 256, ptr(out), 1,
                     stream_ptr())
            return out
        return ((x.float() * cache[1]).sum(-1, keepdim=True) + cache[2]).to(torch.bfloat16)


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
    with on_device(a.device):
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
    with on_device(x.device):
        call("unopose_scale_residual", ptr(x), ptr(y), ptr(gamma), x.numel() // C, C, stream_ptr())
    return x


def scale_residual_layernorm_f32_(x, y, gamma, norm, wide=None, block=0):
    """fp32 twin for the no-autocast path: x (fp32, contiguous) += gamma * y (fp32) in place (y None: no update); returns
    LayerNorm(x) in the split layout of csrc/gemm_f32.hip as a (rows, 2C) bf16 tensor (norm None: residual update only, returns x).
    `wide` (rows, 2 n C) bf16, `block`: the result goes into column block `block` of that wider split-layout matrix instead (the tap
    LayerNorms of ViT_AE side by side: the K = n C operand of the up-projection without a concatenation or a split pass); returns wide."""
    note_mutation()
    assert x.dtype == torch.float32 and x.is_contiguous() and (y is None or y.dtype == torch.float32)
    C = x.shape[-1]
    rows = x.numel() // C
    ld, dst = 0, None
    if wide is not None:
        assert norm is not None and wide.dtype == torch.bfloat16 and wide.is_contiguous() and wide.shape[0] == rows and wide.shape[1] % (2 * C) == 0
        ld = wide.shape[1] * 2
        assert 0 <= block < wide.shape[1] // (2 * C)
        dst = ctypes.c_void_p(wide.data_ptr() + block * C * 4)
    out = None if (norm is None or wide is not None) else torch.empty(rows, 2 * C, dtype=torch.bfloat16, device=x.device)
    with on_device(x.device):
        call("unopose_scale_residual_layernorm_f32", ptr(x), None if y is None else ptr(_c(y)), None if y is None else ptr(gamma),
             None if norm is None else ptr(norm.weight), None if norm is None else ptr(norm.bias), rows, C,
             0.0 if norm is None else float(norm.eps), dst if wide is not None else (None if out is None else ptr(out)), ld, stream_ptr())
    return x if norm is None else (wide if wide is not None else out)


def vit_f32_fused_ok(x, vit):
    """The no-autocast ViT on the fused fp32-class path (split-layout LayerNorm outputs feeding csrc/gemm_f32.hip directly)?"""
    C = x.shape[-1]
    return (x.is_cuda and x.dtype == torch.float32 and not st._DIFF and _no_autograd() and not torch.is_autocast_enabled() and st.USE_F32X3 and C % 32 == 0
            and C <= 1024 and C % 64 == 0 and f32x3_ok(x.numel() // C, 3 * C, C) and f32x3_ok(x.numel() // C, C, 4 * C))


def scale_residual_layernorm_(x, y, gamma, norm):
    """x (fp32) += gamma * y (bf16) in place; returns LayerNorm(x) in bf16 -- one pass over the residual stream."""
    note_mutation()
    assert x.dtype == torch.float32 and x.is_contiguous() and y.dtype == torch.bfloat16
    y = _c(y)
    C = x.shape[-1]
    out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    with on_device(x.device):
        call("unopose_scale_residual_layernorm", ptr(x), ptr(y), ptr(gamma), ptr(norm.weight), ptr(norm.bias),
             x.numel() // C, C, float(norm.eps), ptr(out), stream_ptr())
    return out


def ln_fold_ok(rows, C):
    """The fold runs on the 256 x 256-tile kernel only: shapes whose proj / fc2 grid the small-tile kernel would take keep the separate pass."""
    if not (st.USE_LN_FOLD and st.USE_HIP_GEMM and st.HIP_GEMM_ALL) or C % 256 != 0 or C // 256 > 4 or rows * C * 4 >= 2 ** 31:
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
    with on_device(x.device):
        call("unopose_linear_bf16_residual", ptr(a), ptr(w), ptr(b), ptr(x), ptr(xb), ptr(stats), rows, C, K, stream_ptr())
    return xb, stats


def linear_lnfold(xb, stats, lin, norm, gelu=False):
    """lin(LayerNorm(x)) [-> GELU] from the un-normalised bf16 rows `xb` and the row partial sums of `linear_residual_`; bf16 out."""
    _, w, c, d = _fold_consumer_weights(lin, norm)
    N, K = w.shape
    rows = xb.numel() // K
    out = torch.empty(*xb.shape[:-1], N, dtype=torch.bfloat16, device=xb.device)
    with on_device(xb.device):
        call("unopose_linear_bf16_lnfold", ptr(xb), ptr(w), ptr(d), ptr(c), ptr(stats), K // 256, float(norm.eps), ptr(out), rows, N, K, int(gelu), stream_ptr())
    return out
