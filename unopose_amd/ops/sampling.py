"""Pixel features: bilinear sampling of the up-projected ViT map at the chosen pixels, and the sparse up-projection that computes only the map cells
those pixels read (csrc/upproj.hip, fused.hip)."""
import torch

from .._lib import call, on_device, ptr, stream_ptr
from . import _state as st
from .common import _c, note_mutation
from .dense import _bf16_weights


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
    with on_device(z.device):
        call("unopose_bilinear_sample_tokens", ptr(z), int(z.dtype == torch.bfloat16), ptr(choose), B, side, Np, int(H),
             int(W), int(tok_offset), int(tok_stride), ptr(out), stream_ptr())
    return out


def sparse_upproj_ok(x):
    """True when `upproj_plan` + `sparse_pixel_features` may stand in for the dense up-projection + pixel sampling:
    HIP device, autocast (bf16 operands, as the dense autocast GEMM), inference."""
    return st.USE_SPARSE_UPPROJ and x.is_cuda and torch.is_autocast_enabled() and not st._DIFF


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
    with on_device(dev):
        call("unopose_upproj_plan", ptr(choose), B2, Np, int(H), int(W), int(side), int(tok_offset), int(tok_stride), cap_rows,
             ptr(plan["ws"]), ptr(plan["row_list"]), ptr(plan["cellmap"]), ptr(plan["tile_info"]), stream_ptr())
    return plan


def sparse_pixel_features(acts, lin, plan, out=None):
    """ViT_AE's Linear 3072->4096 + pixel shuffle + bilinear upsampling + pixel gather (oneref_feature_extraction.py:
    200-236, model_utils.py:215-227) evaluated only where the chosen pixels look: acts (B2, tok_stride, K) bf16 token
    activations (prefix tokens in place), lin the up-projection -> (B2, Np, 256) fp32, or (`PIXEL_FEATS_BF16`, the default) the same
    values rounded to bf16 -- what the autocast Linears that consume them (coarse / fine in_proj) would make of them first.  Same bf16
    operands, fp32 accumulation and bf16 rounding of the cell values as the dense path (`linear` + `bilinear_sample_native`)."""
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
    if out is None:  # bf16 under the switch: the consumers are autocast Linears (coarse / fine in_proj), which would cast it first
        out = torch.empty(B2, Np, 256, dtype=torch.bfloat16 if st.PIXEL_FEATS_BF16 else torch.float32, device=dev)
    assert out.shape == (B2, Np, 256) and out.dtype in (torch.float32, torch.bfloat16) and out.is_contiguous()
    with on_device(dev):
        cells = torch.empty(plan["cap_rows"], 256, dtype=torch.bfloat16, device=dev)
        call("unopose_linear_bf16_gather", ptr(acts), B2 * ts, K, ptr(w), N, ptr(bias), ptr(plan["row_list"]),
             ptr(plan["tile_info"]), plan["cap_rows"] // 256, ptr(cells), stream_ptr())
        call("unopose_bilinear_sample_compact", ptr(cells), ptr(plan["cellmap"]), ptr(plan["choose"]), B2, plan["side"], Np,
             plan["H"], plan["W"], ptr(out), int(out.dtype == torch.bfloat16), stream_ptr())
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
