"""Attention cores: timm ViT attention (csrc/vit_attn*.hip), the correspondence transformer's RPE / cross token attention (csrc/attn*.hip) and the
focused linear attention (csrc/linattn.hip), each with its op-by-op torch composite (training path, fall-back)."""
import ctypes

import torch
import torch.nn.functional as F

from .._lib import call, lib, on_device, ptr, stream_ptr
from . import _state as st
from .common import _c, _own_f32, _own_glue, _params_key, note_fallback
from .dense import _lin, bf16_linear_2d, bmm_nt_f32, linear, linear_f32_raw


def vit_attention(qkv, heads):
    """timm Attention core: qkv (B,T,3C) -> (B,T,C).  bf16 (autocast) inputs run the flash-style HIP
    kernel (csrc/vit_attn.hip, head dim 64); fp32 inputs the op-by-op composite."""
    if st._DIFF:
        return vit_attention_torch(qkv, heads)
    if qkv.dtype == torch.bfloat16 and qkv.shape[-1] == 3 * heads * 64:
        B, T, C3 = qkv.shape
        qkv = _c(qkv)
        out = torch.empty(B, T, C3 // 3, dtype=torch.bfloat16, device=qkv.device)
        with on_device(qkv.device):
            call("unopose_vit_attention", ptr(qkv), B, T, heads, ptr(out), stream_ptr())
        return out
    if qkv.dtype == torch.float32 and qkv.is_cuda and qkv.shape[-1] == 3 * heads * 64:
        B, T, C3 = qkv.shape
        qkv = _c(qkv)
        out = torch.empty(B, T, C3 // 3, dtype=torch.float32, device=qkv.device)
        with on_device(qkv.device):
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
    with on_device(qkv.device):
        call("unopose_vit_attention_f32_split", ptr(qkv), B, T, heads, ptr(out), stream_ptr())
    return out


def vit_attention_f32_ss(qkv_split, B, T, heads):
    """qkv in the split layout ((B*T, 2 * 3C) bf16, as `linear_f32x3(..., out="split")` writes it) -> the attention output in the
    split layout ((B*T, 2C) bf16): csrc/vit_attn_f32s.hip, the fp32 ViT block's attention core (no fp32 tensor in between)."""
    C3 = qkv_split.shape[-1] // 2
    assert qkv_split.dtype == torch.bfloat16 and qkv_split.is_cuda and qkv_split.is_contiguous() and C3 == 3 * heads * 64
    out = torch.empty(B * T, 2 * (C3 // 3), dtype=torch.bfloat16, device=qkv_split.device)
    with on_device(qkv_split.device):
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


_KEY_PAD = None


def token_attention(x, mem, att, heads, embed=None):
    """MultiHeadAttention / RPEMultiHeadAttention core (transformer.py:130-148, 386-405): returns the
    concatenated heads (B,n,C) before the output Linear.  The RPE term q.proj_p(E) is folded:
    q.(W_p e + b_p) = (q W_p).e + q.b_p  (SURVEY.md App-F), so no (B,4,n,m,64) tensor exists.
    Under autocast(bf16) the whole core (q k^T, folded RPE term, softmax, P v) is ONE HIP kernel on the
    bf16 matrix cores (csrc/attn.hip) that streams E once; in fp32 the op-by-op composite below runs."""
    global _KEY_PAD
    if not st._DIFF and x.is_cuda and heads == 4 and x.shape[-1] == 256:
        if _KEY_PAD is None:
            from .._lib import lib
            _KEY_PAD = lib().unopose_token_attention_key_pad()
        if mem.shape[1] <= _KEY_PAD:
            if torch.is_autocast_enabled():
                return _token_attention_hip(x, mem, att, embed)
            if x.dtype == torch.float32:
                return _token_attention_hip_f32(x, mem, att, embed)
    if not st._DIFF and x.is_cuda:
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
    if ykv.is_contiguous() and C % 32 == 0 and m <= _KEY_PAD:  # V^T, zero-padded to the kernel's key count: one launch (csrc/glue.hip)
        vt = torch.empty(B, C, _KEY_PAD, dtype=torch.float32, device=x.device)
        with on_device(x.device):
            call("unopose_transpose_pad_f32", ctypes.c_void_p(ykv.data_ptr() + C * 4), ykv.stride(1), B, m, C, _KEY_PAD, ptr(vt), stream_ptr())
    else:
        vt = torch.zeros(B, C, _KEY_PAD, dtype=torch.float32, device=x.device)
        vt[:, :, :m] = ykv[..., C:].transpose(1, 2)
    E = _c(embed.float()) if rpe else None
    out = torch.empty(B, n, C, dtype=torch.float32, device=x.device)
    qptr, kptr = yq.data_ptr(), ykv.data_ptr()
    with on_device(x.device):
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
    vt = torch.empty(B, C, _KEY_PAD, dtype=bf, device=x.device)

    def project_kv(a2, w, b32, rows):
        """a2 @ w^T + b with the V columns (the last 256) written straight into `vt` (csrc/gemm_small.hip EPI 4): no transpose launch"""
        y = torch.empty(rows, w.shape[0], dtype=bf, device=x.device)
        with on_device(x.device):
            call("unopose_linear_bf16_kv_vt", ptr(a2), ptr(w), ptr(b32), ptr(y), ptr(vt), rows, w.shape[0], C, m, _KEY_PAD, stream_ptr())
        return y

    fused_vt = st.USE_KV_VT and C == 256 and _KEY_PAD - m <= 64 and st.USE_HIP_GEMM and st.HIP_GEMM_ALL
    with torch.autocast("cuda", enabled=False):
        if mem is x:  # self-attention: one GEMM for q | qp | k | v
            if fused_vt:
                y = project_kv(_c(xb).reshape(B * n, C), w_all, ball32, B * n).reshape(B, n, -1)
            else:
                y = bf16_linear_2d(xb.reshape(B * n, C), w_all, ball32, b_all).reshape(B, n, -1)
            yq, ykv = y[..., :nq], y[..., nq:]
        else:
            yq = bf16_linear_2d(xb.reshape(B * n, C), w_q, bq32, b_q).reshape(B, n, -1)
            if fused_vt:
                ykv = project_kv(_c(mem.to(bf)).reshape(B * m, C), w_kv, bkv32, B * m).reshape(B, m, -1)
            else:
                ykv = bf16_linear_2d(mem.to(bf).reshape(B * m, C), w_kv, bkv32, b_kv).reshape(B, m, -1)
    # q | qp and k | v are consumed in place from the projection outputs (row strides passed to the kernel)
    if not fused_vt:
        with on_device(x.device):
            call("unopose_transpose_pad_bf16", ctypes.c_void_p(ykv.data_ptr() + C * 2), ykv.stride(1), B, m, C, _KEY_PAD, ptr(vt), stream_ptr())
    E = _c(embed.to(bf)) if rpe else None
    out = torch.empty(B, n, C, dtype=bf, device=x.device)
    esz = 2
    q_ptr = yq.data_ptr()
    k_ptr = ykv.data_ptr()
    with on_device(x.device):
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


def focused_linear_attention(xq, xkv, att, heads, focusing, kv_skip=0):
    """LinearAttention.forward (transformer.py:533-568).  With 4 heads x 64 the focusing + per-head
    contraction + z scaling run in ONE HIP kernel per side (csrc/linattn.hip): bf16 MFMAs under autocast,
    hi/lo-split (fp32-class) MFMAs on fp32 data; other shapes take the op-by-op composite.
    `kv_skip`: leading rows of every batch of xkv that are not keys / values (xkv[:, kv_skip:] is meant; the bf16 kernels read the
    window in place, the other paths slice)."""
    if not st._DIFF and heads == 4 and xq.shape[-1] == 256 and xq.is_cuda and float(focusing) == 3.0:
        if torch.is_autocast_enabled():
            if kv_skip and not (st.USE_LA_KV_STATE and xkv.is_contiguous()):
                xkv, kv_skip = xkv[:, kv_skip:], 0
            return _focused_linear_attention_hip(xq, xkv, att, int(focusing), kv_skip)
    if kv_skip:
        xkv = xkv[:, kv_skip:]
    if not st._DIFF and heads == 4 and xq.shape[-1] == 256 and xq.is_cuda and float(focusing) == 3.0:
        if xq.dtype == torch.float32 and xkv.dtype == torch.float32:
            return _focused_linear_attention_hip_f32(xq, xkv, att, int(focusing))
    if not st._DIFF and xq.is_cuda:
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
    with on_device(xq.device):
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


def _focused_linear_attention_hip(xq, xkv, att, focusing, kv_skip=0):
    bf = torch.bfloat16
    B, N, C = xq.shape
    jr = xkv.shape[1]  # rows per pair in memory: the projection runs over all of them, the state over [kv_skip, jr)
    j = jr - kv_skip
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
        ykv = bf16_linear_2d(xkv.to(bf).reshape(B * jr, C), w_kv, bkv32, b_kv).reshape(B, jr, -1)
    out = torch.empty(B, N, C, dtype=bf, device=xq.device)
    if st.USE_LA_KV_STATE and C == 256:
        # focused keys, their sum and k^T v in one launch, read from the projection's rows in place (csrc/linattn.hip)
        kvt = torch.empty(B, 4, 64, 64, dtype=bf, device=xq.device)
        ksum = torch.empty(B, C, dtype=torch.float32, device=xq.device)
        with on_device(xq.device):
            call("unopose_linear_attention_kv_state", ptr(ykv), ptr(inv_sp), B, j, jr, kv_skip, focusing, ptr(kvt), ptr(ksum), stream_ptr())
            call("unopose_linear_attention", ptr(q), ptr(inv_sp), ptr(kvt), ptr(ksum), B, N, focusing, 0, ptr(out), stream_ptr())
        return out
    assert kv_skip == 0
    kproj, v = _c(ykv[..., :C]), ykv[..., C:]
    kf = torch.empty(B, j, C, dtype=bf, device=xq.device)
    with on_device(xq.device):
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
