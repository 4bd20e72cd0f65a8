"""GPU: the single-pass glue kernels of csrc/glue.hip (and the placed / split outputs of csrc/pe.hip, the bf16 + residual epilogue of
csrc/gemm_f32.hip) against the torch op chains they replace -- bit-exact where the arithmetic is the same, else at the result's
own resolution."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _vit(side=16):
    from unopose_amd.model.modules import ViT

    torch.manual_seed(0)
    v = ViT(768, 12, 12, 14, 14 * side).cuda().eval()
    with torch.no_grad():
        for p in v.parameters():
            p.normal_(0, 0.05)
        for b in v.blocks:
            b.norm1.weight.add_(1.0)
            b.norm2.weight.add_(1.0)
        v.norm.weight.add_(1.0)
    return v


@torch.no_grad()
def test_vit_prologue_equals_the_torch_chain():
    """patchify -> patch GEMM -> pos_embed / prefix tokens / first LayerNorm vs unfold + zero-pad + GEMM + add + cat + LayerNorm."""
    from unopose_amd import ops

    v = _vit()
    g = torch.Generator().manual_seed(1)
    xa, xb = torch.randn(3, 3, 224, 224, generator=g).cuda(), torch.randn(2, 3, 224, 224, generator=g).cuda()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert ops.vit_prologue_ok(xa, v)
        x, n1 = ops.vit_prologue(xa, xb, v, v.blocks[0].norm1)
        xx = torch.cat([xa, xb], 0)
        patches = xx.reshape(5, 3, 16, 14, 16, 14).permute(0, 2, 4, 1, 3, 5).reshape(5, 256, 588)
        y = ops.patch_embed(patches, v.patch_embed.proj) + v.pos_embed
        y = torch.cat([v.cls_token.expand(5, -1, -1), v.reg_token.expand(5, -1, -1), y], 1)
    assert x.dtype == torch.float32 and torch.equal(x, y)  # same GEMM, same sums
    ref = F.layer_norm(y, (768,), v.blocks[0].norm1.weight, v.blocks[0].norm1.bias, 1e-6)
    assert n1.dtype == torch.bfloat16 and ((n1.float() - ref).abs() <= ref.abs() * 2.0 ** -8 + 1e-3).all()
    # one batch alone: the same kernels with nb = 0
    with torch.autocast("cuda", dtype=torch.bfloat16):
        x1, _ = ops.vit_prologue(xb, None, v, v.blocks[0].norm1)
    assert (x1 - x[3:]).abs().max().item() < 0.05  # (another tile walk of the GEMM: bf16 rounding flips only)


@torch.no_grad()
def test_vit_forward_pair_equals_concatenated_batch():
    from unopose_amd import ops

    v = _vit()
    g = torch.Generator().manual_seed(2)
    xa, xb = torch.randn(2, 3, 224, 224, generator=g).cuda(), torch.randn(2, 3, 224, 224, generator=g).cuda()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        pair = v((xa, xb), taps_side_by_side=True)
        cat = v(torch.cat([xa, xb], 0), taps_side_by_side=True)
    assert torch.equal(pair, cat)


@torch.no_grad()
def test_row_dot_normalize_transpose_pad():
    from unopose_amd import ops
    from unopose_amd._lib import call, ptr, stream_ptr

    g = torch.Generator().manual_seed(3)
    f = torch.randn(7, 300, 256, generator=g).cuda()
    lin = torch.nn.Linear(256, 1).cuda()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        for x in (f, f.bfloat16()):
            s = ops.score_head(x, lin)
            ref = (x.float() * lin.weight.detach().bfloat16().float().reshape(-1)).sum(-1, keepdim=True) + lin.bias.detach().float()
            assert s.shape == (7, 300, 1) and s.dtype == torch.bfloat16
            assert ((s.float() - ref).abs() <= ref.abs() * 2.0 ** -8 + 1e-5).all()
    for x in (f, f.bfloat16()):
        n = ops.normalize_rows_bf16(x, 0.1)
        ref = F.normalize(x.float(), p=2, dim=2) / 0.1
        assert n.dtype == torch.bfloat16 and ((n.float() - ref).abs() <= ref.abs() * 2.0 ** -8 + 1e-6).all()
        n32 = ops.normalize_rows_bf16(x, 0.1, as_f32=True)  # the same bf16-rounded values, stored as fp32 (bmm_nt_f32's operand type)
        assert n32.dtype == torch.float32 and torch.equal(n32, n.float())
    nf = ops.normalize_rows_f32(f)  # unrounded (the fp32 forward's F.normalize)
    want = F.normalize(f, p=2, dim=2)
    assert nf.dtype == torch.float32 and (nf - want).abs().max().item() < 2e-7
    assert torch.equal(ops.normalize_rows_bf16(torch.zeros(2, 3, 256, device="cuda"), 0.1), torch.zeros(2, 3, 256, device="cuda", dtype=torch.bfloat16))
    y = torch.randn(5, 197, 768, generator=g).cuda().bfloat16()  # q | k | v side by side, v = the last 256 columns
    vt = torch.full((5, 256, 256), 7.0, device="cuda", dtype=torch.bfloat16)
    import ctypes
    call("unopose_transpose_pad_bf16", ctypes.c_void_p(y.data_ptr() + 512 * 2), y.stride(1), 5, 197, 256, 256, ptr(vt), stream_ptr())
    assert torch.equal(vt[:, :, :197], y[..., 512:].transpose(1, 2)) and (vt[:, :, 197:] == 0).all()
    y32 = torch.randn(5, 197, 512, generator=g).cuda()  # fp32: k | v side by side
    vt32 = torch.full((5, 256, 224), 7.0, device="cuda")
    call("unopose_transpose_pad_f32", ctypes.c_void_p(y32.data_ptr() + 256 * 4), y32.stride(1), 5, 197, 256, 224, ptr(vt32), stream_ptr())
    assert torch.equal(vt32[:, :, :197], y32[..., 256:].transpose(1, 2)) and (vt32[:, :, 197:] == 0).all()


@torch.no_grad()
def test_pe_split_output_and_mlp3_epilogue():
    """Both PE scales written straight into the split-layout operand = the fp32 outputs, split; mlp3 on the fp32-class GEMM with the bf16
    residual add = the reference's  d + PE(p).to(bf16)  at bf16 resolution."""
    from unopose_amd import ops
    from unopose_amd.model import UNOPose, default_model_cfg
    from unopose_amd.synthetic import trained_like_

    torch.manual_seed(0)
    model = trained_like_(UNOPose(default_model_cfg())).cuda().eval()
    PE = model.fine_point_matching.PE
    g = torch.Generator().manual_seed(5)
    pts = (torch.randn(3, 512, 3, generator=g) * 0.2).cuda()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert PE.split_ok(pts)
        ref_groups = PE.groups(pts)  # (3,512,256) fp32
        buf = torch.zeros(4, 512, 512, dtype=torch.bfloat16, device="cuda")
        PE.groups_split(pts, buf, 1)
        assert (buf[0] == 0).all()
        blk = buf[1:].reshape(3, 512, 8, 2, 32).float()
        hi, lo = blk[:, :, :, 0].reshape(3, 512, 256), blk[:, :, :, 1].reshape(3, 512, 256)
        assert torch.equal(hi, ref_groups.bfloat16().float()) and ((hi + lo - ref_groups).abs() <= ref_groups.abs() * 2.0 ** -16).all()
        d = torch.randn(3, 512, 256, generator=g).cuda().bfloat16()
        out = PE.project_add(buf[1:].contiguous(), d)
        ref = d + PE.project(ref_groups).to(torch.bfloat16)
    assert out.dtype == torch.bfloat16 and out.shape == ref.shape
    assert ((out.float() - ref.float()).abs() <= ref.float().abs() * 2.0 ** -7 + 2e-3).all()
    assert (out != ref).float().mean().item() < 0.02  # the same two roundings; a flip where the fp32-class sums differ in the last bits


@torch.no_grad()
def test_bmm_f32_strided_contractions():
    """csrc/bmm_f32.hip vs float64 einsum: the coarse similarity shape (K-contiguous operands), the linear attention's k^T v
    ((pair, head) batches read in place, contraction index strided) and a ragged shape."""
    from unopose_amd import ops

    g = torch.Generator().manual_seed(9)
    a, b = torch.randn(6, 197, 256, generator=g).cuda(), torch.randn(6, 197, 256, generator=g).cuda()
    c = ops.bmm_nt_f32(a, b, 10.0)
    ref = torch.einsum("bik,bjk->bij", a.double(), b.double()) * 10.0
    assert c.shape == (6, 197, 197) and (c.double() - ref).abs().max().item() < 2e-4 * 10
    v, k = torch.randn(3, 196, 256, generator=g).cuda(), torch.rand(3, 196, 256, generator=g).cuda()
    kvt = ops.bmm_nt_f32(v.reshape(3, 196, 4, 64).permute(0, 2, 3, 1), k.reshape(3, 196, 4, 64).permute(0, 2, 3, 1))
    ref = torch.einsum("bjhd,bjhc->bhdc", v.double().reshape(3, 196, 4, 64), k.double().reshape(3, 196, 4, 64))
    assert kvt.shape == (3, 4, 64, 64) and kvt.is_contiguous() and (kvt.double() - ref).abs().max().item() < 1e-4
    x, r = torch.randn(2, 33, 3, generator=g).cuda(), torch.randn(2, 3, 3, generator=g).cuda()
    assert (ops.bmm_nt_f32(x, r.transpose(1, 2)) - x @ r).abs().max().item() < 1e-5
    # the 64 x 64-per-wave form (outputs >= 256 x 256: the fp32 path's fine similarity, ragged 2049 x 2049 there): the same k-ordered fma
    # chain per element, so it equals the 32 x 32 form BIT FOR BIT (reached here through row slices narrower than 256) and float64 to 2e-4
    a, b = torch.randn(2, 321, 256, generator=g).cuda(), torch.randn(2, 449, 256, generator=g).cuda()
    c = ops.bmm_nt_f32(a, b)
    assert c.shape == (2, 321, 449) and (c.double() - torch.einsum("bik,bjk->bij", a.double(), b.double())).abs().max().item() < 2e-4
    parts = torch.cat([ops.bmm_nt_f32(a[:, i0:i0 + 107], b) for i0 in (0, 107, 214)], 1)
    assert torch.equal(c, parts)


def test_gather_rows_kernel_matches_torch_composite():
    """unopose_gather_rows (csrc/glue.hip): plain (B,N,C)-layout gather for fp32 points (12-byte rows), fp32 / bf16 features, int32 and
    int64 indices, and the background-token form (index 0 -> the alternative row, optional prepended row) of transformer.py:655-662."""
    from unopose_amd import ops

    g = torch.Generator().manual_seed(0)
    for dtype, C in ((torch.float32, 3), (torch.float32, 256), (torch.bfloat16, 256), (torch.bfloat16, 6)):
        feats = torch.randn(5, 301, C, generator=g).cuda().to(dtype)
        for idt in (torch.int32, torch.int64):
            idx = torch.randint(0, 301, (5, 77), generator=g).cuda().to(idt)
            want = torch.gather(feats, 1, idx.long().unsqueeze(2).expand(-1, -1, C))
            assert torch.equal(ops.gather_rows(feats, idx), want)
            idx[:, ::5] = 0
            bg = torch.randn(5, 1, C, generator=g).cuda()
            base = torch.where((idx == 0).unsqueeze(-1), bg.to(dtype), torch.gather(feats, 1, (idx.long() - 1).clamp(min=0).unsqueeze(2).expand(-1, -1, C)))
            assert torch.equal(ops.gather_rows(feats, idx, off=1, alt=bg), base)
            assert torch.equal(ops.gather_rows(feats, idx, off=1, alt=bg, prepend=True), torch.cat([bg.to(dtype), base], 1))
            if (C * feats.element_size()) % 4 == 0:  # the alternative rows read in place as row 0 of a (B, 1 + n, C) tensor (no copy)
                holder = torch.randn(5, 9, C, generator=g).cuda().to(dtype)
                holder[:, 0:1] = bg.to(dtype)
                assert torch.equal(ops.gather_rows(feats, idx, off=1, alt=holder[:, 0:1], prepend=True), torch.cat([bg.to(dtype), base], 1))
    # int64 pixel indices as 8-byte rows, written into one half of a stacked tensor (the FPS subset's `choose`, model/unopose.py)
    ch = torch.randint(0, 50000, (5, 301), generator=g).cuda()
    for idt in (torch.int32, torch.int64):
        idx = torch.randint(0, 301, (5, 77), generator=g).cuda().to(idt)
        dst = torch.full((10, 77), -1, dtype=torch.int64, device="cuda")
        got = ops.gather_rows(ch.unsqueeze(-1), idx, out=dst[5:].unsqueeze(-1)).squeeze(-1)
        assert torch.equal(got, torch.gather(ch, 1, idx.long())) and got.data_ptr() == dst[5:].data_ptr() and bool((dst[:5] == -1).all())
    # differentiable mode / tensors that carry gradients keep the autograd-recorded composite
    f = torch.randn(2, 10, 4, device="cuda", requires_grad=True)
    out = ops.gather_rows(f, torch.randint(0, 10, (2, 3), device="cuda"))
    assert out.grad_fn is not None


def _unsplit(xs, C):
    """(rows, 2C) bf16 split layout -> fp32 hi + lo."""
    rows = xs.shape[0]
    blocks = xs.reshape(rows, C // 32, 2, 32).float()
    return (blocks[:, :, 0] + blocks[:, :, 1]).reshape(rows, C)


def test_scale_residual_layernorm_f32_kernel():
    """unopose_scale_residual_layernorm_f32: the residual update is the two-rounding x + (gamma * y) of the op-by-op path bit for bit;
    the LayerNorm comes back in the split layout of csrc/gemm_f32.hip (hi + lo within 2^-16 of the fp32 value)."""
    from unopose_amd import ops

    torch.manual_seed(0)
    x = torch.randn(3, 101, 768, device="cuda")
    y = torch.randn(3, 101, 768, device="cuda")
    gamma = torch.rand(768, device="cuda")
    norm = torch.nn.LayerNorm(768, eps=1e-6).cuda()
    with torch.no_grad():
        norm.weight.uniform_(0.5, 1.5)
        norm.bias.uniform_(-0.5, 0.5)
    want_x = x + y * gamma
    want_n = norm(want_x)
    xs = x.clone()
    ns = ops.scale_residual_layernorm_f32_(xs, y, gamma, norm)
    assert torch.equal(xs, want_x)
    got_n = _unsplit(ns, 768).reshape(3, 101, 768)
    assert float((got_n - want_n).abs().max()) < 2e-5 * float(want_n.abs().max())
    # LayerNorm only (y = None) and residual only (norm = None)
    xs2 = x.clone()
    n_only = _unsplit(ops.scale_residual_layernorm_f32_(xs2, None, None, norm), 768).reshape(3, 101, 768)
    assert torch.equal(xs2, x) and float((n_only - norm(x)).abs().max()) < 2e-5 * float(norm(x).abs().max())
    xs3 = x.clone()
    assert ops.scale_residual_layernorm_f32_(xs3, y, gamma, None) is xs3 and torch.equal(xs3, want_x)
    # into column block 2 of a four-block-wide split matrix (the tap LayerNorms of the fp32 ViT side by side): that block equals the dense result
    # bit for bit, the other blocks stay untouched
    wide = torch.full((303, 2 * 4 * 768), 7.0, dtype=torch.bfloat16, device="cuda")
    assert ops.scale_residual_layernorm_f32_(x.clone(), None, None, norm, wide=wide, block=2) is wide
    dense = ops.scale_residual_layernorm_f32_(x.clone(), None, None, norm)
    assert torch.equal(wide[:, 2 * 1536:3 * 1536], dense) and bool((wide[:, :2 * 1536] == 7.0).all()) and bool((wide[:, 3 * 1536:] == 7.0).all())
    # the layout is the split kernel's: the hi halves are the bf16 roundings of the values the pair encodes (lo < half an ulp of hi)
    hi = ns.reshape(-1, 24, 2, 32)[:, :, 0].float()
    assert float((hi - _unsplit(ns, 768).reshape(-1, 24, 32)).abs().max()) <= 2.0 ** -8 * float(hi.abs().max())


@torch.no_grad()
def test_vit_fp32_fused_blocks_match_the_op_by_op_path(monkeypatch):
    """ViT without autocast: the fused fp32-class blocks (split-layout LayerNorms feeding the GEMMs) against the same model run block
    by block through torch LayerNorm / multiply / add (both on csrc/gemm_f32.hip): taps agree to fp32 rounding."""
    from unopose_amd import ops
    from unopose_amd.model.modules import ViT

    torch.manual_seed(1)
    vit = ViT(img_size=224).cuda().eval()
    for p in vit.parameters():
        if p.dim() == 1 and p.numel() == 768 and float(p.abs().max()) < 1e-3:
            p.fill_(0.3)  # LayerScale gammas: make the branches matter
    x = torch.randn(3, 3, 224, 224, device="cuda")
    fused = vit(x)
    monkeypatch.setattr(ops, "vit_f32_fused_ok", lambda *a, **k: False)
    plain = vit(x)
    assert len(fused) == len(plain) == 4
    for a, b in zip(fused, plain):
        assert float((a - b).abs().max()) < 5e-5 * float(b.abs().max())


def test_vit_attention_f32_split_output():
    """The fp32 ViT attention writing the split layout == the split kernel applied to its fp32 output, bit for bit."""
    from unopose_amd import ops

    torch.manual_seed(2)
    qkv = torch.randn(3, 261, 2304, device="cuda")
    want = ops.split_f32(ops.vit_attention(qkv, 12).reshape(-1, 768))
    assert torch.equal(ops.vit_attention_f32_split(qkv, 12), want)


@torch.no_grad()
def test_small_fp32_glue_kernels_vs_torch():
    """Round 5: the radius / scale / sigmoid / rigid transform / token sum / pose score kernels of csrc/glue.hip against the torch ops
    they replace (the last reductions of the eval forward that ran as at::native kernels)."""
    from unopose_amd import ops

    g = torch.Generator().manual_seed(5)
    pts = (torch.randn(6, 5000, 3, generator=g) * 0.07 + torch.tensor([0.02, -0.03, 0.8])).cuda()
    ref_r = torch.norm(pts.double() - pts.double().mean(1, keepdim=True), dim=2).max(1)[0]
    r = ops.cloud_radius(pts)
    assert (r.double() - ref_r).abs().max().item() < 2e-7 * ref_r.max().item()
    assert torch.equal(ops.scale_by_radius(pts, r), pts / (r.reshape(-1, 1, 1) + 1e-6))     # IEEE division: bit-identical
    t = torch.randn(6, 3, generator=g).cuda()
    assert torch.equal(ops.scale_by_radius(t, r, multiply=True), t * (r.reshape(-1, 1) + 1e-6))
    # overlap scores: both dtypes, background tokens dropped
    n1, n2 = 196, 150
    sc = (3 * torch.randn(4, n1 + n2 + 2, 1, generator=g)).cuda()
    want = torch.clamp(torch.sigmoid(torch.cat((sc[:, 1:n1 + 1], sc[:, n1 + 2:]), 1).squeeze(-1)), 0, 1)
    assert (ops.overlap_scores(sc, n1) - want).abs().max().item() < 2e-7
    scb = sc.bfloat16()
    wantb = torch.clamp(torch.sigmoid(torch.cat((scb[:, 1:n1 + 1], scb[:, n1 + 2:]), 1).squeeze(-1).float()), 0, 1)
    assert (ops.overlap_scores(scb, n1) - wantb).abs().max().item() < 2e-7
    # (p - t) @ R under autocast: bf16-rounded operands, fp32 accumulation, bf16 result
    p = torch.randn(3, 777, 3, generator=g).cuda()
    tt = torch.randn(3, 3, generator=g).cuda()
    R = torch.linalg.qr(torch.randn(3, 3, 3, generator=g))[0].cuda()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = ops.rigid_rows(p, tt, R)
    bf = torch.bfloat16
    xb, Rb = (p - tt.unsqueeze(1)).to(bf).float(), R.to(bf).float()
    ref = (xb[..., 0:1] * Rb[:, None, 0, :] + xb[..., 1:2] * Rb[:, None, 1, :] + xb[..., 2:3] * Rb[:, None, 2, :]).to(bf)
    assert y.dtype == bf and torch.equal(y, ref)
    # pose score
    dis, w = torch.rand(5, 2048, generator=g).cuda() * 0.3, torch.rand(5, 2048, generator=g).cuda()
    want = ((dis < 0.15).float() * w).double().sum(1) / (w.double().sum(1) + 1e-8) * w.double().mean(1)
    assert (ops.pose_score(dis, w, 0.15).double() - want).abs().max().item() < 1e-6
    # token sum (through the C ABI: ops uses it inside the linear attention)
    import ctypes

    from unopose_amd._lib import call, ptr, stream_ptr
    x = torch.randn(3, 2049, 256, generator=g).bfloat16().cuda()
    out = torch.empty(3, 256, device="cuda")
    call("unopose_token_sum_bf16", ptr(x), 3, 2049, 256, ptr(out), stream_ptr())
    assert (out.double() - x.double().sum(1)).abs().max().item() < 2e-3


@pytest.mark.parametrize("B,n,m", [(2, 196, 196), (3, 1000, 1300), (2, 4096, 4096), (1, 5, 700)])
def test_nearest_partner_labels_vs_distance_matrix(B, n, m):
    """csrc/glue.hip::nearest_partner_kernel (training labels without the (B, n, m) matrix) vs the reference's formulation on the
    matrix (loss_utils.py:150-176, as unopose_amd/losses.py evaluates it on the CPU): minimum distances to 1e-6, the any-close flags and
    the arg mins identical except where the two smallest distances are within rounding of each other (the matrix product's summation
    order is the library's)."""
    from unopose_amd import ops
    from unopose_amd.losses import _pairwise_sq_dist

    g = torch.Generator().manual_seed(B * 7 + n)
    a = (torch.rand(B, n, 3, generator=g) - 0.5).cuda()
    b = (a[:, torch.randint(0, n, (m,), generator=g)] + 0.02 * torch.randn(B, m, 3, generator=g).cuda()).contiguous()
    thr = 0.03
    dist = torch.sqrt(_pairwise_sq_dist(a, b))  # fp32, the formulation the kernel restates (the expansion cancels: only its own rounding is comparable)
    for over_b, dim in ((True, 2), (False, 1)):
        d, idx, anyc = ops.nearest_partner(a, b, thr, over_b=over_b)
        dr, ir = dist.min(dim)
        # squared distances agree to the rounding of one product sum (the library's K = 3 dot product may associate differently)
        assert float((d.double() ** 2 - dr.double() ** 2).abs().max()) < 3e-7
        other = dist.gather(dim, idx.unsqueeze(dim)).squeeze(dim)  # the matrix's distance of the partner the kernel chose
        assert float((other.double() ** 2 - dr.double() ** 2).max()) < 3e-7  # a different index only among (near-)ties
        assert (idx == ir).float().mean().item() > 0.99
        close = (dist <= thr).any(dim)
        border = ((dr.double() ** 2 - thr ** 2).abs() < 1e-6)
        assert torch.equal(anyc[~border], close[~border])


@torch.no_grad()
def test_topk_smallest_and_coarse_pick_kernels():
    """unopose_topk_smallest (rank by counting) vs torch.topk(largest=False): same values in ascending order, indices equal where the values are
    distinct, ties in index order, NaN last; unopose_coarse_pick vs max + gathers (first maximum); coarse_pose with and without them."""
    import os
    import numpy as np
    from unopose_amd import ops
    from unopose_amd._lib import call, ptr, stream_ptr

    def load(name):
        z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
        return {k: torch.from_numpy(z[k]).cuda() if z[k].ndim else z[k].item() for k in z.files}

    g = torch.Generator().manual_seed(5)
    for B, n, k in ((3, 6000, 300), (2, 257, 257), (1, 5, 1), (4, 1000, 37), (2, 9000, 2048), (2, 6000, 2500), (33, 6000, 300)):  # (k > 2048: the counting kernel)
        x = torch.randn(B, n, generator=g).cuda()
        idx = torch.full((B, k), -1, dtype=torch.int64, device="cuda")
        call("unopose_topk_smallest", ptr(x), B, n, k, ptr(idx), stream_ptr())
        want = torch.topk(x, k, dim=1, largest=False)
        assert torch.equal(torch.gather(x, 1, idx), want[0]) and torch.equal(idx, want[1])  # (distinct values: one answer)
    xt = torch.randint(0, 7, (3, 4000), generator=g).float().cuda()  # seven distinct values: ties everywhere, the stable order is the only answer
    for k in (1, 300, 2048, 3000):
        idx = torch.empty(3, k, dtype=torch.int64, device="cuda")
        call("unopose_topk_smallest", ptr(xt), 3, 4000, k, ptr(idx), stream_ptr())
        want = torch.sort(xt, dim=1, stable=True)[1][:, :k]
        assert torch.equal(idx, want), k
    x = torch.tensor([[3.0, 1.0, float("nan"), 1.0, -0.0, 0.0, float("inf"), 1.0, -5.0, float("-inf")]]).cuda()
    idx = torch.empty(1, 10, dtype=torch.int64, device="cuda")
    call("unopose_topk_smallest", ptr(x), 1, 10, 10, ptr(idx), stream_ptr())
    assert idx.tolist() == [[9, 8, 4, 5, 1, 3, 7, 0, 6, 2]]  # ties by index (-0.0 sorts below 0.0: distinct bit patterns), +inf before NaN
    B, ncand, nprop = 5, 300, 6000
    sc = torch.randn(B, ncand, generator=g).cuda()
    sc[1, 7] = sc[1, 200] = sc[1].max() + 1  # a tie: the first wins
    sc[2, 50] = float("nan")                 # a NaN wins (torch.max)
    top = torch.stack([torch.randperm(nprop, generator=g)[:ncand] for _ in range(B)]).cuda()
    rs, ts = torch.randn(B, nprop, 3, 3, generator=g).cuda(), torch.randn(B, nprop, 3, generator=g).cuda()
    R, t, best = torch.empty(B, 3, 3, device="cuda"), torch.empty(B, 3, device="cuda"), torch.empty(B, device="cuda")
    call("unopose_coarse_pick", ptr(sc), ptr(top), B, ncand, ptr(rs), ptr(ts), nprop, ptr(R), ptr(t), ptr(best), stream_ptr())
    ps, bi = sc.max(1)
    assert bi[1].item() == 7 and bi[2].item() == 50
    hyp = torch.gather(top, 1, bi.unsqueeze(1)).squeeze(1)
    ar = torch.arange(B, device="cuda")
    assert torch.equal(R, rs[ar, hyp]) and torch.equal(t, ts[ar, hyp]) and torch.equal(best.nan_to_num(7.0), ps.nan_to_num(7.0))
    z = load("coarse_rt")
    a = ops.coarse_pose(z["atten"], z["score"], z["p1"], z["p2"], z["rand"], 6000, 300)
    ops.USE_OWN_TOPK = False
    try:
        b = ops.coarse_pose(z["atten"], z["score"], z["p1"], z["p2"], z["rand"], 6000, 300)
    finally:
        ops.USE_OWN_TOPK = True
    assert all(torch.equal(u, v) for u, v in zip(a, b))


@torch.no_grad()
def test_copy_rows_and_overlap_scores_halves():
    """unopose_copy_rows through ops.set_first_rows_ (one row into row 0 of every batch, nothing else touched) and the two-halves form of
    unopose_overlap_scores (score head run over both clouds as one batch of 2B) against the concatenated form."""
    from unopose_amd import ops

    g = torch.Generator().manual_seed(9)
    for dtype in (torch.bfloat16, torch.float32):
        x = torch.randn(6, 197, 256, generator=g).cuda().to(dtype)
        keep = x.clone()
        row = torch.randn(256, generator=g).cuda().to(dtype)
        y = ops.set_first_rows_(x, row)
        assert y.data_ptr() == x.data_ptr() and torch.equal(x[:, 0], row.expand(6, -1)) and torch.equal(x[:, 1:], keep[:, 1:])
        sc = torch.randn(10, 197, 1, generator=g).cuda().to(dtype)
        want = ops.overlap_scores(torch.cat((sc[:5], sc[5:]), dim=1), 196)
        assert torch.equal(ops.overlap_scores(sc, 196, halves=True), want) and want.shape == (5, 392)


@torch.no_grad()
def test_coarse_slot_path_equals_concatenated_path():
    """UNOPose.forward with the coarse matcher's stacked input (one gather target with a slot row, one in_proj GEMM, background token copied
    in, scores read from the two halves) against round 5's concatenations: same poses, bit for bit (row-wise kernels, same rows)."""
    import unopose_amd.model.unopose as mu
    from unopose_amd.model import UNOPose, default_model_cfg
    from unopose_amd.synthetic import make_batch, trained_like_

    m = trained_like_(UNOPose(default_model_cfg())).cuda().eval()
    ep, _, _ = make_batch(3, S=224, device="cuda")
    ep["coarse_rand"] = torch.rand(3, 18000, generator=torch.Generator().manual_seed(1)).cuda()
    outs = []
    for flag in (True, False):
        mu.COARSE_SLOT = flag
        try:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                o = m(dict(ep))
        finally:
            mu.COARSE_SLOT = True
        outs.append((o["init_R"].clone(), o["init_t"].clone(), o["pred_R"].clone(), o["pred_t"].clone(), o["pred_pose_score"].clone()))
    assert all(torch.equal(a, b) for a, b in zip(*outs))
