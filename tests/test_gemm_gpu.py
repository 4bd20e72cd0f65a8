"""GPU: the hand-written bf16 linear GEMM (csrc/gemm.hip, C ABI unopose_linear_bf16) against a plain PyTorch fp32
reference of the same op: C = act(A W^T + b).  Tolerance: the bf16 output's own resolution (half an ulp of the
result = 2^-9 relative) plus the fp32 accumulation-order noise."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@torch.no_grad()
@pytest.mark.parametrize("M,K,N,gelu", [(1, 64, 256, False), (255, 128, 256, True), (256, 64, 512, False), (4173, 768, 768, True),
                                        (8192, 3072, 768, False), (5000, 768, 2304, True), (64 * 261, 768, 3072, True),
                                        # >= 256 tiles of 256 x 256: the persistent kernel (csrc/gemm.hip), below: 128 x 128 tiles (gemm_small.hip).
                                        # One / two / three (odd) / many K-tiles, ragged last row panel, more tiles than CUs (stream continues)
                                        (70000, 64, 256, False), (65536 + 37, 128, 256, True), (33000, 192, 512, False), (64 * 1374, 768, 768, False),
                                        (40000, 3072, 768, True),
                                        # W larger than an XCD's L2 and row panels a multiple of 8: the column-blocked tile walk (round 4)
                                        (8192, 768, 3072, True), (16384 + 5, 1024, 2304, False)])
def test_linear_bf16_vs_fp32_reference(M, K, N, gelu):
    from unopose_amd import ops

    g = torch.Generator().manual_seed(M + K + N)
    a = torch.randn(M, K, generator=g).bfloat16().cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16().cuda()
    b = torch.randn(N, generator=g).cuda()
    ref = a.float() @ w.float().t() + b
    if gelu:
        ref = F.gelu(ref)  # exact (erf) GELU, timm Mlp's act_layer
    out = ops.linear_bf16_hip(a, w, b, gelu)
    assert out.dtype == torch.bfloat16 and out.shape == (M, N)
    err = (out.float() - ref).abs()
    tol = ref.abs() * 2.0 ** -8 + 1e-3
    assert (err <= tol).all(), (err.max().item(), int((err > tol).sum()))
    # the fused GELU is erf-class (a 2.5e-5 fit of the erf form), not the tanh approximation: its error before rounding is far
    # below the 4.7e-4 gap between the two (checked where bf16 resolves it: |y| < 0.06 -> ulp < 2.5e-4)
    if gelu:
        small = ref.abs() < 0.06
        assert (err[small] < 2.5e-4).all()


@torch.no_grad()
def test_linear_bf16_rejects_unsupported_shapes():
    from unopose_amd import ops

    a = torch.zeros(8, 100, dtype=torch.bfloat16, device="cuda")
    w = torch.zeros(256, 100, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(RuntimeError):
        ops.linear_bf16_hip(a, w, torch.zeros(256, device="cuda"))  # K % 64 != 0


@torch.no_grad()
def test_vit_mlp_fused_gelu_path_is_taken_and_matches():
    """ops.linear(gelu=True) under autocast at ViT size runs the fused kernel; result = F.gelu(linear) at bf16 resolution."""
    from unopose_amd import ops

    g = torch.Generator().manual_seed(0)
    lin = torch.nn.Linear(768, 3072).cuda()
    x = torch.randn(8 * 1374, 768, generator=g).cuda()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = ops.linear(x, lin, gelu=True)
    ref = F.gelu(x.bfloat16().float() @ lin.weight.bfloat16().float().t() + lin.bias.float())
    err = (y.float() - ref).abs()
    assert y.dtype == torch.bfloat16 and (err <= ref.abs() * 2.0 ** -8 + 1e-3).all()


@torch.no_grad()
@pytest.mark.parametrize("M,K", [(197 * 64, 256), (131136, 512), (300, 256), (1, 64)])
def test_linear_add_layernorm_epilogue(M, K):
    """LayerNorm(lin(h) + x) in the GEMM epilogue (256-wide) vs an fp32 reference of the op and vs the two-launch form."""
    from unopose_amd import ops

    g = torch.Generator().manual_seed(M + K)
    lin = torch.nn.Linear(K, 256).cuda()
    ln = torch.nn.LayerNorm(256).cuda()
    with torch.no_grad():
        ln.weight.copy_(torch.rand(256, generator=g) + 0.5)
        ln.bias.copy_(torch.randn(256, generator=g))
    h = (torch.randn(M, K, generator=g) * 2).bfloat16().cuda()
    x = (torch.randn(M, 256, generator=g) * 3).bfloat16().cuda()
    ref = F.layer_norm(h.float() @ lin.weight.bfloat16().float().t() + lin.bias.float() + x.float(), (256,), ln.weight, ln.bias, ln.eps)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = ops.linear_add_layernorm(h, lin, x, ln)
        ops.USE_FUSED_LINEAR_LN = False
        try:
            two = ops.linear_add_layernorm(h, lin, x, ln)
        finally:
            ops.USE_FUSED_LINEAR_LN = True
    assert out.dtype == torch.bfloat16 and out.shape == (M, 256)
    e = (out.float() - ref).abs()
    assert (e <= ref.abs() * 2.0 ** -8 + 2e-3).all(), e.max().item()
    # the two-launch form rounds lin(h) to bf16 before the add: it is the less accurate of the two
    assert (two.float() - ref).abs().max().item() >= e.max().item() * 0.5
    assert (out.float() - two.float()).abs().max().item() < 0.1


# ---- fp32-class GEMM (csrc/gemm_f32.hip): hi / lo-split bf16 operands, three MFMAs per product ---------------------------
@torch.no_grad()
def test_split_layout_roundtrip():
    """unopose_split_bf16x2: per row and 32-k block one 128-byte line [hi | lo]; hi + lo = x to 2^-16 relative."""
    from unopose_amd import ops

    g = torch.Generator().manual_seed(3)
    x = (torch.randn(37, 96, generator=g) * torch.logspace(-3, 3, 96)).cuda()
    s = ops.split_f32(x)
    assert s.shape == (37, 192) and s.dtype == torch.bfloat16
    blk = s.reshape(37, 3, 2, 32).float()
    hi, lo = blk[:, :, 0].reshape(37, 96), blk[:, :, 1].reshape(37, 96)
    assert torch.equal(hi, x.bfloat16().float())
    assert ((hi + lo - x).abs() <= x.abs() * 2.0 ** -16).all()


@torch.no_grad()
@pytest.mark.parametrize("M,K,N,epi", [(1, 32, 256, 0), (300, 256, 256, 2), (4173, 768, 768, 1), (5000, 3072, 768, 0),
                                       (2049 * 3, 256, 512, 2), (64 * 261, 768, 2304, 0), (12608, 256, 512, 1), (6304, 512, 256, 0),
                                       (127, 64, 1792, 2), (131136, 256, 256, 1),
                                       # split W larger than an XCD's L2, row panels a multiple of 8: the column-blocked tile walk
                                       (8192, 768, 2304, 0), (16384, 768, 3072, 1)])
def test_linear_f32x3_vs_fp64_reference(M, K, N, epi):
    """C = act(A W^T + b) on fp32 data: error budget 3 x 2^-17 of sum |a| |w| (the split's representation error and the
    dropped lo x lo term) -- fp32-class, 400 x tighter than a bf16 GEMM.  Shapes with fewer 256 x 256 tiles than CUs (the matcher's
    197-token layers among them) run on gemm_small.hip's 128 x 128 fp32-class kernel, the others on gemm_kernel.h's."""
    from unopose_amd import ops

    g = torch.Generator().manual_seed(M + K + N)
    a = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    out, outs = ops.linear_f32x3(ops.split_f32(a), ops.split_f32(w), b, M, N, K, gelu=epi == 1, relu=epi == 2, out="both")
    rows = slice(0, M) if M <= 4200 else torch.randperm(M, generator=g)[:2048].cuda()
    ref = a[rows].double() @ w.double().t() + b.double()
    mag = a[rows].abs().double() @ w.abs().double().t() + b.abs().double()
    ref = F.gelu(ref) if epi == 1 else (F.relu(ref) if epi == 2 else ref)
    err = (out[rows].double() - ref).abs()
    assert (err <= mag * 3 * 2.0 ** -17 + 1e-6).all(), (err / mag).max().item()
    # the split-layout output is the same numbers, split
    blk = outs[rows].reshape(ref.shape[0], N // 32, 2, 32).float()
    rec = (blk[:, :, 0] + blk[:, :, 1]).reshape(ref.shape[0], N)
    assert ((rec - out[rows]).abs() <= out[rows].abs() * 2.0 ** -16 + 1e-30).all()


@torch.no_grad()
def test_fp32_linears_take_the_f32x3_path_and_match_torch():
    """ops.linear / ops.mlp outside autocast: csrc/gemm_f32.hip; result = the fp32 torch composite at 1e-5 of the operand scale."""
    from unopose_amd import ops

    g = torch.Generator().manual_seed(1)
    fc1, fc2 = torch.nn.Linear(768, 3072).cuda(), torch.nn.Linear(3072, 768).cuda()
    x = torch.randn(2, 1374, 768, generator=g).cuda()
    y = ops.mlp(x, fc1, fc2)
    ref = F.gelu(x.double() @ fc1.weight.double().t() + fc1.bias.double()) @ fc2.weight.double().t() + fc2.bias.double()
    assert y.dtype == torch.float32 and (y.double() - ref).abs().max().item() < 2e-5
    ops.USE_F32X3 = False
    try:
        lib = ops.mlp(x, fc1, fc2)
    finally:
        ops.USE_F32X3 = True
    assert (lib.double() - ref).abs().max().item() < 2e-5 and (y - lib).abs().max().item() < 2e-5
    lin = torch.nn.Linear(256, 512).cuda()
    h = torch.randn(3, 197, 256, generator=g).cuda()
    z = ops.linear(h, lin, relu=True)
    mag = h.abs().double() @ lin.weight.abs().double().t() + lin.bias.abs().double()
    assert ((z.double() - F.relu(h.double() @ lin.weight.double().t() + lin.bias.double())).abs() <= mag * 3 * 2.0 ** -17).all()


def test_strided_linear_matches_the_dense_path():
    """unopose_linear_bf16_ld (row strides for A, W and C): bit-identical to the dense kernel call (same tiles, same K order -- only
    the addresses change); pad columns of A / W are never read, those of C never written."""
    from unopose_amd import ops
    from unopose_amd._lib import call, ptr, stream_ptr

    torch.manual_seed(0)
    M, K, N = 1500, 1024, 512
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda")
    want = ops.linear_bf16_hip(a, w, b, gelu=True)
    ap = torch.full((M, K + 64), 7.0, device="cuda").bfloat16()
    ap[:, :K] = a
    wp = torch.full((N, K + 128), -3.0, device="cuda").bfloat16()
    wp[:, :K] = w
    cp = torch.zeros(M, N + 8, device="cuda", dtype=torch.bfloat16)
    call("unopose_linear_bf16_ld", ptr(ap), K + 64, ptr(wp), K + 128, ptr(b), ptr(cp), N + 8, M, N, K, 1, stream_ptr())
    assert torch.equal(cp[:, :N], want) and float(cp[:, N:].abs().max()) == 0.0  # pad columns of A / W never read, of C never written


@torch.no_grad()
def test_dynamic_tile_tickets_are_placement_independent():
    """The persistent kernel draws its tiles as tickets (csrc/gemm_kernel.h): which workgroup computes which tile depends on when it is
    dispatched.  The result must not: the same GEMM alone, beside a kernel of another stream that holds CUs for milliseconds (the
    5000 -> 2048 FPS: one 512-thread workgroup per cloud) and beside another persistent GEMM of a third stream is bit-identical, launch
    after launch -- including past the 1024-slot ring of ticket counters (every slot re-zeroed by its launch's last workgroup)."""
    from unopose_amd import _lib
    from unopose_amd.pointnet2 import _ext

    g = torch.Generator().manual_seed(7)
    M, K, N = 64 * 1374, 768, 768  # 1032 tiles on 256 CUs: several rounds of tickets per workgroup, ragged last row panel
    a = torch.randn(M, K, generator=g).bfloat16().cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16().cuda()
    b = torch.randn(N, generator=g).cuda()
    a2 = torch.randn(40000, 3072, generator=g).bfloat16().cuda()
    w2 = (torch.randn(768, 3072, generator=g) / 55).bfloat16().cuda()
    tem = torch.rand(32, 5000, 3, generator=g).cuda()
    side, third = torch.cuda.Stream(), torch.cuda.Stream()

    def run(out, x=a, ww=w, n=N, k=K, gelu=False):
        _lib.call("unopose_linear_bf16", _lib.ptr(x), _lib.ptr(ww), _lib.ptr(b), _lib.ptr(out), x.shape[0], n, k, 1 if gelu else 0, _lib.stream_ptr())

    want = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    run(want)
    torch.cuda.synchronize()
    ref = (a[:2048].float() @ w.float().t() + b)
    assert ((want[:2048].float() - ref).abs() <= ref.abs() * 2.0 ** -8 + 1e-3).all()
    outs = [torch.empty_like(want) for _ in range(4)]
    o2 = torch.empty(40000, 768, dtype=torch.bfloat16, device="cuda")
    for rnd in range(3):
        with torch.cuda.stream(side):
            _ext.furthest_point_sampling(tem, 2048)  # 32 CUs held for ~1.9 ms
        with torch.cuda.stream(third):
            run(o2, a2, w2, 768, 3072, True)        # another ticketed launch in flight at the same time (its own slot)
        for o in outs:
            run(o)
        torch.cuda.synchronize()
        for o in outs:
            assert torch.equal(o, want), rnd
    for i in range(1100):  # the ring comes round: slot i % 1024 is reused
        run(outs[i & 3])
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o, want)


@pytest.mark.gpu
@pytest.mark.parametrize("M,Kp,Nc,gelu", [(256 * 5 + 77, 768, 2304, 0), (256 * 3, 3072, 3072, 1), (70000, 768, 768, 0), (300, 768, 1024, 1)])
def test_residual_and_layernorm_fold_vs_fp32_composite(M, Kp, Nc, gelu):
    """Round 6: `unopose_linear_bf16_residual` (proj / fc2 with the LayerScale residual on the fp32 stream in its epilogue) and
    `unopose_linear_bf16_lnfold` (qkv / fc1 with LayerNorm applied algebraically in the epilogue) against the fp32 composite
    x' = x + gamma (a W^T + b);  out = act(LayerNorm(x') W2^T + b2)  (timm Block, oneref_feature_extraction.py:24-42): the residual
    stream to fp32 accuracy of a bf16-operand GEMM, the bf16 rows bit-equal to bf16(x'), the row partial sums exact to fp32 summation,
    the consumer's output at the bf16 level of the unfused chain.  Ragged last tiles, one-tile and many-round grids."""
    import torch.nn as nn
    from unopose_amd import ops

    dev = torch.device("cuda")
    g = torch.Generator(device="cuda").manual_seed(M + Nc)
    C = 768
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731
    a = rn(M, Kp).bfloat16()
    lin_p, lin_c, norm = nn.Linear(Kp, C).to(dev), nn.Linear(C, Nc).to(dev), nn.LayerNorm(C, eps=1e-6).to(dev)
    with torch.no_grad():
        norm.weight.copy_(1 + 0.3 * rn(C))
        norm.bias.copy_(0.2 * rn(C))
    gamma = nn.Parameter(0.05 + 0.45 * torch.rand(C, device=dev, generator=g))
    x0 = (rn(M, C) * 2 + 0.5 * rn(1, C)).contiguous()
    x = x0.clone()
    with torch.no_grad():
        xb, stats = ops.linear_residual_(x, a, lin_p, gamma)
        out = ops.linear_lnfold(xb, stats, lin_c, norm, gelu=bool(gelu))
        xr = x0 + gamma * (a.float() @ lin_p.weight.T + lin_p.bias)
        ref = torch.nn.functional.layer_norm(xr, (C,), norm.weight, norm.bias, 1e-6) @ lin_c.weight.T + lin_c.bias
        if gelu:
            ref = torch.nn.functional.gelu(ref)
    assert (x - xr).abs().max().item() < 2e-2 and (x - xr).abs().mean().item() < 1.5e-3  # bf16 operands, fp32 accumulation and residual
    assert torch.equal(xb, x.bfloat16())
    st = stats[:M].sum(1)
    assert (st[:, 0] - x.sum(1)).abs().max().item() < 1e-2 and ((st[:, 1] - (x * x).sum(1)).abs() / (x * x).sum(1)).max().item() < 1e-5
    e = (out.float() - ref).abs()
    assert e.max().item() < 8e-2 and e.mean().item() < 5e-3, (e.max().item(), e.mean().item())
    # the weight caches follow an in-place edit of ANY tensor they are derived from (LayerScale, LayerNorm bias)
    with torch.no_grad():
        norm.bias.add_(1.0)
        out2 = ops.linear_lnfold(xb, stats, lin_c, norm, gelu=bool(gelu))
        ref2 = torch.nn.functional.layer_norm(xr, (C,), norm.weight, norm.bias, 1e-6) @ lin_c.weight.T + lin_c.bias
        if gelu:
            ref2 = torch.nn.functional.gelu(ref2)
    assert (out2.float() - ref2).abs().mean().item() < 5e-3
