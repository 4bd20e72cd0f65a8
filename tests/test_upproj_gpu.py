"""csrc/upproj.hip + the gathered form of csrc/gemm.hip: pixel features computed only where the chosen pixels look, against
(a) the dense path (ops.linear + ops.bilinear_sample_native: same bf16 operands, so equal up to fp32 summation order
inside the GEMM) and (b) the oracle's chosen_pixel_feats on an fp32 torch reference of the op
(oracle/unopose_ref.py:445-460 = oneref_feature_extraction.py:200-236 + model_utils.py:215-227)."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def make(B2, side, Np, K, seed, npre=5, clustered=False):
    g = torch.Generator().manual_seed(seed)
    S = side * 14
    acts = torch.randn(B2, npre + side * side, K, generator=g).to(BF)
    lin = torch.nn.Linear(K, 4096)
    with torch.no_grad():
        lin.weight.copy_(torch.randn(4096, K, generator=g) / K ** 0.5)
        lin.bias.copy_(torch.randn(4096, generator=g))
    if clustered:  # every pixel in one corner: a handful of cells, most sub-position groups tiny
        yy = torch.randint(0, 9, (B2, Np), generator=g)
        xx = torch.randint(0, 9, (B2, Np), generator=g)
        choose = yy * S + xx
        choose[:, 0] = S * S - 1  # and the far corner (clamped taps)
    else:
        choose = torch.randint(0, S * S, (B2, Np), generator=g)
    return acts, lin, choose, S


def plan_invariants(plan, B2, side, npre):
    ti = plan["tile_info"].cpu()
    ntiles = int(ti[0])
    starts = ti[1:18]
    assert int(starts[0]) == 0 and int(starts[16]) == ntiles and (starts[1:] >= starts[:-1]).all()
    rl = plan["row_list"].cpu()[: ntiles * 256]
    ts = npre + side * side
    valid = rl >= 0
    assert (rl[valid] < B2 * ts).all() and ((rl[valid] % ts) >= npre).all()
    assert (plan["row_list"].cpu()[ntiles * 256:] == -1).all()
    # every needed cell exactly once: (row, group) pairs are unique
    grp = torch.bucketize(torch.arange(ntiles * 256) // 256, starts[1:17].contiguous(), right=True)
    key = rl[valid].long() * 16 + grp[valid]
    assert key.unique().numel() == key.numel()
    return ntiles, int(valid.sum())


@torch.no_grad()
@pytest.mark.parametrize("B2,side,Np,K,clustered", [(4, 16, 2048, 3072, False), (3, 5, 700, 128, False), (2, 16, 512, 256, True),
                                                    (64, 37, 2048, 3072, False)])
def test_sparse_pixel_features_vs_dense(B2, side, Np, K, clustered):
    from unopose_amd import ops

    acts, lin, choose, S = make(B2, side, Np, K, 11 + side, clustered=clustered)
    acts, lin, choose = acts.cuda(), lin.cuda(), choose.cuda()
    with torch.autocast("cuda", dtype=BF):
        assert ops.sparse_upproj_ok(acts)
        plan = ops.upproj_plan(choose, S, S, side, 5, 5 + side * side)
        ntiles, nrows = plan_invariants(plan, B2, side, 5)
        out = ops.sparse_pixel_features(acts, lin, plan)
        ops.PIXEL_FEATS_BF16 = False  # the fp32 result; the default is exactly its bf16 rounding (what the consuming autocast Linear would cast it to)
        try:
            out32 = ops.sparse_pixel_features(acts, lin, plan)
        finally:
            ops.PIXEL_FEATS_BF16 = True
        assert out.dtype == BF and out32.dtype == torch.float32 and torch.equal(out, out32.to(BF))
        out = out32
        z = ops.linear(acts, lin).reshape(B2, 5 + side * side, 4, 4, 256)
        ref = ops.bilinear_sample_native(z, choose, S, S, tok_offset=5)
    e = (out - ref).abs().max().item()
    scale = ref.abs().max().item()
    print(f"sparse vs dense {B2}x{side}^2x{Np} K={K}: {nrows} of {B2 * 16 * side * side} cells in {ntiles} tiles; max |diff| {e:.2e} (|ref| max {scale:.1f})")
    # both round the cell values to bf16 after an fp32 accumulation in a different order: a rounding flip is one bf16 ulp
    assert e <= 2 ** -7 * scale
    assert (out - ref).abs().mean().item() < 1e-3 * scale
    if B2 <= 4:
        w, b = lin.weight.to(BF).float(), lin.bias.float()
        zf = (acts.float()[:, 5:] @ w.t() + b).reshape(B2, side, side, 4, 4, 256)
        low = zf.permute(0, 5, 1, 3, 2, 4).reshape(B2, 256, 4 * side, 4 * side)
        up = F.interpolate(low, (S, S), mode="bilinear", align_corners=False).flatten(2)
        exact = torch.gather(up, 2, choose.unsqueeze(1).expand(-1, 256, -1)).transpose(1, 2)
        ee = (out - exact).abs().max().item()
        print(f"  vs fp32 torch reference of the op: max err {ee:.2e}")
        assert ee <= 2 ** -7 * scale


@torch.no_grad()
def test_model_forward_sparse_equals_dense_upprojection():
    """UNOPose.forward with and without the sparse up-projection: same poses (the cell values differ by bf16 rounding flips)."""
    from unopose_amd import ops
    from unopose_amd.model import UNOPose, default_model_cfg
    from unopose_amd.synthetic import make_batch, trained_like_

    m = UNOPose(default_model_cfg())
    trained_like_(m)
    m = m.cuda().eval()
    ep, Rg, tg = make_batch(2, S=224, device="cuda")
    ep["coarse_rand"] = torch.rand(2, 18000, generator=torch.Generator().manual_seed(1)).cuda()
    outs = []
    for flag in (True, False):
        ops.USE_SPARSE_UPPROJ = flag
        try:
            with torch.autocast("cuda", dtype=BF):
                o = m(dict(ep))
        finally:
            ops.USE_SPARSE_UPPROJ = True
        outs.append((o["pred_R"].float().cpu(), o["pred_t"].float().cpu()))
    dR = (outs[0][0] - outs[1][0]).abs().max().item()
    dt = (outs[0][1] - outs[1][1]).abs().max().item()
    print(f"sparse vs dense forward: dR {dR:.2e} dt {dt:.2e}; vs gt {(outs[0][0] - Rg.cpu()).abs().max().item():.2e}")
    assert dR < 2e-3 and dt < 2e-3
    assert (outs[0][0] - Rg.cpu()).abs().max().item() < 0.05


def test_upproj_plan_rejects_small_capacity():
    from unopose_amd._lib import call, ptr, stream_ptr
    x = torch.zeros(1 << 16, dtype=torch.int32, device="cuda")
    ch = torch.zeros(1, 8, dtype=torch.int64, device="cuda")
    with pytest.raises(RuntimeError, match="cap_rows"):
        call("unopose_upproj_plan", ptr(ch), 1, 8, 28, 28, 2, 0, 4, 256, ptr(x), ptr(x), ptr(x), ptr(x), stream_ptr())
