"""GPU parity: HIP `_ext` operators (through the C ABI) vs the CPU oracle.

Integer / index outputs must be BIT-EXACT; float copies must be bit-exact;
atomic-add gradients are compared at fp32 tolerance (summation order differs).
"""
import numpy as np
import pytest
import torch

from helpers import object_cloud

pytestmark = pytest.mark.gpu


def _dev(t):
    return t.cuda().contiguous()


@pytest.mark.parametrize("n,m", [(5000, 2048), (2048, 196), (4096, 512), (1024, 196), (700, 64), (196, 196),
                                 (513, 40), (512, 40), (511, 40), (64, 10), (3, 3), (1, 1), (5, 9), (10000, 64)])
def test_fps_bit_exact(hip_ext, oracle_ext, n, m):
    g = torch.Generator().manual_seed(n * 7 + m)
    x = torch.randn(3, n, 3, generator=g)
    if n >= 64:
        x[0, n // 2: n // 2 + n // 4] = x[0, : n // 4]  # duplicates -> ties
        x[1] = object_cloud(g, n, with_replacement=True)
    ref = oracle_ext.furthest_point_sampling(x.contiguous(), m)
    out = hip_ext.furthest_point_sampling(_dev(x), m)
    assert out.dtype == torch.int32 and out.is_cuda
    assert torch.equal(out.cpu(), ref)


def test_fps_identical_points_and_full_batch(hip_ext, oracle_ext):
    x = torch.ones(2, 600, 3)
    assert torch.equal(hip_ext.furthest_point_sampling(_dev(x), 8).cpu(), oracle_ext.furthest_point_sampling(x, 8))
    # BASELINE config 2 sizes: 32 clouds; property check on all + oracle on 2
    g = torch.Generator().manual_seed(1)
    x = torch.stack([object_cloud(g, 5000) for _ in range(32)])
    out = hip_ext.furthest_point_sampling(_dev(x), 2048).cpu()
    assert (out[:, 0] == 0).all()
    for b in range(32):
        assert len(torch.unique(out[b])) == 2048  # distinct points -> no repeats
    ref = oracle_ext.furthest_point_sampling(x[:2].contiguous(), 2048)
    assert torch.equal(out[:2], ref)


@pytest.mark.parametrize("n,m,r,ns", [(2048, 2048, 0.2, 256), (2048, 2048, 0.1, 64), (500, 77, 0.5, 16),
                                      (100, 100, 0.01, 8), (65, 3, 10.0, 256), (5000, 300, 0.3, 32), (2048, 2048, 5.0, 64)])
def test_ball_query_bit_exact(hip_ext, oracle_ext, n, m, r, ns):
    g = torch.Generator().manual_seed(n + m + ns)
    xyz = torch.rand(2, n, 3, generator=g)
    xyz[1] = xyz[1] * 0.5
    new_xyz = xyz[:, :m].clone()
    if r < 0.05:
        new_xyz = new_xyz + 5.0
    ref = oracle_ext.ball_query(new_xyz.contiguous(), xyz.contiguous(), r, ns)
    out = hip_ext.ball_query(_dev(new_xyz), _dev(xyz), r, ns)
    assert out.dtype == torch.int32
    assert torch.equal(out.cpu(), ref)


def test_ball_query_unit_radius_normalised_clouds(hip_ext, oracle_ext):
    # the shape UNOPose uses: radius-normalised object clouds, r = 0.1 / 0.2, self-query
    g = torch.Generator().manual_seed(5)
    x = torch.stack([object_cloud(g, 2048, with_replacement=(i == 1)) for i in range(4)])
    c = x.mean(1, keepdim=True)
    x = (x / (x - c).norm(dim=2).max(1)[0].reshape(-1, 1, 1)).contiguous()
    for r, ns in ((0.1, 64), (0.2, 256)):
        ref = oracle_ext.ball_query(x, x, r, ns)
        out = hip_ext.ball_query(_dev(x), _dev(x), r, ns)
        assert torch.equal(out.cpu(), ref)


@pytest.mark.parametrize("C,N,M,S", [(3, 2048, 2048, 256), (3, 2048, 2048, 64), (5, 40, 7, 6), (256, 300, 50, 4), (3, 100, 9, 3)])
def test_group_points_bit_exact(hip_ext, oracle_ext, C, N, M, S):
    g = torch.Generator().manual_seed(C + N + M + S)
    pts = torch.randn(2, C, N, generator=g)
    idx = torch.randint(0, N, (2, M, S), generator=g, dtype=torch.int32)
    ref = oracle_ext.group_points(pts, idx)
    out = hip_ext.group_points(_dev(pts), _dev(idx))
    assert torch.equal(out.cpu(), ref)
    go = torch.randn(2, C, M, S, generator=g)
    gref = oracle_ext.group_points_grad(go, idx, N)
    gout = hip_ext.group_points_grad(_dev(go), _dev(idx), N)
    torch.testing.assert_close(gout.cpu(), gref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("C,N,M", [(256, 5000, 2048), (3, 2048, 196), (256, 2049, 196), (1, 7, 3)])
def test_gather_points_bit_exact(hip_ext, oracle_ext, C, N, M):
    g = torch.Generator().manual_seed(C + N + M)
    pts = torch.randn(2, C, N, generator=g)
    idx = torch.randint(0, N, (2, M), generator=g, dtype=torch.int32)
    assert torch.equal(hip_ext.gather_points(_dev(pts), _dev(idx)).cpu(), oracle_ext.gather_points(pts, idx))
    go = torch.randn(2, C, M, generator=g)
    torch.testing.assert_close(hip_ext.gather_points_grad(_dev(go), _dev(idx), N).cpu(),
                               oracle_ext.gather_points_grad(go, idx, N), rtol=1e-4, atol=1e-4)


def test_three_nn_interpolate(hip_ext, oracle_ext):
    g = torch.Generator().manual_seed(4)
    unknown = torch.randn(2, 300, 3, generator=g)
    known = torch.randn(2, 70, 3, generator=g)
    d_ref, i_ref = oracle_ext.three_nn(unknown, known)
    d, i = hip_ext.three_nn(_dev(unknown), _dev(known))
    assert torch.equal(i.cpu(), i_ref) and torch.equal(d.cpu(), d_ref)
    feats = torch.randn(2, 6, 70, generator=g)
    w = torch.rand(2, 300, 3, generator=g)
    torch.testing.assert_close(hip_ext.three_interpolate(_dev(feats), i, _dev(w)).cpu(),
                               oracle_ext.three_interpolate(feats, i_ref, w), rtol=1e-6, atol=1e-6)
    go = torch.randn(2, 6, 300, generator=g)
    torch.testing.assert_close(hip_ext.three_interpolate_grad(_dev(go), i, _dev(w), 70).cpu(),
                               oracle_ext.three_interpolate_grad(go, i_ref, w, 70), rtol=1e-4, atol=1e-4)


def test_full_size_batch_properties(hip_ext, oracle_ext):
    """BASELINE configs[1] sizes (32 clouds x 2048 points, nsample 256): properties that hold at any size.
    ball_query rows are ascending up to the first repeat of the row's first index (the padding rule),
    every listed neighbour is inside the radius, results do not depend on the batch composition;
    group_points equals an index gather; one element is compared with the oracle bit for bit."""
    g = torch.Generator().manual_seed(3)
    x = torch.stack([object_cloud(g, 2048) for _ in range(32)])
    x = (x / x.norm(dim=2).amax(1).reshape(-1, 1, 1)).contiguous()
    xd = _dev(x)
    idx = hip_ext.ball_query(xd, xd, 0.2, 256)
    assert idx.shape == (32, 2048, 256) and idx.dtype == torch.int32
    assert torch.equal(idx[5:6], hip_ext.ball_query(xd[5:6].contiguous(), xd[5:6].contiguous(), 0.2, 256))
    assert torch.equal(idx[7].cpu(), oracle_ext.ball_query(x[7:8], x[7:8], 0.2, 256)[0])
    li = idx.long()
    nb = torch.gather(xd.unsqueeze(1).expand(-1, 2048, -1, -1), 2, li.unsqueeze(-1).expand(-1, -1, -1, 3))
    d2 = ((nb - xd.unsqueeze(2)) ** 2).sum(-1)
    assert (d2 < 0.2 * 0.2 + 1e-7).all()
    pad = li == li[:, :, :1]  # padding repeats the first hit
    inc = li[:, :, 1:] > li[:, :, :-1]
    assert (inc | pad[:, :, 1:]).all()
    xt = xd.transpose(1, 2).contiguous()
    grp = hip_ext.group_points(xt, idx)
    assert torch.equal(grp, nb.permute(0, 3, 1, 2))
    fps = hip_ext.furthest_point_sampling(xd, 196)
    assert torch.equal(fps[11:12], hip_ext.furthest_point_sampling(xd[11:12].contiguous(), 196))


def test_empty_batch(hip_ext):
    """B = 0 (an image whose detections were all rejected): every operator returns an empty tensor."""
    z = torch.zeros(0, 64, 3).cuda()
    assert hip_ext.furthest_point_sampling(z, 8).shape == (0, 8)
    assert hip_ext.ball_query(z, z, 0.1, 4).shape == (0, 64, 4)
    f = torch.zeros(0, 5, 64).cuda()
    assert hip_ext.gather_points(f, torch.zeros(0, 8, dtype=torch.int32).cuda()).shape == (0, 5, 8)
    assert hip_ext.group_points(f, torch.zeros(0, 8, 4, dtype=torch.int32).cuda()).shape == (0, 5, 8, 4)


def test_reference_error_behaviour(hip_ext):
    x = torch.randn(1, 10, 3)
    with pytest.raises(RuntimeError, match="CPU not supported"):
        hip_ext.furthest_point_sampling(x, 4)
    with pytest.raises(RuntimeError, match="contiguous"):
        hip_ext.furthest_point_sampling(torch.randn(1, 3, 10).cuda().transpose(1, 2), 4)
    with pytest.raises(RuntimeError, match="float"):
        hip_ext.furthest_point_sampling(x.double().cuda(), 4)
    with pytest.raises(RuntimeError, match="int"):
        hip_ext.gather_points(torch.randn(1, 3, 10).cuda(), torch.zeros(1, 4, dtype=torch.int64).cuda())


def test_pointnet2_utils_api_and_autograd(oracle_ext):
    """The reference's python-level operator API on the HIP ops: names, argument order, non-differentiable
    indices, and gradients of gather / group through the atomic scatter kernels (vs torch autograd)."""
    from unopose_amd.pointnet2 import pointnet2_utils as pu

    g = torch.Generator().manual_seed(2)
    xyz = torch.rand(2, 300, 3, generator=g).cuda()
    idx = pu.furthest_point_sample(xyz, 64)
    assert idx.dtype == torch.int32 and not idx.requires_grad
    feats = torch.randn(2, 16, 300, generator=g).cuda().requires_grad_(True)
    out = pu.gather_operation(feats, idx)
    ref = torch.gather(feats, 2, idx.long()[:, None, :].expand(-1, 16, -1))
    w = torch.randn_like(out)
    (g1,) = torch.autograd.grad((out * w).sum(), feats, retain_graph=True)
    (g2,) = torch.autograd.grad((ref * w).sum(), feats)
    torch.testing.assert_close(g1, g2, rtol=1e-5, atol=1e-5)
    bq = pu.ball_query(0.2, 16, xyz, xyz[:, :50].contiguous())  # (radius, nsample, xyz, new_xyz)
    assert torch.equal(bq.cpu(), oracle_ext.ball_query(xyz[:, :50].cpu().contiguous(), xyz.cpu(), 0.2, 16))
    grouped = pu.grouping_operation(feats, bq)
    ref = torch.gather(feats, 2, bq.long().reshape(2, 1, -1).expand(-1, 16, -1)).reshape(2, 16, 50, 16)
    w = torch.randn_like(grouped)
    (g1,) = torch.autograd.grad((grouped * w).sum(), feats, retain_graph=True)
    (g2,) = torch.autograd.grad((ref * w).sum(), feats)
    torch.testing.assert_close(g1, g2, rtol=1e-4, atol=1e-4)
    grp = pu.QueryAndLRFGroup(0.2, 32, use_xyz=True)
    f = grp(xyz, xyz, xyz.transpose(1, 2).contiguous())
    assert f.shape == (2, 6, 300, 32)
    qg = pu.QueryAndGroup(0.2, 16, use_xyz=True)
    assert qg(xyz, xyz[:, :50].contiguous(), feats.detach()).shape == (2, 19, 50, 16)
