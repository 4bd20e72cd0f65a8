"""CPU: the C oracle of the `_ext` operators against independent numpy models."""
import numpy as np
import pytest
import torch

from helpers import fps_closed_form, object_cloud


@pytest.mark.parametrize("n,m", [(5000, 300), (2048, 196), (700, 64), (196, 196), (64, 10), (3, 3), (1, 1), (5, 9)])
def test_fps_literal_block_emulation_equals_closed_form(oracle_ext, n, m):
    g = torch.Generator().manual_seed(n * 7 + m)
    x = torch.randn(2, n, 3, generator=g)
    if n >= 64:  # duplicated points make ties routine (dataset samples with replacement)
        x[0, n // 2: n // 2 + n // 4] = x[0, : n // 4]
    idx = oracle_ext.furthest_point_sampling(x.contiguous(), m)
    assert idx.dtype == torch.int32 and tuple(idx.shape) == (2, m)
    for b in range(2):
        np.testing.assert_array_equal(idx[b].numpy(), fps_closed_form(x[b].numpy(), m))


def test_fps_all_points_identical(oracle_ext):
    x = torch.ones(1, 600, 3)
    idx = oracle_ext.furthest_point_sampling(x, 8)[0].numpy()
    # all distances tie at 0: winner minimises (bitrev9(k mod 512), k) -> k = 0 every time
    np.testing.assert_array_equal(idx, np.zeros(8, np.int32))


def _ball_query_np(new_xyz, xyz, r, ns):
    r2 = np.float32(r) * np.float32(r)
    out = np.zeros((len(new_xyz), ns), np.int32)
    for j, c in enumerate(new_xyz):
        d = c[None, :] - xyz
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        hits = np.where(d2 < r2)[0][:ns]
        if len(hits):
            out[j, :] = hits[0]
            out[j, : len(hits)] = hits
    return out


@pytest.mark.parametrize("n,m,r,ns", [(2048, 128, 0.2, 64), (500, 77, 0.5, 16), (100, 100, 0.01, 8), (65, 3, 10.0, 256)])
def test_ball_query(oracle_ext, n, m, r, ns):
    g = torch.Generator().manual_seed(n + m)
    xyz = torch.rand(2, n, 3, generator=g)
    new_xyz = xyz[:, :m].clone() if m <= n else torch.rand(2, m, 3, generator=g)
    if r < 0.05:
        new_xyz = new_xyz + 5.0  # nothing in range: rows must stay zero
    idx = oracle_ext.ball_query(new_xyz.contiguous(), xyz.contiguous(), r, ns)
    for b in range(2):
        np.testing.assert_array_equal(idx[b].numpy(), _ball_query_np(new_xyz[b].numpy(), xyz[b].numpy(), r, ns))


def test_gather_group_and_grads(oracle_ext):
    g = torch.Generator().manual_seed(3)
    pts = torch.randn(2, 5, 40, generator=g)
    idx = torch.randint(0, 40, (2, 13), generator=g, dtype=torch.int32)
    out = oracle_ext.gather_points(pts, idx)
    ref = torch.gather(pts, 2, idx.long()[:, None, :].expand(-1, 5, -1))
    assert torch.equal(out, ref)
    go = torch.randn(2, 5, 13, generator=g)
    gp = oracle_ext.gather_points_grad(go, idx, 40)
    ref = torch.zeros(2, 5, 40).scatter_add_(2, idx.long()[:, None, :].expand(-1, 5, -1), go)
    torch.testing.assert_close(gp, ref)

    gidx = torch.randint(0, 40, (2, 7, 6), generator=g, dtype=torch.int32)
    out = oracle_ext.group_points(pts, gidx)
    ref = torch.gather(pts, 2, gidx.long().reshape(2, 1, 42).expand(-1, 5, -1)).reshape(2, 5, 7, 6)
    assert torch.equal(out, ref)
    go = torch.randn(2, 5, 7, 6, generator=g)
    gp = oracle_ext.group_points_grad(go, gidx, 40)
    ref = torch.zeros(2, 5, 40).scatter_add_(2, gidx.long().reshape(2, 1, 42).expand(-1, 5, -1), go.reshape(2, 5, 42))
    torch.testing.assert_close(gp, ref)


def test_three_nn_and_interpolate(oracle_ext):
    g = torch.Generator().manual_seed(4)
    unknown = torch.randn(2, 50, 3, generator=g)
    known = torch.randn(2, 20, 3, generator=g)
    dist2, idx = oracle_ext.three_nn(unknown, known)
    d = ((unknown[:, :, None, :] - known[:, None, :, :]) ** 2).sum(-1)
    dref, iref = d.topk(3, dim=2, largest=False)
    torch.testing.assert_close(dist2, dref, rtol=1e-5, atol=1e-6)
    assert torch.equal(idx.long(), iref)
    feats = torch.randn(2, 4, 20, generator=g)
    w = torch.rand(2, 50, 3, generator=g)
    out = oracle_ext.three_interpolate(feats, idx, w)
    ref = (torch.gather(feats[:, :, None, :].expand(-1, -1, 50, -1), 3, idx.long()[:, None].expand(-1, 4, -1, -1)) * w[:, None]).sum(-1)
    torch.testing.assert_close(out, ref)
    go = torch.randn(2, 4, 50, generator=g)
    gp = oracle_ext.three_interpolate_grad(go, idx, w, 20)
    ref = torch.zeros(2, 4, 20)
    ref.scatter_add_(2, idx.long().reshape(2, 1, 150).expand(-1, 4, -1), (go[..., None] * w[:, None]).reshape(2, 4, 150))
    torch.testing.assert_close(gp, ref)


def test_object_cloud_shapes():
    g = torch.Generator().manual_seed(0)
    p = object_cloud(g, 2048, with_replacement=True)
    assert p.shape == (2048, 3) and len(torch.unique(p, dim=0)) < 2048
