"""CPU: the host half of the table-interpolated geometric embedding (ops._geo_tables) -- the tables csrc/embed.hip::geo_embed_table_kernel
interpolates -- checked against the defining function proj(sinus(x)) of GeometricStructureEmbedding (transformer.py:303-350), with the
kernel's own interpolation rule restated in numpy: row r holds x = (r - (NP / 2 - 1)) / 4, NP-point Lagrange on the nodes around x."""
import numpy as np
import pytest
import torch


def _module():
    from unopose_amd.model.modules import GeometricStructureEmbedding

    from types import SimpleNamespace

    torch.manual_seed(3)
    m = GeometricStructureEmbedding(SimpleNamespace(hidden_dim=256, sigma_d=0.2, sigma_a=15.0, angle_k=3, reduction_a="max"))
    with torch.no_grad():  # trained-like magnitudes: well above the 1/16 of the default initialisation
        m.proj_d.weight.mul_(4.0)
        m.proj_a.weight.mul_(4.0)
    return m.double()


def _exact(m, lin, x):
    om = x[:, None] * m.embedding.div_term.double().numpy()[None, :]
    s = np.stack([np.sin(om), np.cos(om)], -1).reshape(len(x), -1)
    return s @ lin.weight.detach().double().numpy().T


def _interp(tab, x, npoint):
    lo = npoint // 2 - 1
    t = x * 4.0
    r0 = np.floor(t).astype(int)
    f = t - r0
    out = np.zeros((len(x), tab.shape[1]))
    nodes = np.arange(npoint) - lo
    for j in range(npoint):
        w = np.ones_like(f)
        for mth in range(npoint):
            if mth != j:
                w *= (f - nodes[mth]) / (nodes[j] - nodes[mth])
        out += w[:, None] * tab[r0 + j]
    return out


@pytest.mark.parametrize("npoint,tol", [(4, 2e-4), (6, 3e-6)])
def test_tables_interpolate_the_projected_sinusoids(npoint, tol):
    from unopose_amd import ops

    m = _module()
    tabs = ops._geo_tables(m, ("cpu-test", npoint), npoint)
    assert tabs is not None
    td, ta, wd = (t.double().numpy() for t in tabs)
    lo = npoint // 2 - 1
    assert td.shape == (64 * 4 + npoint, 256) and ta.shape[0] >= int(np.floor(np.pi * m.factor_a * 4)) + npoint + 1
    assert np.array_equal(wd, m.proj_d.weight.detach().float().double().numpy())
    # the rows ARE the function on the grid (fp32 storage)
    xg = (np.arange(td.shape[0]) - lo) / 4.0
    assert np.abs(td - _exact(m, m.proj_d, xg)).max() < 2e-6
    rng = np.random.default_rng(0)
    xd = np.concatenate([rng.uniform(0.0, 63.9, 4000), [0.0, 0.25, 15.999, 16.0, 63.99]])
    xa = np.concatenate([rng.uniform(0.0, np.pi * m.factor_a, 4000), [0.0, np.pi * m.factor_a]])
    ed = np.abs(_interp(td, xd, npoint) - _exact(m, m.proj_d, xd)).max()
    ea = np.abs(_interp(ta, xa, npoint) - _exact(m, m.proj_a, xa)).max()
    assert ed < tol and ea < tol, (ed, ea)
    # every index the kernel can form stays inside the tables: the last rows read are floor(x 4) + npoint - 1
    assert int(np.floor(xa.max() * 4)) + npoint - 1 < ta.shape[0]
    assert int(np.floor(((td.shape[0] - npoint) / 4.0) * 4)) + npoint - 1 < td.shape[0]


def test_tables_follow_the_weight_version():
    from unopose_amd import ops

    m = _module().float()
    key = (m.proj_d.weight._version, m.proj_a.weight._version, m.proj_d.weight.data_ptr(), m.proj_d.weight.device)
    t0 = ops._geo_tables(m, key, 4)
    assert ops._geo_tables(m, key, 4) is t0  # cached per (weights, order)
    assert ops._geo_tables(m, key, 6) is not t0
    with torch.no_grad():
        m.proj_d.weight.add_(0.01)
    key2 = (m.proj_d.weight._version, m.proj_a.weight._version, m.proj_d.weight.data_ptr(), m.proj_d.weight.device)
    assert key2 != key
    t1 = ops._geo_tables(m, key2, 4)
    assert t1 is not t0 and not torch.equal(t1[0], t0[0])
