"""CPU: oracle/unopose_ref.py against the golden fixtures captured from the reference
(tests/golden/make_golden.py).  Does not read /root/reference."""
import os

import numpy as np
import pytest
import torch

from oracle import unopose_ref as R

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    return {k: torch.from_numpy(z[k]) if z[k].ndim else z[k].item() for k in z.files}


@pytest.fixture(scope="module")
def sd():
    return R.random_state_dict(R.default_cfg(), seed=0, prefix=None)


@pytest.fixture(scope="module")
def cfg():
    return R.default_cfg()


def close(a, b, tol):
    assert (a.float() - b.float()).abs().max().item() <= tol


def test_lrf_global():
    z = load("lrf_global")
    close(R.get_batch_lrf(z["pts"]), z["out"], 2e-5)


@pytest.mark.parametrize("name", ["query_lrf_group_r0.2_ns32", "query_lrf_group_r0.4_ns64"])
def test_query_lrf_group(oracle_ext, name):
    z = load(name)
    close(R.query_and_lrf_group(z["xyz"], z["radius"], z["nsample"], oracle_ext), z["out"], 5e-4)


def test_positional_encoding(oracle_ext, sd, cfg):
    z = load("positional_encoding")
    pcfg = R.Cfg(dict(cfg.fine_point_matching, pe_radius1=z["r1"], pe_radius2=z["r2"], nsample1=z["ns1"], nsample2=z["ns2"]))
    close(R.positional_encoding(z["xyz"], sd, "fine_point_matching.PE", pcfg, oracle_ext), z["out"], 2e-3)


def test_geo_embedding(sd, cfg):
    z = load("geo_embedding")
    d, a = R.geo_embedding_indices(z["points"], cfg.geo_embedding)
    close(d, z["d_idx"], 1e-5)
    close(a, z["a_idx"], 1e-4)
    close(R.geo_embedding(z["points"], sd, "geo_embedding", cfg.geo_embedding), z["out"], 1e-4)


def test_transformer_layers(sd, cfg):
    z = load("transformer_layers")
    geo = R.geo_embedding(z["points"], sd, "geo_embedding", cfg.geo_embedding)
    tp = "coarse_point_matching.transformers.0"
    close(R.transformer_layer(z["f0"], z["f0"], sd, tp + ".layers.0", embed=geo[0:1]), z["rpe_self"], 5e-5)
    close(R.transformer_layer(z["f0"], z["f1"], sd, tp + ".layers.1"), z["cross"], 5e-5)
    m0, m1 = R.geometric_transformer(z["f0"], geo[0:1], z["f1"], geo[1:2], sd, tp)
    close(m0, z["gt0"], 1e-4)
    close(m1, z["gt1"], 1e-4)


def test_sparse_to_dense(oracle_ext, sd, cfg):
    z = load("sparse_to_dense")
    geo = R.geo_embedding(z["points"], sd, "geo_embedding", cfg.geo_embedding)
    sp = "fine_point_matching.transformers.0"
    close(R.linear_transformer_layer(z["d0"][:, 1:].contiguous(), z["sparse0"][:, 1:].contiguous(), sd, sp + ".dense_layer"),
          z["linear"], 1e-4)
    m0, m1 = R.sparse_to_dense_transformer(z["d0"], geo[0:1], z["i0"], z["d1"], geo[1:2], z["i1"], sd, sp, oracle_ext)
    close(m0, z["out0"], 2e-4)
    close(m1, z["out1"], 2e-4)


def test_weighted_procrustes():
    z = load("weighted_procrustes")
    Rm, tm = R.weighted_procrustes(z["src"], z["ref"], z["w"], thresh=0.001)
    close(Rm, z["R"], 1e-5)
    close(tm, z["t"], 1e-5)
    Rm, tm = R.weighted_procrustes(z["src3"], z["ref3"], None, thresh=0.5)
    close(Rm, z["R3"], 1e-5)
    close(tm, z["t3"], 1e-5)


def test_coarse_rt():
    z = load("coarse_rt")
    Rm, tm, sm, det = R.compute_coarse_rt_overlap(z["atten"], z["score"], z["p1"], z["p2"], z["rand"], detail=True)
    assert torch.equal(det["idx"].to(torch.int32), z["hyp_idx"])
    close(Rm, z["R"], 1e-5)
    close(tm, z["t"], 1e-5)
    close(sm, z["pose_score"], 1e-3)
    close(Rm, z["R_gt"], 5e-3)  # the constructed problem is actually solved


def test_fine_rt():
    z = load("fine_rt")
    Rm, tm, sm = R.compute_fine_rt_overlap(z["atten"], z["score"], z["p1"], z["p2"])
    close(Rm, z["R"], 1e-5)
    close(tm, z["t"], 1e-5)
    close(sm, z["pose_score"], 1e-5)
    close(Rm, z["R_gt"], 5e-3)


@pytest.fixture(scope="module")
def sdt():
    return R.random_state_dict(R.default_cfg(), seed=0, tame=0.1)


def test_coarse_matcher(sdt, cfg):
    z = load("coarse_matcher")
    g1 = R.geo_embedding(z["lrf1"], sdt, "geo_embedding", cfg.geo_embedding)
    g2 = R.geo_embedding(z["lrf2"], sdt, "geo_embedding", cfg.geo_embedding)
    Rm, tm, sm = R.coarse_point_matching(z["p1"], z["f1"], g1, z["p2"], z["f2"], g2, sdt, "coarse_point_matching",
                                         cfg.coarse_point_matching, z["rand"])
    close(Rm, z["R"], 1e-5)
    close(tm, z["t"], 1e-5)


def test_forward_cfg1_end_to_end(oracle_ext, sdt):
    """BASELINE configs[0]: single pair, 1024 points, CPU forward -- oracle vs the reference's outputs."""
    z = load("forward_cfg1")
    cfg1 = R.default_cfg(fine_npoint=1024)
    ep = {k: z[k] for k in ("pts", "tem1_pts", "rgb", "tem1_rgb", "rgb_choose", "tem1_choose")}
    out = R.unopose_forward(ep, sdt, cfg1, z["rand"], oracle_ext, detail=True)
    assert torch.equal(out["fps_idx_m"], z["fps_idx_m"]) and torch.equal(out["fps_idx_o"], z["fps_idx_o"])
    for k in ("init_R", "init_t", "pred_R", "pred_t"):
        close(out[k], z[k], 1e-5)
    close(out["pred_R"][0], z["R_gt"], 5e-3)  # and the pose is actually right
    close(out["pred_t"][0], z["t_gt"], 5e-3)


def test_state_dict_layout_counts(sd):
    n_coarse = sum(v.numel() for k, v in sd.items() if k.startswith("coarse_point_matching"))
    n_fine = sum(v.numel() for k, v in sd.items()
                 if k.startswith("fine_point_matching") and "running" not in k and "tracked" not in k)
    assert n_coarse == 3492611 and n_fine == 5163782


# ---- production-size layer fixtures (untamed weights, from the reference modules; make_golden.py::production_size_layers)
def test_geo_embedding_n197(sd, cfg):
    z = load("geo_embedding_n197")
    E = R.geo_embedding(z["points"], sd, "geo_embedding", cfg.geo_embedding)
    close(E[:, z["sel_i"].long()][:, :, z["sel_j"].long()], z["out"], 1e-4)


def test_geometric_transformer_n197(sd, cfg):
    z = load("geometric_transformer_n197")
    E = R.geo_embedding(z["points"], sd, "geo_embedding", cfg.geo_embedding)
    tp = "coarse_point_matching.transformers.1"
    close(R.transformer_layer(z["f0"], z["f0"], sd, tp + ".layers.0", embed=E[0:1]), z["rpe_self"], 5e-5)
    close(R.transformer_layer(z["f0"], z["f1"], sd, tp + ".layers.1"), z["cross"], 5e-5)
    m0, m1 = R.geometric_transformer(z["f0"], E[0:1], z["f1"], E[1:2], sd, tp)
    close(m0, z["gt0"], 1e-4)
    close(m1, z["gt1"], 1e-4)


def test_sparse_to_dense_2049(oracle_ext, sd, cfg):
    from helpers import seeded_checked

    z = load("sparse_to_dense_2049")
    d0 = seeded_checked((1, 2049, 256), z["d_seeds"][0], z["d_checksum"][0])
    d1 = seeded_checked((1, 2049, 256), z["d_seeds"][1], z["d_checksum"][1])
    E = R.geo_embedding(z["points"], sd, "geo_embedding", cfg.geo_embedding)
    sp = "fine_point_matching.transformers.1"
    rows, rows1 = z["rows"].long(), z["rows1"].long()
    dq, kv = d0[:, 1:].contiguous(), z["sparse0"][:, 1:].contiguous()
    close(R.linear_attention(dq, kv, sd, sp + ".dense_layer.attention.attention")[:, rows], z["linear_core"], 1e-4)
    close(R.linear_transformer_layer(dq, kv, sd, sp + ".dense_layer")[:, rows], z["linear"], 1e-4)
    m0, m1 = R.sparse_to_dense_transformer(d0, E[0:1], z["i0"], d1, E[1:2], z["i1"], sd, sp, oracle_ext)
    close(m0[:, rows1], z["out0"], 2e-4)
    close(m1[:, rows1], z["out1"], 2e-4)


def test_positional_encoding_production_radii(oracle_ext, sd, cfg):
    z = load("positional_encoding_prod")
    f = cfg.fine_point_matching
    assert (z["r1"], z["r2"], z["ns1"], z["ns2"]) == (np.float32(f.pe_radius1), np.float32(f.pe_radius2), 64, 256)
    out = R.positional_encoding(z["xyz"], sd, "fine_point_matching.PE", f, oracle_ext)
    close(out[:, z["sel"].long()], z["out"], 5e-3)


def test_fine_rt_2049():
    from helpers import constructed_similarity, tensor_checksum

    z = load("fine_rt_2049")
    atten, score = constructed_similarity(z["perm"].long(), 2048, torch.Generator().manual_seed(z["sim_seed"]), n_bg=z["n_bg"])
    assert np.allclose(tensor_checksum(atten), z["sim_checksum"].numpy(), rtol=1e-9) and torch.equal(score, z["score"])
    Rm, tm, sm = R.compute_fine_rt_overlap(atten, score, z["p1"], z["p2"])
    close(Rm, z["R"], 1e-5)
    close(tm, z["t"], 1e-5)
    close(sm, z["pose_score"], 1e-5)
    close(Rm, z["R_gt"], 5e-3)


def test_vit_attention_core_is_the_softmax_definition():
    """The factored-out attention core equals the textbook per-head softmax(q k^T / sqrt(d)) v in float64."""
    g = torch.Generator().manual_seed(3)
    qkv = torch.randn(2, 37, 3 * 128, generator=g)
    out = R.vit_attention_core(qkv, 2)
    q, k, v = qkv.double().reshape(2, 37, 3, 2, 64).permute(2, 0, 3, 1, 4)
    ref = (torch.softmax(q @ k.transpose(-1, -2) / 8.0, -1) @ v).transpose(1, 2).reshape(2, 37, 128)
    close(out, ref, 1e-5)
