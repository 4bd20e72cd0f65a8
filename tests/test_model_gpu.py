"""GPU parity of the product model (unopose_amd.model) against the golden fixtures captured from
the reference's Python (tests/golden/*.npz).  Tolerance for R / t: 1e-4 (BASELINE.json north_star)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    return {k: torch.from_numpy(z[k]).cuda() if z[k].ndim else z[k].item() for k in z.files}


@pytest.fixture(scope="module")
def model():
    from oracle.unopose_ref import default_cfg, random_state_dict  # weights only (test infrastructure)
    from unopose_amd.model import UNOPose, default_model_cfg

    m = UNOPose(default_model_cfg())
    m.load_state_dict(random_state_dict(default_cfg(), seed=0), strict=True)
    return m.cuda().eval()


def err(a, b):
    return (a.float() - b.float()).abs().max().item()


@torch.no_grad()
def test_geo_embedding(model):
    z = load("geo_embedding")
    assert err(model.geo_embedding(z["points"]), z["out"]) < 1e-4


@torch.no_grad()
def test_transformer_layers(model):
    z = load("transformer_layers")
    geo = model.geo_embedding(z["points"])
    gt = model.coarse_point_matching.transformers[0]
    assert err(gt.layers[0](z["f0"], z["f0"], geo[0:1]), z["rpe_self"]) < 1e-4
    assert err(gt.layers[1](z["f0"], z["f1"]), z["cross"]) < 1e-4
    m0, m1 = gt(z["f0"], geo[0:1], z["f1"], geo[1:2])
    assert err(m0, z["gt0"]) < 2e-4 and err(m1, z["gt1"]) < 2e-4


@torch.no_grad()
def test_sparse_to_dense(model):
    z = load("sparse_to_dense")
    geo = model.geo_embedding(z["points"])
    s2d = model.fine_point_matching.transformers[0]
    assert err(s2d.dense_layer(z["d0"][:, 1:], z["sparse0"][:, 1:]), z["linear"]) < 2e-4
    m0, m1 = s2d(z["d0"], geo[0:1], z["i0"], z["d1"], geo[1:2], z["i1"])
    assert err(m0, z["out0"]) < 5e-4 and err(m1, z["out1"]) < 5e-4


@torch.no_grad()
def test_positional_encoding(model):
    from unopose_amd.model.modules import PositionalEncoding

    z = load("positional_encoding")
    pe = PositionalEncoding(256, z["r1"], z["r2"], z["ns1"], z["ns2"]).cuda().eval()
    pe.load_state_dict(model.fine_point_matching.PE.state_dict())
    out = pe(z["xyz"])
    # per-point frames of ill-conditioned neighbourhoods are implementation-defined (see test_geom_gpu);
    # the max-pooled MLP output is compared on the bulk of the points
    e = (out - z["out"]).abs().amax(dim=2)
    assert (e < 5e-3).float().mean().item() > 0.5
    assert e.median().item() < 1e-3


@torch.no_grad()
def test_pose_heads_on_constructed_similarity():
    from unopose_amd import ops

    z = load("coarse_rt")
    R, t, s = ops.coarse_pose(z["atten"], z["score"], z["p1"], z["p2"], z["rand"], 6000, 300)
    assert err(R, z["R"]) < 1e-4 and err(t, z["t"]) < 1e-4 and err(s, z["pose_score"]) < 1e-2
    z = load("fine_rt")
    R, t, s = ops.fine_pose(z["atten"], z["score"], z["p1"], z["p2"])
    assert err(R, z["R"]) < 1e-4 and err(t, z["t"]) < 1e-4 and err(s, z["pose_score"]) < 1e-4


@torch.no_grad()
def test_coarse_matcher(model):
    z = load("coarse_matcher")
    g1, g2 = model.geo_embedding(z["lrf1"]), model.geo_embedding(z["lrf2"])
    ep = model.coarse_point_matching(z["p1"], z["f1"], g1, z["p2"], z["f2"], g2, torch.ones(1).cuda(),
                                     {"coarse_rand": z["rand"]})
    assert err(ep["init_R"], z["R"]) < 1e-4 and err(ep["init_t"], z["t"]) < 1e-4


@torch.no_grad()
def test_fine_matcher(model):
    z = load("fine_matcher")
    g1, g2 = model.geo_embedding(z["lrf1"]), model.geo_embedding(z["lrf2"])
    ep = {"init_R": z["init_R"], "init_t": z["init_t"]}
    ep = model.fine_point_matching(z["p1"], z["f1"], g1, z["i1"], z["p2"], z["f2"], g2, z["i2"],
                                   torch.ones(1).cuda(), ep)
    assert err(ep["pred_R"], z["R"]) < 1e-4 and err(ep["pred_t"], z["t"]) < 1e-4
    assert err(ep["pred_pose_score"], z["pose_score"]) < 1e-4


@torch.no_grad()
def test_forward_full_golden(model):
    """UNOPose.forward at the reference's full sizes (2048 / 5000 / 196 points, 224x224 crops), B=1."""
    z = load("forward_full")
    ep = {k: z[k] for k in ("pts", "tem1_pts", "rgb", "tem1_rgb", "rgb_choose", "tem1_choose")}
    ep["coarse_rand"] = z["rand"]
    out = model(ep)
    assert out is ep  # mutates and returns the same dict
    for k in ("init_R", "init_t", "pred_R", "pred_t"):
        assert err(out[k], z[k]) < 1e-4, (k, err(out[k], z[k]))
    assert err(out["init_pose_score"], z["init_pose_score"]) < 1e-2
    assert err(out["pred_pose_score"], z["pred_pose_score"]) < 1e-4
