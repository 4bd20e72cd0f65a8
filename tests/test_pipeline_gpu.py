"""Provider -> runner -> UNOPose.forward on the GPU: items of the synthetic BOP
folder at the production shapes (2048 observed / 5000 reference points, 224x224 crops).  The small
synthetic objects are sampled WITH replacement, so the clouds are full of duplicate points -- the case the
reference's dataset produces for small masks (pfoneref_bop_test_dataset_v2.py:200-203) and the one that
makes FPS ties routine."""
import numpy as np
import pytest
import torch

import bop_synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def images(tmp_path_factory):
    from unopose_amd.provider import BOPTestsetOneRef, collate_image

    root = str(tmp_path_factory.mktemp("bop"))
    cfg, det_path = bop_synth.build(root)
    cfg.update(img_size=224, n_sample_observed_point=2048, n_sample_template_point=5000)
    ds = BOPTestsetOneRef(cfg, "ycbv", det_path)
    np.random.seed(11)
    return [collate_image(ds[i]) for i in range(len(ds))]


@pytest.fixture(scope="module")
def model():
    from unopose_amd.model import UNOPose, default_model_cfg
    from unopose_amd.synthetic import trained_like_

    torch.manual_seed(0)
    return trained_like_(UNOPose(default_model_cfg())).cuda().eval()


@torch.no_grad()
def test_provider_items_through_the_model(images, model, tmp_path):
    from unopose_amd.runner import ReferenceCache, inference_and_save

    cache = ReferenceCache(model)
    torch.manual_seed(5)
    lines = inference_and_save(model, images, str(tmp_path / "r.csv"), instance_batch_size=2, device="cuda",
                               sync=torch.cuda.synchronize, ref_cache=cache)
    assert len(lines) == 3 and (cache.misses, cache.hits) == (2, 1)  # views (10,5,2) and (49,7,5); the first is reused
    for ln in lines:
        f = ln.split(",")
        R = np.array(f[4].split(), np.float64).reshape(3, 3)
        t = np.array(f[5].split(), np.float64)
        assert np.isfinite(R).all() and np.isfinite(t).all() and np.isfinite(float(f[3]))
        assert np.abs(R @ R.T - np.eye(3)).max() < 1e-3 and abs(np.linalg.det(R) - 1) < 1e-3
    # the same draw of the coarse hypotheses -> cached and uncached poses agree.  Loose bound: these are
    # noise crops through random weights, where the hypothesis arg-max amplifies GEMM-rounding differences
    # between a 2B-crop and a B-crop ViT batch (the tight 1e-4 check of the cache is in test_model_gpu.py)
    torch.manual_seed(5)
    lines2 = inference_and_save(model, images, str(tmp_path / "r2.csv"), instance_batch_size=2, device="cuda")
    for a, b in zip(lines, lines2):
        fa, fb = a.split(","), b.split(",")
        assert np.abs(np.array(fa[4].split(), float) - np.array(fb[4].split(), float)).max() < 5e-2
        assert np.abs(np.array(fa[5].split(), float) - np.array(fb[5].split(), float)).max() < 25.0  # mm


@torch.no_grad()
def test_duplicate_points_fps_matches_oracle(images, oracle_ext, hip_ext):
    """Clouds sampled with replacement: FPS indices still bit-exact against the oracle (tie rule)."""
    pts = images[0]["tem1_pts"][0].contiguous()
    assert pts.shape[1] == 5000 and len(np.unique(pts[0].numpy(), axis=0)) < 5000  # duplicates present
    want = oracle_ext.furthest_point_sampling(pts, 2048)
    got = hip_ext.furthest_point_sampling(pts.cuda(), 2048).cpu()
    assert torch.equal(got, want)
