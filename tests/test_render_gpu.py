"""GPU: the HIP depth rasteriser (csrc/raster.hip through unopose_amd.render.HipDepthRenderer) against the numpy rasteriser with
the same float32 arithmetic (bit for bit), and the full BOP'19 average recall with it against the bop_toolkit golden values."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_hip_rasteriser_equals_the_numpy_rasteriser_bit_for_bit():
    from bop_eval_case import make_vsd_case
    from raster_np import render_depth
    from unopose_amd.render import HipDepthRenderer

    models, scene_gt, cameras, results, _, _, (W, H) = make_vsd_case()
    ren = HipDepthRenderer(W, H)
    for oid, m in models.items():
        ren.add_object(oid, m["verts"], m["faces"])
    for oid in models:
        rows = [r for r in results if r["obj_id"] == oid]
        K = cameras[rows[0]["scene_id"]][rows[0]["im_id"]]
        k4 = [K[0, 0], K[1, 1], K[0, 2], K[1, 2]]
        got = ren.render_batch(oid, np.stack([r["R"] for r in rows]), np.stack([r["t"] for r in rows]), k4).cpu().numpy()
        for g, r in zip(got, rows):
            want = render_depth(models[oid]["verts"], models[oid]["faces"], r["R"], r["t"], *k4, H, W)
            assert (g > 0).sum() > 200 and np.array_equal(g, want)
    # one pose through the toolkit interface, a pose partly outside the image, a pose behind the camera
    m = models[1]
    d = ren.render_object(1, np.eye(3), np.array([-250.0, 0.0, 600.0]), 260.0, 262.0, 118.5, 91.0)["depth"]
    assert np.array_equal(d, render_depth(m["verts"], m["faces"], np.eye(3), [-250.0, 0.0, 600.0], 260.0, 262.0, 118.5, 91.0, H, W)) and (d > 0).any()
    assert (ren.render_object(1, np.eye(3), np.array([0.0, 0.0, -600.0]), 260.0, 262.0, 118.5, 91.0)["depth"] == 0).all()


def test_bop_average_recall_with_the_hip_renderer_equals_bop_toolkit():
    from bop_eval_case import make_vsd_case
    from unopose_amd import bop_eval
    from unopose_amd.render import HipDepthRenderer

    want = json.load(open(os.path.join(GOLD, "bop_eval.json")))["vsd"]
    models, scene_gt, cameras, results, im_width, depth_images, (W, H) = make_vsd_case()
    ren = HipDepthRenderer(W, H)
    for oid, m in models.items():
        ren.add_object(oid, m["verts"], m["faces"])
    out = bop_eval.average_recall(results, scene_gt, models, cameras, im_width, n_top=1, renderer=ren, depth_images=depth_images)
    assert np.allclose(out["recalls_vsd"], want["recalls_vsd"])
    for k in ("AR_VSD", "AR_MSSD", "AR_MSPD", "AR"):
        assert abs(out[k] - want[k]) < 1e-12, k
