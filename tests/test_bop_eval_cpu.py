"""CPU: unopose_amd/bop_eval.py (MSSD / MSPD errors, greedy matching, recall averaging) against the reference's vendored
bop_toolkit_lib run on the same synthetic problem (tests/golden/make_bop_eval_golden.py), + the CSV reader on runner output."""
import json
import os
import tempfile

import numpy as np

from bop_eval_case import make_case
from unopose_amd import bop_eval

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_errors_and_recalls_match_bop_toolkit():
    want = json.load(open(os.path.join(GOLD, "bop_eval.json")))
    models, scene_gt, cameras, results, im_width = make_case()
    # per-estimate errors against every ground truth of its object, in result order
    for i, r in enumerate(results):
        m = models[r["obj_id"]]
        gts = [g for g in scene_gt[r["scene_id"]][r["im_id"]] if g["obj_id"] == r["obj_id"]]
        e1 = [bop_eval.mssd(r["R"], r["t"], g["R"], g["t"], m["pts"], m["symmetries"]) / m["diameter"] for g in gts]
        e2 = [bop_eval.mspd(r["R"], r["t"], g["R"], g["t"], cameras[r["scene_id"]][r["im_id"]], m["pts"], m["symmetries"]) * 640.0 / im_width
              for g in gts]
        assert np.allclose(e1, want["errors_mssd"][i], rtol=1e-9, atol=1e-12) and np.allclose(e2, want["errors_mspd"][i], rtol=1e-9, atol=1e-9)
    out = bop_eval.average_recall(results, scene_gt, models, cameras, im_width, n_top=1)
    assert np.allclose(out["recalls_mssd"], want["recalls_mssd"]) and np.allclose(out["recalls_mspd"], want["recalls_mspd"])
    assert abs(out["AR_MSSD"] - want["AR_MSSD"]) < 1e-12 and abs(out["AR_MSPD"] - want["AR_MSPD"]) < 1e-12
    assert out["AR_VSD"] is None  # needs a renderer: never faked
    assert 0.2 < out["recalls_mssd"][0] < out["recalls_mssd"][-1] < 1.0  # a graded, non-degenerate problem


def test_symmetric_twin_scores_zero_error():
    models, *_ = make_case()
    m = models[2]
    R, t = np.eye(3), np.array([0.0, 0.0, 800.0])
    twin = R @ m["symmetries"][1]["R"]
    assert bop_eval.mssd(twin, t, R, t, m["pts"], m["symmetries"]) < 1e-9
    assert bop_eval.mssd(twin, t, R, t, m["pts"], m["symmetries"][:1]) > 10.0  # without the symmetry it is a gross error


def test_read_results_round_trips_runner_csv():
    from unopose_amd.runner import csv_line

    R = np.arange(9, dtype=np.float32) * 0.1
    line = csv_line(48, 3, 5, np.float32(0.75), R, np.array([1.5, -2.0, 900.25], np.float32), 0.3)
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "r.csv")
        open(p, "w").write(line + line)
        rows = bop_eval.read_results(p)
    assert len(rows) == 2 and rows[0]["scene_id"] == 48 and rows[0]["obj_id"] == 5 and abs(rows[0]["score"] - 0.75) < 1e-7
    assert np.allclose(rows[0]["R"].reshape(-1), R, atol=1e-7) and np.allclose(rows[0]["t"], [1.5, -2.0, 900.25])


def test_vsd_errors_recalls_and_bop_ar_match_bop_toolkit():
    """VSD (pose_error.py:17-101) for every (estimate, ground truth, tau), the 10 x 10 recalls and AR = mean(AR_VSD, AR_MSSD, AR_MSPD)
    against the toolkit run on the same problem with the same rendered depth (tests/raster_np.py handed to it as `renderer`)."""
    from bop_eval_case import make_vsd_case
    from raster_np import NumpyRenderer

    want = json.load(open(os.path.join(GOLD, "bop_eval.json")))["vsd"]
    models, scene_gt, cameras, results, im_width, depth_images, (W, H) = make_vsd_case()
    ren = NumpyRenderer(W, H)
    for oid, m in models.items():
        ren.add_object(oid, m["verts"], m["faces"])
    for i, r in enumerate(results):
        m, K = models[r["obj_id"]], cameras[r["scene_id"]][r["im_id"]]
        d_est = ren.render_object(r["obj_id"], r["R"], r["t"], K[0, 0], K[1, 1], K[0, 2], K[1, 2])["depth"]
        gts = [g for g in scene_gt[r["scene_id"]][r["im_id"]] if g["obj_id"] == r["obj_id"]]
        for g, w in zip(gts, want["errors_vsd"][i]):
            d_gt = ren.render_object(r["obj_id"], g["R"], g["t"], K[0, 0], K[1, 1], K[0, 2], K[1, 2])["depth"]
            e = bop_eval.vsd(d_est, d_gt, depth_images[r["scene_id"]][r["im_id"]], K, 15.0, bop_eval.VSD_TAUS, m["diameter"])
            assert np.allclose(e, w, rtol=0, atol=1e-12)
    out = bop_eval.average_recall(results, scene_gt, models, cameras, im_width, n_top=1, renderer=ren, depth_images=depth_images)
    assert np.allclose(out["recalls_vsd"], want["recalls_vsd"]) and np.allclose(out["recalls_mssd"], want["recalls_mssd"])
    assert np.allclose(out["recalls_mspd"], want["recalls_mspd"])
    for k in ("AR_VSD", "AR_MSSD", "AR_MSPD", "AR"):
        assert abs(out[k] - want[k]) < 1e-12, k
    assert 0.3 < out["AR_VSD"] < 0.9 and abs(out["AR"] - np.mean([out["AR_VSD"], out["AR_MSSD"], out["AR_MSPD"]])) < 1e-12
    # occlusion matters: without the test depth (all "no depth") the occluded object's estimates score differently
    blank = {s: {i: np.zeros_like(d) for i, d in ims.items()} for s, ims in depth_images.items()}
    assert bop_eval.average_recall(results, scene_gt, models, cameras, im_width, 1, ren, blank)["AR_VSD"] != out["AR_VSD"]


def test_numpy_rasteriser_against_analytic_depth():
    """The checker's checker: a tilted plane (two triangles) and a finely tessellated sphere against closed-form depth."""
    from bop_eval_case import icosphere
    from raster_np import render_depth

    K = (300.0, 300.0, 79.5, 59.5)
    # plane through (0,0,700) with normal n: z(x, y) = n.p0 / n.(ray)
    n, p0 = np.array([0.2, -0.1, 1.0]), np.array([0.0, 0.0, 700.0])
    ex = np.cross(n, [0, 1, 0]); ex /= np.linalg.norm(ex)
    ey = np.cross(n, ex); ey /= np.linalg.norm(ey)
    quad = np.stack([p0 + 1200 * (a * ex + b * ey) for a, b in ((-1, -1), (1, -1), (1, 1), (-1, 1))])
    d = render_depth(quad, [[0, 1, 2], [0, 2, 3]], np.eye(3), np.zeros(3), *K, 120, 160)
    xs, ys = np.meshgrid(np.arange(160), np.arange(120))
    rays = np.stack([(xs - K[2]) / K[0], (ys - K[3]) / K[1], np.ones_like(xs, float)], -1)
    assert (d > 0).all() and np.abs(d - (n @ p0) / (rays @ n)).max() < 0.05
    v, f = icosphere()
    for _ in range(2):  # two more subdivisions: 642 vertices
        cache, nf = {}, []
        v = list(v)
        for a, b, c in f:
            m = []
            for i, j in ((a, b), (b, c), (c, a)):
                k = (min(i, j), max(i, j))
                if k not in cache:
                    p = v[i] + v[j]
                    v.append(p / np.linalg.norm(p))
                    cache[k] = len(v) - 1
                m.append(cache[k])
            nf += [(a, m[0], m[2]), (b, m[1], m[0]), (c, m[2], m[1]), (m[0], m[1], m[2])]
        v, f = np.stack(v), np.asarray(nf)
    c, r = np.array([10.0, -5.0, 600.0]), 80.0
    d = render_depth(v * r, f, np.eye(3), c, *K, 120, 160)
    bq = rays @ c
    disc = bq ** 2 - (rays ** 2).sum(-1) * (c @ c - r * r)
    z = (bq - np.sqrt(np.maximum(disc, 0))) / (rays ** 2).sum(-1)
    inner = disc > 2000.0  # well inside the silhouette
    err = d[inner] - z[inner]  # the inscribed mesh lies behind the sphere surface by at most the chord sagitta / cos(view angle)
    assert (d[inner] > 0).all() and err.min() > -1e-3 and err.max() < 1.0
    assert (d[disc < -50.0] == 0).all()
