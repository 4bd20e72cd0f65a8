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
