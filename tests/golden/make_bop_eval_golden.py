"""Expected MSSD / MSPD errors and recalls for tests/bop_eval_case.py from the REFERENCE's vendored bop_toolkit_lib
(third_party/bop_toolkit): pose_error.mssd / mspd, pose_matching.match_poses_scene, score.calc_localization_scores, with the
normalisations of scripts/eval_calc_scores.py:222-234 and the thresholds of scripts/eval_bop19_pose.py:41-44.
    python tests/golden/make_bop_eval_golden.py   ->  tests/golden/bop_eval.json"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE)))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/third_party/bop_toolkit")

from bop_eval_case import make_case  # noqa: E402
from bop_toolkit_lib import pose_error, pose_matching, score  # noqa: E402


def main():
    models, scene_gt, cameras, results, im_width = make_case()
    n_top = 1
    out = {"errors_mssd": [], "errors_mspd": []}
    recalls = {}
    for kind, ths in (("mssd", np.arange(0.05, 0.51, 0.05)), ("mspd", np.arange(5, 51, 5))):
        scene_errs = {sid: [] for sid in scene_gt}
        est_id = 0
        for r in results:
            m = models[r["obj_id"]]
            errs = {}
            for gid, g in enumerate(scene_gt[r["scene_id"]][r["im_id"]]):
                if g["obj_id"] != r["obj_id"]:
                    continue
                a = (r["R"], r["t"].reshape(3, 1), g["R"], g["t"].reshape(3, 1))
                if kind == "mssd":
                    e = pose_error.mssd(*a, m["pts"], m["symmetries_bop"]) / m["diameter"]
                else:
                    e = pose_error.mspd(*a, cameras[r["scene_id"]][r["im_id"]], m["pts"], m["symmetries_bop"]) * 640.0 / im_width
                errs[gid] = [float(e)]
            out["errors_" + kind].append([errs[k][0] for k in sorted(errs)])
            scene_errs[r["scene_id"]].append(dict(im_id=r["im_id"], obj_id=r["obj_id"], est_id=est_id, score=r["score"], errors=errs))
            est_id += 1
        recs = []
        for th in ths:
            matches = []
            for sid in scene_gt:
                gt_valid = {iid: [g["valid"] for g in gts] for iid, gts in scene_gt[sid].items()}
                matches += pose_matching.match_poses_scene(sid, scene_gt[sid], gt_valid, scene_errs[sid], [th], n_top)
            recs.append(score.calc_localization_scores(list(scene_gt), list(models), matches, n_top, do_print=False)["recall"])
        recalls[kind] = recs
    out.update(recalls_mssd=recalls["mssd"], recalls_mspd=recalls["mspd"], AR_MSSD=float(np.mean(recalls["mssd"])),
               AR_MSPD=float(np.mean(recalls["mspd"])))
    out["vsd"] = vsd_case()
    json.dump(out, open(os.path.join(HERE, "bop_eval.json"), "w"), indent=0)
    print({k: v for k, v in out.items() if k.startswith("AR") or k.startswith("recalls")})
    print({k: v for k, v in out["vsd"].items() if k.startswith("AR")})


def vsd_case():
    """The toolkit's own pose_error.vsd / mssd / mspd, matching and scoring on tests/bop_eval_case.make_vsd_case, with the numpy
    rasteriser of tests/raster_np.py as the toolkit's `renderer`: VSD errors per (estimate, ground truth, tau), the 10 x 10 recalls,
    AR_VSD / AR_MSSD / AR_MSPD and the BOP'19 AR = their mean (scripts/eval_bop19_pose.py:17-44, 218-222)."""
    from bop_eval_case import make_vsd_case
    from raster_np import NumpyRenderer

    models, scene_gt, cameras, results, im_width, depth_images, (W, H) = make_vsd_case()
    ren = NumpyRenderer(W, H)
    for oid, m in models.items():
        ren.add_object(oid, m["verts"], m["faces"])
        m["symmetries_bop"] = [dict(R=s["R"], t=s["t"].reshape(3, 1)) for s in m["symmetries"]]
    taus = list(np.arange(0.05, 0.51, 0.05))
    n_top = 1
    errs = {"vsd": [], "mssd": [], "mspd": []}
    for r in results:
        m = models[r["obj_id"]]
        K = cameras[r["scene_id"]][r["im_id"]]
        e = {"vsd": {}, "mssd": {}, "mspd": {}}
        for gid, g in enumerate(scene_gt[r["scene_id"]][r["im_id"]]):
            if g["obj_id"] != r["obj_id"]:
                continue
            a = (r["R"], r["t"].reshape(3, 1), g["R"], g["t"].reshape(3, 1))
            e["vsd"][gid] = [float(x) for x in pose_error.vsd(*a, depth_images[r["scene_id"]][r["im_id"]], K, 15.0, taus, True, m["diameter"], ren,
                                                              r["obj_id"], "step")]
            e["mssd"][gid] = [float(pose_error.mssd(*a, m["pts"], m["symmetries_bop"]) / m["diameter"])]
            e["mspd"][gid] = [float(pose_error.mspd(*a, K, m["pts"], m["symmetries_bop"]) * 640.0 / im_width)]
        for k in errs:
            errs[k].append(e[k])

    def recalls(kind, idx, ths):
        scene_errs = {sid: [] for sid in scene_gt}
        for est_id, (r, e) in enumerate(zip(results, errs[kind])):
            scene_errs[r["scene_id"]].append(dict(im_id=r["im_id"], obj_id=r["obj_id"], est_id=est_id, score=r["score"],
                                                  errors={gid: [v[idx]] for gid, v in e.items()}))
        recs = []
        for th in ths:
            matches = []
            for sid in scene_gt:
                gt_valid = {iid: [g["valid"] for g in gts] for iid, gts in scene_gt[sid].items()}
                matches += pose_matching.match_poses_scene(sid, scene_gt[sid], gt_valid, scene_errs[sid], [th], n_top)
            recs.append(score.calc_localization_scores(list(scene_gt), list(models), matches, n_top, do_print=False)["recall"])
        return recs

    rec_v = [recalls("vsd", ti, np.arange(0.05, 0.51, 0.05)) for ti in range(len(taus))]
    rec_s, rec_p = recalls("mssd", 0, np.arange(0.05, 0.51, 0.05)), recalls("mspd", 0, np.arange(5, 51, 5))
    ar = [float(np.mean(rec_v)), float(np.mean(rec_s)), float(np.mean(rec_p))]
    return dict(errors_vsd=[[e[k] for k in sorted(e)] for e in errs["vsd"]], recalls_vsd=rec_v, recalls_mssd=rec_s, recalls_mspd=rec_p,
                AR_VSD=ar[0], AR_MSSD=ar[1], AR_MSPD=ar[2], AR=float(np.mean(ar)))


if __name__ == "__main__":
    # the toolkit wants symmetries as {"R": 3x3, "t": 3x1}
    import bop_eval_case

    _orig = bop_eval_case.make_case

    def _wrapped(*a, **k):
        models, *rest = _orig(*a, **k)
        for m in models.values():
            m["symmetries_bop"] = [dict(R=s["R"], t=s["t"].reshape(3, 1)) for s in m["symmetries"]]
        return (models, *rest)

    make_case = _wrapped
    main()
