"""BOP'19 localization scoring of a result CSV at the level this environment allows.

What the reference runs after `save_unopose.sh` is bop_toolkit's `eval_bop19_pose.py`: per estimate the pose errors
VSD, MSSD and MSPD against the ground truth, greedy matching per image / object in order of decreasing score, recall per
correctness threshold, AR = mean of the three average recalls (`core/unopose/engine/bop_eval_utils.py:340-454` then only
tabulates the toolkit's score files).  This module implements the two errors that need no renderer -- **MSSD** (thresholds
0.05 ... 0.5 of the object diameter) and **MSPD** (5 ... 50 px at 640 px image width) -- the matching and the recall
averaging, vectorised over symmetries and points; **VSD needs a depth renderer and the `models_eval` meshes and is not
computed**, so `average_recall` reports AR_MSSD, AR_MSPD and their mean, never a BOP AR.  Pinned against bop_toolkit_lib's
own `pose_error.mssd / mspd`, `pose_matching.match_poses_scene` and `score.calc_localization_scores`
(tests/golden/make_bop_eval_golden.py).  Host code (numpy): scoring runs once per result file, off the hot path."""
import numpy as np

MSSD_THRESHOLDS = np.arange(0.05, 0.51, 0.05)  # fractions of the object diameter
MSPD_THRESHOLDS = np.arange(5, 51, 5)          # pixels at 640 px image width


def read_results(path):
    """The runner's CSV (scene_id,im_id,obj_id,score,R 9 values,t 3 values in mm,time) -> list of dicts."""
    out = []
    with open(path) as f:
        for line in f:
            c = line.strip().split(",")
            if len(c) != 7 or c[0] == "scene_id":
                continue
            out.append(dict(scene_id=int(c[0]), im_id=int(c[1]), obj_id=int(c[2]), score=float(c[3]),
                            R=np.array(c[4].split(), dtype=np.float64).reshape(3, 3), t=np.array(c[5].split(), dtype=np.float64),
                            time=float(c[6])))
    return out


def _sym_poses(R_gt, t_gt, syms):
    """Ground-truth pose composed with every symmetry of the object: (S,3,3), (S,3).  syms: list of {"R", "t"}."""
    Rs = np.stack([np.asarray(s["R"], np.float64).reshape(3, 3) for s in syms])
    ts = np.stack([np.asarray(s["t"], np.float64).reshape(3) for s in syms])
    return R_gt @ Rs, ts @ R_gt.T + t_gt.reshape(1, 3)


def mssd(R_est, t_est, R_gt, t_gt, pts, syms):
    """Maximum symmetry-aware surface distance: min over symmetries of the largest point displacement (model units)."""
    est = pts @ R_est.T + t_est.reshape(1, 3)
    Rg, tg = _sym_poses(R_gt, t_gt.reshape(3), syms)
    gt = np.einsum("sij,nj->sni", Rg, pts) + tg[:, None, :]
    return float(np.linalg.norm(gt - est[None], axis=2).max(axis=1).min())


def _project(pts_cam, K):
    uvw = pts_cam @ K.T
    return uvw[..., :2] / uvw[..., 2:3]


def mspd(R_est, t_est, R_gt, t_gt, K, pts, syms):
    """Maximum symmetry-aware projection distance in pixels."""
    est = _project(pts @ R_est.T + t_est.reshape(1, 3), K)
    Rg, tg = _sym_poses(R_gt, t_gt.reshape(3), syms)
    gt = _project(np.einsum("sij,nj->sni", Rg, pts) + tg[:, None, :], K)
    return float(np.linalg.norm(gt - est[None], axis=2).max(axis=1).min())


def _recall_at(per_image, threshold):
    """Greedy matching (decreasing score; an estimate takes the free ground truth of its object with the smallest error
    below the threshold) and recall = matched valid ground truths / valid ground truths."""
    tp = targets = 0
    for gts, ests in per_image:
        targets += sum(1 for g in gts if g["valid"])
        taken = set()
        for e in sorted(ests, key=lambda e: e["score"], reverse=True):
            best, best_err = -1, threshold
            for gid, err in e["errors"].items():
                if gts[gid]["valid"] and gid not in taken and err < best_err:
                    best, best_err = gid, err
            if best >= 0:
                taken.add(best)
        tp += len(taken)
    return tp / targets if targets else 0.0


def average_recall(results, scene_gt, models, cameras, im_width, n_top=1):
    """results: `read_results` rows; scene_gt[scene_id][im_id] = list of {"obj_id", "R" (3,3), "t" (3,) mm, optional "valid"};
    models[obj_id] = {"pts" (n,3) mm, "diameter", "symmetries": [{"R","t"}] incl. identity}; cameras[scene_id][im_id] = K.
    Only the `n_top` best-scored estimates per (image, object) take part (BOP: the instance count of the target).
    -> {"AR_MSSD", "AR_MSPD", "AR_MSSD_MSPD", "recalls_mssd", "recalls_mspd", "AR_VSD": None}."""
    by_im = {}
    for r in results:
        by_im.setdefault((r["scene_id"], r["im_id"]), []).append(r)
    sets = {"mssd": [], "mspd": []}
    for sid, ims in scene_gt.items():
        for iid, gts in ims.items():
            gts = [dict(g, valid=g.get("valid", True)) for g in gts]
            per_obj = {}
            for r in by_im.get((sid, iid), []):
                per_obj.setdefault(r["obj_id"], []).append(r)
            ests = {"mssd": [], "mspd": []}
            for obj_id, rows in per_obj.items():
                rows = sorted(rows, key=lambda r: r["score"], reverse=True)[:n_top if n_top > 0 else None]
                m = models[obj_id]
                for r in rows:
                    e1, e2 = {}, {}
                    for gid, g in enumerate(gts):
                        if g["obj_id"] != obj_id:
                            continue
                        Rg, tg = np.asarray(g["R"], np.float64), np.asarray(g["t"], np.float64)
                        e1[gid] = mssd(r["R"], r["t"], Rg, tg, m["pts"], m["symmetries"]) / m["diameter"]
                        e2[gid] = mspd(r["R"], r["t"], Rg, tg, cameras[sid][iid], m["pts"], m["symmetries"]) * (640.0 / im_width)
                    ests["mssd"].append(dict(score=r["score"], errors=e1))
                    ests["mspd"].append(dict(score=r["score"], errors=e2))
            for k in sets:
                sets[k].append((gts, ests[k]))
    rec_s = [_recall_at(sets["mssd"], th) for th in MSSD_THRESHOLDS]
    rec_p = [_recall_at(sets["mspd"], th) for th in MSPD_THRESHOLDS]
    ar_s, ar_p = float(np.mean(rec_s)), float(np.mean(rec_p))
    return dict(AR_MSSD=ar_s, AR_MSPD=ar_p, AR_MSSD_MSPD=0.5 * (ar_s + ar_p), recalls_mssd=rec_s, recalls_mspd=rec_p, AR_VSD=None)
