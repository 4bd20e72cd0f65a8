"""BOP'19 localization scoring of a result CSV at the level this environment allows.

What the reference runs after `save_unopose.sh` is bop_toolkit's `eval_bop19_pose.py`: per estimate the pose errors
VSD, MSSD and MSPD against the ground truth, greedy matching per image / object in order of decreasing score, recall per
correctness threshold, AR = mean of the three average recalls (`core/unopose/engine/bop_eval_utils.py:340-454` then only
tabulates the toolkit's score files).  This module implements all three: **MSSD** (thresholds 0.05 ... 0.5 of the object
diameter), **MSPD** (5 ... 50 px at 640 px image width) and **VSD** (misalignment tolerances tau = 0.05 ... 0.5 of the diameter x
correctness thresholds 0.05 ... 0.5, visibility tolerance delta = 15 mm; `pose_error.py:17-101`, `visibility.py:44-70`,
`misc.py:142-162`), the matching and the recall averaging.  VSD needs depth maps of the object model in the estimated and the
ground-truth pose: `average_recall(..., renderer=..., depth_images=...)` takes any object with the toolkit's `render_object`
call -- `unopose_amd.render.HipDepthRenderer` is the HIP rasteriser (csrc/raster.hip); without a renderer AR_VSD and the BOP AR
are None, never faked.  Pinned against bop_toolkit_lib's own `pose_error.vsd / mssd / mspd`, `pose_matching.match_poses_scene` and
`score.calc_localization_scores` (tests/golden/make_bop_eval_golden.py; for VSD both sides score the same rendered depth).
Host code (numpy): scoring runs once per result file, off the hot path; only the rasteriser is a device kernel."""
import numpy as np

MSSD_THRESHOLDS = np.arange(0.05, 0.51, 0.05)  # fractions of the object diameter
MSPD_THRESHOLDS = np.arange(5, 51, 5)          # pixels at 640 px image width
VSD_TAUS = np.arange(0.05, 0.51, 0.05)         # misalignment tolerances, fractions of the object diameter
VSD_THRESHOLDS = np.arange(0.05, 0.51, 0.05)   # correctness thresholds on the VSD error
VSD_DELTA = 15.0                               # visibility tolerance in mm (every BOP dataset but ITODD: bop_eval_utils.py:348-362)


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


def depth_to_dist(depth, K):
    """Depth image (z) -> distance to the camera centre, 0 where there is no depth (misc.py:142-162)."""
    H, W = depth.shape
    xs, ys = np.meshgrid(np.arange(W), np.arange(H))
    pre_x, pre_y = (xs - K[0, 2]) / np.float64(K[0, 0]), (ys - K[1, 2]) / np.float64(K[1, 1])
    return np.sqrt(np.multiply(pre_x, depth) ** 2 + np.multiply(pre_y, depth) ** 2 + depth.astype(np.float64) ** 2)


def _visib_mask(d_test, d_model, delta):
    """visibility.py:30-41, visib_mode "bop19": the model surface is visible where it is not behind the scene by more than
    delta, or where the scene has no depth."""
    d_diff = d_model.astype(np.float32) - d_test.astype(np.float32)
    return np.logical_and(np.logical_or(d_diff <= delta, d_test == 0), d_model > 0)


def vsd(depth_est, depth_gt, depth_test, K, delta, taus, diameter, normalized_by_diameter=True):
    """Visible Surface Discrepancy for every tau (pose_error.py:17-101, cost_type "step") from the two rendered depth maps."""
    dist_test, dist_gt, dist_est = depth_to_dist(depth_test, K), depth_to_dist(depth_gt, K), depth_to_dist(depth_est, K)
    visib_gt = _visib_mask(dist_test, dist_gt, delta)
    visib_est = np.logical_or(_visib_mask(dist_test, dist_est, delta), np.logical_and(visib_gt, dist_est > 0))
    inter, union = np.logical_and(visib_gt, visib_est), np.logical_or(visib_gt, visib_est)
    n_union = union.sum()
    n_comp = n_union - inter.sum()
    dists = np.abs(dist_gt[inter] - dist_est[inter])
    if normalized_by_diameter:
        dists /= diameter
    if n_union == 0:
        return [1.0] * len(taus)
    return [float((np.sum(dists >= tau) + n_comp) / float(n_union)) for tau in taus]


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


def average_recall(results, scene_gt, models, cameras, im_width, n_top=1, renderer=None, depth_images=None, vsd_delta=VSD_DELTA):
    """results: `read_results` rows; scene_gt[scene_id][im_id] = list of {"obj_id", "R" (3,3), "t" (3,) mm, optional "valid"};
    models[obj_id] = {"pts" (n,3) mm, "diameter", "symmetries": [{"R","t"}] incl. identity}; cameras[scene_id][im_id] = K.
    Only the `n_top` best-scored estimates per (image, object) take part (BOP: the instance count of the target).
    With `renderer` (render_object(obj_id, R, t, fx, fy, cx, cy) -> {"depth"}) and depth_images[scene_id][im_id] (mm, (H,W)) the VSD
    errors are computed too and "AR" = mean(AR_VSD, AR_MSSD, AR_MSPD) is the BOP'19 average recall; else AR_VSD = AR = None.
    -> {"AR_VSD", "AR_MSSD", "AR_MSPD", "AR", "AR_MSSD_MSPD", "recalls_vsd" [tau][threshold], "recalls_mssd", "recalls_mspd"}."""
    by_im = {}
    for r in results:
        by_im.setdefault((r["scene_id"], r["im_id"]), []).append(r)
    do_vsd = renderer is not None and depth_images is not None
    sets = {"mssd": [], "mspd": []}
    vsd_sets = [[] for _ in VSD_TAUS]
    gt_depth = {}
    for sid, ims in scene_gt.items():
        for iid, gts in ims.items():
            gts = [dict(g, valid=g.get("valid", True)) for g in gts]
            per_obj = {}
            for r in by_im.get((sid, iid), []):
                per_obj.setdefault(r["obj_id"], []).append(r)
            ests = {"mssd": [], "mspd": []}
            vests = [[] for _ in VSD_TAUS]
            K = np.asarray(cameras[sid][iid], np.float64)
            for obj_id, rows in per_obj.items():
                rows = sorted(rows, key=lambda r: r["score"], reverse=True)[:n_top if n_top > 0 else None]
                m = models[obj_id]
                for r in rows:
                    e1, e2, e3 = {}, {}, {}
                    d_est = renderer.render_object(obj_id, r["R"], r["t"], K[0, 0], K[1, 1], K[0, 2], K[1, 2])["depth"] if do_vsd else None
                    for gid, g in enumerate(gts):
                        if g["obj_id"] != obj_id:
                            continue
                        Rg, tg = np.asarray(g["R"], np.float64), np.asarray(g["t"], np.float64)
                        e1[gid] = mssd(r["R"], r["t"], Rg, tg, m["pts"], m["symmetries"]) / m["diameter"]
                        e2[gid] = mspd(r["R"], r["t"], Rg, tg, K, m["pts"], m["symmetries"]) * (640.0 / im_width)
                        if do_vsd:
                            key = (sid, iid, gid)
                            if key not in gt_depth:
                                gt_depth[key] = renderer.render_object(obj_id, Rg, tg, K[0, 0], K[1, 1], K[0, 2], K[1, 2])["depth"]
                            e3[gid] = vsd(d_est, gt_depth[key], depth_images[sid][iid], K, vsd_delta, VSD_TAUS, m["diameter"])
                    ests["mssd"].append(dict(score=r["score"], errors=e1))
                    ests["mspd"].append(dict(score=r["score"], errors=e2))
                    for ti in range(len(VSD_TAUS)):
                        vests[ti].append(dict(score=r["score"], errors={gid: e[ti] for gid, e in e3.items()}))
            for k in sets:
                sets[k].append((gts, ests[k]))
            for ti in range(len(VSD_TAUS)):
                vsd_sets[ti].append((gts, vests[ti]))
    rec_s = [_recall_at(sets["mssd"], th) for th in MSSD_THRESHOLDS]
    rec_p = [_recall_at(sets["mspd"], th) for th in MSPD_THRESHOLDS]
    ar_s, ar_p = float(np.mean(rec_s)), float(np.mean(rec_p))
    out = dict(AR_MSSD=ar_s, AR_MSPD=ar_p, AR_MSSD_MSPD=0.5 * (ar_s + ar_p), recalls_mssd=rec_s, recalls_mspd=rec_p, AR_VSD=None, AR=None,
               recalls_vsd=None)
    if do_vsd:
        rec_v = [[_recall_at(vsd_sets[ti], th) for th in VSD_THRESHOLDS] for ti in range(len(VSD_TAUS))]
        ar_v = float(np.mean(rec_v))
        out.update(recalls_vsd=rec_v, AR_VSD=ar_v, AR=float(np.mean([ar_v, ar_s, ar_p])))
    return out
