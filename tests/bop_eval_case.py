"""A synthetic BOP scoring problem shared by tests/golden/make_bop_eval_golden.py (bop_toolkit_lib on it) and the tests."""
import numpy as np


def rot(axis, ang):
    axis = np.asarray(axis, np.float64) / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)


def make_case(seed=4):
    rs = np.random.RandomState(seed)
    models = {}
    for obj_id in (1, 2, 5):
        pts = rs.randn(150, 3) * np.array([30.0, 45.0, 60.0])
        syms = [dict(R=np.eye(3), t=np.zeros(3))]
        if obj_id == 2:  # a two-fold symmetry about z
            syms.append(dict(R=rot([0, 0, 1], np.pi), t=np.zeros(3)))
            pts = np.concatenate([pts, pts @ syms[1]["R"].T])
        d = np.linalg.norm(pts[:, None] - pts[None], axis=2).max()
        models[obj_id] = dict(pts=pts, diameter=float(d), symmetries=syms)
    K = np.array([[572.4, 0, 325.3], [0, 573.6, 242.0], [0, 0, 1.0]])
    scene_gt, cameras, results = {}, {}, []
    for sid in (48, 49):
        scene_gt[sid], cameras[sid] = {}, {}
        for iid in (1, 7, 12):
            cameras[sid][iid] = K
            gts = []
            for obj_id in ([1, 2, 5] if iid != 7 else [2, 5]):
                R = rot(rs.randn(3), rs.rand() * 3)
                t = np.array([rs.uniform(-150, 150), rs.uniform(-100, 100), rs.uniform(600, 1100)])
                gts.append(dict(obj_id=obj_id, R=R, t=t, valid=not (sid == 49 and iid == 12 and obj_id == 5)))
                # estimates: a graded error (none .. gross), sometimes a second lower-scored one, sometimes missing
                if rs.rand() < 0.12:
                    continue
                level = rs.choice([0.0, 0.01, 0.03, 0.08, 0.3])
                Re = R @ rot(rs.randn(3), level * 2.0)
                if obj_id == 2 and rs.rand() < 0.5:
                    Re = Re @ models[2]["symmetries"][1]["R"]  # the symmetric twin is as good as the pose itself
                te = t + rs.randn(3) * level * 200
                results.append(dict(scene_id=sid, im_id=iid, obj_id=obj_id, score=float(rs.rand()), R=Re, t=te, time=0.1))
                if rs.rand() < 0.3:
                    results.append(dict(scene_id=sid, im_id=iid, obj_id=obj_id, score=float(rs.rand()) * 0.5,
                                        R=R @ rot(rs.randn(3), 0.02), t=t + rs.randn(3) * 2, time=0.1))
            scene_gt[sid][iid] = gts
    return models, scene_gt, cameras, results, 640 * 1.5
