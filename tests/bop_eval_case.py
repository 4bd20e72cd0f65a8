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


def icosphere():
    """42-vertex / 80-face unit sphere (an icosahedron subdivided once)."""
    p = (1 + 5 ** 0.5) / 2
    v = [(-1, p, 0), (1, p, 0), (-1, -p, 0), (1, -p, 0), (0, -1, p), (0, 1, p), (0, -1, -p), (0, 1, -p), (p, 0, -1), (p, 0, 1), (-p, 0, -1), (-p, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2),
         (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.asarray(x, np.float64) / np.linalg.norm(x) for x in v]
    cache, out = {}, []

    def mid(a, b):
        k = (min(a, b), max(a, b))
        if k not in cache:
            m = v[a] + v[b]
            v.append(m / np.linalg.norm(m))
            cache[k] = len(v) - 1
        return cache[k]

    for a, b, c in f:
        ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
        out += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
    return np.stack(v), np.asarray(out, np.int32)


def make_vsd_case(seed=11):
    """A synthetic problem WITH meshes and test depth images (mm), for the VSD error and the full BOP'19 average recall: three
    ellipsoid-like objects (one with a two-fold symmetry), 2 scenes x 3 small images; the test depth of an image is the z-buffer of
    its ground-truth objects over an empty (0) or planar background, with an occluder slab in front of one object, 1 mm noise and
    dropped pixels.  -> models, scene_gt, cameras, results, im_width, depth_images, (W, H)."""
    from raster_np import render_depth

    rs = np.random.RandomState(seed)
    sv, sf = icosphere()
    W, H = 240, 180
    K = np.array([[260.0, 0, 118.5], [0, 262.0, 91.0], [0, 0, 1.0]])
    models = {}
    for obj_id, axes in ((1, (35.0, 50.0, 65.0)), (2, (60.0, 60.0, 30.0)), (5, (25.0, 40.0, 40.0))):
        verts = sv * np.asarray(axes)
        if obj_id != 2:
            verts = verts + 0.15 * np.asarray(axes) * np.sin(3.0 * sv[:, [1, 2, 0]])  # break the symmetries
        syms = [dict(R=np.eye(3), t=np.zeros(3))]
        if obj_id == 2:
            syms.append(dict(R=rot([0, 0, 1], np.pi), t=np.zeros(3)))
        d = np.linalg.norm(verts[:, None] - verts[None], axis=2).max()
        models[obj_id] = dict(pts=verts, verts=verts, faces=sf, diameter=float(d), symmetries=syms)
    scene_gt, cameras, depth_images, results = {}, {}, {}, []
    for sid in (3, 4):
        scene_gt[sid], cameras[sid], depth_images[sid] = {}, {}, {}
        for iid in (0, 5, 9):
            cameras[sid][iid] = K
            gts = []
            z = np.full((H, W), np.inf)
            for k, obj_id in enumerate([1, 2, 5] if iid != 5 else [2, 5]):
                R = rot(rs.randn(3), rs.rand() * 3)
                t = np.array([(k - 1) * 130.0 + rs.uniform(-20, 20), rs.uniform(-60, 60), rs.uniform(550, 800)])
                gts.append(dict(obj_id=obj_id, R=R, t=t, valid=not (sid == 4 and iid == 9 and obj_id == 5)))
                d = render_depth(models[obj_id]["verts"], sf, R, t, K[0, 0], K[1, 1], K[0, 2], K[1, 2], H, W).astype(np.float64)
                z = np.where((d > 0) & (d < z), d, z)
                if rs.rand() < 0.1:
                    continue
                level = rs.choice([0.0, 0.01, 0.03, 0.08, 0.3])
                Re = R @ rot(rs.randn(3), level * 2.0)
                if obj_id == 2 and rs.rand() < 0.5:
                    Re = Re @ models[2]["symmetries"][1]["R"]
                te = t + rs.randn(3) * level * 150
                results.append(dict(scene_id=sid, im_id=iid, obj_id=obj_id, score=float(rs.rand()), R=Re, t=te, time=0.1))
                if rs.rand() < 0.3:
                    results.append(dict(scene_id=sid, im_id=iid, obj_id=obj_id, score=float(rs.rand()) * 0.5,
                                        R=R @ rot(rs.randn(3), 0.02), t=t + rs.randn(3) * 2, time=0.1))
            back = 1400.0 if sid == 4 else 0.0
            z = np.where(np.isinf(z), back, z)
            if iid == 9:  # an occluder slab 120 mm in front of the first object, over the left part of its silhouette
                u0 = int(K[0, 0] * gts[0]["t"][0] / gts[0]["t"][2] + K[0, 2])
                z[:, max(0, u0 - 40):max(0, u0)] = gts[0]["t"][2] - 120.0
            z = np.where(z > 0, z + rs.randn(H, W), 0.0)
            z[rs.rand(H, W) < 0.02] = 0.0
            scene_gt[sid][iid] = gts
            depth_images[sid][iid] = z.astype(np.float32)
    return models, scene_gt, cameras, results, float(W), depth_images, (W, H)
