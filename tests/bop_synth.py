"""A tiny synthetic BOP-format dataset (ycbv layout) written to a directory: the input of the provider
tests and of tests/golden/make_provider_golden.py.  Deterministic (own RandomState, Pillow PNG writer)."""
import json
import os
import os.path as osp

import numpy as np

from unopose_amd.provider import rle_counts_to_string, rle_encode

H, W = 96, 128
CFG = dict(ref_targets_name="test_ref_targets_crossscene_rot50.json", rgb_mask_flag=True, img_size=56,
           n_sample_observed_point=256, n_sample_template_point=400, minimum_n_point=8, seg_filter_score=0.25,
           obj_idxs=None)


def _blob(cy, cx, ry, rx):
    yy, xx = np.mgrid[0:H, 0:W]
    return ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0


def _write_png(path, arr):
    from PIL import Image

    os.makedirs(osp.dirname(path), exist_ok=True)
    Image.fromarray(arr).save(path)


def _scene(root, split, scene_id, images, rs):
    folder = osp.join(root, "ycbv", split, f"{scene_id:06d}")
    cam, gt = {}, {}
    for im_id, objs in images.items():
        rgb = rs.randint(0, 256, size=(H, W, 3)).astype(np.uint8)
        depth = np.zeros((H, W), np.uint16)
        gts = []
        for j, (obj_id, (cy, cx, ry, rx), z_mm) in enumerate(objs):
            m = _blob(cy, cx, ry, rx)
            yy, xx = np.mgrid[0:H, 0:W]
            d = z_mm + 0.4 * (xx - cx) + 0.2 * (yy - cy) + rs.randint(-2, 3, size=(H, W))
            depth[m] = d[m].astype(np.uint16)
            # a few invalid-depth holes inside the object, as real sensors have
            holes = rs.rand(H, W) < 0.03
            depth[m & holes] = 0
            _write_png(osp.join(folder, "mask_visib", f"{im_id:06d}_{j:06d}.png"), (m * 255).astype(np.uint8))
            ang = 0.3 * obj_id + 0.1 * im_id
            R = [np.cos(ang), -np.sin(ang), 0, np.sin(ang), np.cos(ang), 0, 0, 0, 1]
            gts.append({"cam_R_m2c": [float(v) for v in R], "cam_t_m2c": [10.0 * obj_id, -5.0, float(z_mm)],
                        "obj_id": obj_id})
        _write_png(osp.join(folder, "rgb", f"{im_id:06d}.png"), rgb)
        _write_png(osp.join(folder, "depth", f"{im_id:06d}.png"), depth)
        cam[str(im_id)] = {"cam_K": [110.0, 0, 64.5, 0, 112.0, 47.5, 0, 0, 1], "depth_scale": 0.1 if split == "test" else 0.2}
        gt[str(im_id)] = gts
    json.dump(cam, open(osp.join(folder, "scene_camera.json"), "w"))
    json.dump(gt, open(osp.join(folder, "scene_gt.json"), "w"))


def build(root):
    """Writes the dataset under `root` and returns (cfg dict, detections path)."""
    rs = np.random.RandomState(1234)
    # test scene 48: image 1 holds objects 2 and 5 (+ a low-score detection of 5), image 2 holds object 2 at the border
    test = {1: [(2, (40, 40, 22, 17), 7000), (5, (55, 95, 14, 25), 9000)], 2: [(2, (10, 118, 16, 14), 6500)]}
    _scene(root, "test", 48, test, rs)
    # references: object 2 from train_real scene 10 image 5; object 5 from test scene 48 image 2's neighbour scene 49
    _scene(root, "train_real", 10, {5: [(2, (50, 60, 30, 24), 3600)]}, rs)
    _scene(root, "test", 49, {7: [(5, (45, 64, 20, 33), 8000), (2, (80, 20, 8, 8), 7000)]}, rs)
    targets = [dict(scene_id=48, im_id=1, obj_id=2, ref_scene_id=10, ref_im_id=5),
               dict(scene_id=48, im_id=1, obj_id=5, ref_scene_id=49, ref_im_id=7),
               dict(scene_id=48, im_id=2, obj_id=2, ref_scene_id=10, ref_im_id=5)]
    json.dump(targets, open(osp.join(root, "ycbv", CFG["ref_targets_name"]), "w"))
    dets = []
    for im_id, objs in test.items():
        for j, (obj_id, (cy, cx, ry, rx), _) in enumerate(objs):
            m = _blob(cy + 1, cx - 1, ry + 2, rx + 1)  # detector masks are not the GT masks
            seg = rle_encode(m)
            if j == 1:  # compressed-string flavour of COCO RLE
                seg = {"size": seg["size"], "counts": rle_counts_to_string(seg["counts"])}
            ys, xs = np.nonzero(m)
            dets.append(dict(scene_id=48, image_id=im_id, category_id=obj_id, score=0.9 - 0.2 * j, time=0.25,
                             bbox=[int(xs.min()), int(ys.min()), int(np.ptp(xs)) + 1, int(np.ptp(ys)) + 1], segmentation=seg))
    # a duplicate, low-score detection that the score filter must drop
    dets.append(dict(dets[1], score=0.1))
    det_path = osp.join(root, "detections.json")
    json.dump(dets, open(det_path, "w"))
    cfg = dict(CFG, data_dir=root)
    return cfg, det_path
