"""A tiny synthetic MegaPose-format training tree (two subsets, web-dataset shards unpacked to files) written to a directory: the
input of the training-provider tests and of tests/golden/make_provider_train_golden.py.  Deterministic (own RandomState, Pillow
writers).  Layout = what core/unopose/provider/pfoneref_training_dataset_v2.py:95-147 reads."""
import json
import os
import os.path as osp

import numpy as np

H, W = 90, 120
CFG = dict(img_size=56, n_sample_observed_point=192, n_sample_model_point=192, n_sample_template_point=300, min_px_count_visib=512,
           min_visib_fract=0.1, dilate_mask=True, rgb_mask_flag=True, shift_range=0.01, rgb_to_bgr=False)
SUBSETS = (("MegaPose-GSO", "megapose_gso_fixed", "gso_models.json"), ("MegaPose-ShapeNetCore", "megapose_shapenetcore_fixed", "shapenet_models.json"))


def _rle_list(mask):
    """Uncompressed COCO run lengths of a bool mask (column-major, first run counts zeros)."""
    flat = np.asarray(mask, bool).T.reshape(-1)
    edges = np.flatnonzero(np.diff(flat.astype(np.int8))) + 1
    counts = np.diff(np.concatenate([[0], edges, [flat.size]])).tolist()
    return {"size": [int(mask.shape[0]), int(mask.shape[1])], "counts": ([0] + counts) if flat[0] else counts}


def _view(folder, key, objs, rs):
    """objs: [(obj_id, (cy, cx, ry, rx), z_mm)] -- one elliptic blob per instance on a tilted depth plane."""
    from PIL import Image

    os.makedirs(folder, exist_ok=True)
    head = osp.join(folder, key)
    yy, xx = np.mgrid[0:H, 0:W]
    rgb = rs.randint(0, 256, size=(H, W, 3)).astype(np.uint8)
    depth = np.full((H, W), 4000, np.uint16)
    masks, gts, infos = {}, [], []
    for j, (obj_id, (cy, cx, ry, rx), z_mm) in enumerate(objs):
        m = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
        d = z_mm + 1.5 * (xx - cx) + 0.8 * (yy - cy) + rs.randint(-2, 3, size=(H, W))
        depth[m] = d[m].astype(np.uint16)
        masks[str(j)] = _rle_list(m)
        ang = 0.4 * obj_id + 0.07 * len(key) + 0.3 * j
        c, s = np.cos(ang), np.sin(ang)
        gts.append({"cam_R_m2c": [c, -s, 0, s * 0.8, c * 0.8, -0.6, s * 0.6, c * 0.6, 0.8], "cam_t_m2c": [12.0 * obj_id, -7.0, float(z_mm)], "obj_id": obj_id})
        infos.append({"bbox_visib": [int(cx - rx), int(cy - ry), int(2 * rx), int(2 * ry)], "visib_fract": 0.9, "px_count_visib": int(m.sum())})
    Image.fromarray(rgb).save(head + ".rgb.jpg", quality=92)
    Image.fromarray(depth).save(head + ".depth.png")
    json.dump(masks, open(head + ".mask_visib.json", "w"))
    json.dump(gts, open(head + ".gt.json", "w"))
    json.dump(infos, open(head + ".gt_info.json", "w"))
    json.dump({"cam_K": [105.0, 0, 60.5, 0, 108.0, 44.5, 0, 0, 1], "depth_scale": 0.25}, open(head + ".camera.json", "w"))


def build(root):
    """Writes the tree under `root`/MegaPose-Training-Data and returns the cfg dict (data_dir included)."""
    rs = np.random.RandomState(4321)
    data_dir = osp.join(root, "MegaPose-Training-Data")
    for si, (sub, prefix, models) in enumerate(SUBSETS):
        base = osp.join(data_dir, sub, "train_pbr_web")
        key_to_shard, valid, refs = {}, {}, {}
        for shard in range(2):
            for v in range(3):
                key = f"{10 * si + shard:06d}_{v:06d}"
                objs = [(1 + (v + shard) % 3, (30 + 8 * v, 35 + 5 * shard, 14 + v, 18 - 2 * v), 900 + 40 * v),
                        (1 + (v + shard + 1) % 3, (60 - 3 * v, 85, 16, 12 + 2 * v), 1100 - 30 * v)]
                if v == 2 and shard == 1:
                    objs.append((3, (8, 8, 2, 2), 700))  # a 13-pixel instance: too few points after filtering
                _view(osp.join(base, f"{shard:06d}"), key, objs, rs)
                key_to_shard[key] = shard
                valid[f"{shard:06d}/{key}"] = list(range(len(objs))) if not (v == 1 and shard == 0) else []  # one view without valid instances
                for j, o in enumerate(objs[:2]):
                    refs.setdefault(str(o[0]), []).append([shard, key, j])
        # a key listed in the shard index whose files are missing (an incomplete download): the provider must skip it
        key_to_shard[f"{10 * si:06d}_{9:06d}"] = 0
        valid[f"{0:06d}/{10 * si:06d}_{9:06d}"] = [0]
        json.dump(key_to_shard, open(osp.join(base, "key_to_shard.json"), "w"))
        json.dump([{"obj_id": i} for i in range(1, 4)], open(osp.join(base, models), "w"))
        json.dump(refs, open(osp.join(data_dir, prefix + "_obj_id_to_visib0_8_scene_im_inst_ids.json"), "w"))
        json.dump(valid, open(osp.join(data_dir, prefix + "_valid_inst_ids.json"), "w"))
    return dict(CFG, data_dir=data_dir)
