"""BOP test-set data provider for the one-reference setting: the producer side of UNOPose.forward's
input dict.

Host-side mirror of ``BOPTestsetPoseFreeOneRefv2`` (core/unopose/provider/pfoneref_bop_test_dataset_v2.py:33-354)
and of the helpers it takes from core/unopose/utils/data_utils.py (``get_bbox`` :249-283, ``backproject``
:216-229, ``get_resize_rgb_choose`` :232-246, ``get_bop_depth_map`` :339-351, ``get_bop_image`` :404-430,
``rle_to_binary_mask`` :168-185).  Same constructor fields, same per-instance dict (keys, dtypes, shapes),
same order of ``np.random`` draws, so a seeded run reproduces the reference's samples.

Third-party pieces the reference calls that this image lacks are restated here:
  * pycocotools ``frPyObjects`` / ``decode``  -> :func:`rle_decode` (COCO RLE, compressed or not);
  * ``cv2.resize(..., INTER_LINEAR)`` on uint8 -> :func:`resize_bilinear_u8` (OpenCV's 11-bit fixed-point
    bilinear, incl. its exact-2x shortcut).  PARITY UNPINNED: cv2 is not importable in the build
    container, the restatement follows OpenCV's published algorithm (modules/imgproc/src/resize.cpp);
  * torchvision ``ToTensor`` + ``Normalize`` -> :func:`to_tensor_normalize`;
  * imageio -> Pillow.
"""
import json
import os
import os.path as osp

import numpy as np
import torch

# obj_id -> 0-based label: the key order of ref/<dataset>.py `id2obj` (pfoneref_bop_test_dataset_v2.py:68-73)
DATASET_OBJ_IDS = {
    "ycbv": list(range(1, 22)),
    "lm": list(range(1, 16)),
    "lmo": [1, 5, 6, 8, 9, 10, 11, 12],
    "tudl": [1, 2, 3],
    "tyol": list(range(1, 22)),
    "hb": list(range(1, 34)),
    "hb_bop19": [1, 3, 4, 8, 9, 10, 12, 15, 17, 18, 19, 22, 23, 29, 32, 33],
}
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def load_json(path):
    with open(path, "rb") as f:
        return json.loads(f.read())


def read_image(path):
    """ndarray of the file's own dtype (uint8 RGB / gray, uint16 depth), like imageio.imread."""
    from PIL import Image

    with Image.open(path) as im:
        return np.array(im)


# ------------------------------------------------------------------------------------------------
# COCO run-length masks (pycocotools maskApi.c: rleFrString / rleDecode; data_utils.py:168-185)
# ------------------------------------------------------------------------------------------------
def rle_counts_from_string(s):
    """The LEB128-like, delta-coded count string of a compressed COCO RLE -> list of run lengths."""
    if isinstance(s, str):
        s = s.encode("ascii")
    counts, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = s[p] - 48
            x |= (c & 0x1F) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    return counts


def rle_counts_to_string(counts):
    """Inverse of :func:`rle_counts_from_string` (maskApi.c rleToString); used to build test inputs."""
    out = bytearray()
    for i, x in enumerate(counts):
        x = int(x)
        if i > 2:
            x -= int(counts[i - 2])
        more = True
        while more:
            c = x & 0x1F
            x >>= 5
            more = (x != -1) if (c & 0x10) else (x != 0)
            if more:
                c |= 0x20
            out.append(c + 48)
    return out.decode("ascii")


def rle_decode(seg):
    """{"size": [h, w], "counts": list | str} -> (h, w) uint8 mask.  Runs alternate 0s / 1s starting with
    0s, in column-major (Fortran) order."""
    h, w = seg["size"]
    counts = seg["counts"]
    if isinstance(counts, (str, bytes)):
        counts = rle_counts_from_string(counts)
    flat = np.zeros(h * w, dtype=np.uint8)
    pos = 0
    for i, c in enumerate(counts):
        if i & 1:
            flat[pos:pos + c] = 1
        pos += c
    return flat.reshape((h, w), order="F")


def rle_encode(mask):
    """(h, w) mask -> uncompressed COCO RLE dict (test-input helper)."""
    flat = np.asarray(mask, dtype=np.uint8).flatten(order="F")
    change = np.flatnonzero(np.diff(flat)) + 1
    bounds = np.concatenate([[0], change, [flat.size]])
    counts = np.diff(bounds).tolist()
    if flat.size and flat[0]:
        counts = [0] + counts
    return {"size": [int(mask.shape[0]), int(mask.shape[1])], "counts": counts}


# ------------------------------------------------------------------------------------------------
# geometry helpers (data_utils.py)
# ------------------------------------------------------------------------------------------------
def get_bbox(label):
    """Square box around the mask, side = min(max(h_box, w_box), min(H, W)), centred on the mask's box and
    shifted back inside the image (data_utils.py:249-283).  Returns [y1, y2, x1, x2]."""
    img_width, img_length = label.shape
    rows = np.any(label, axis=1)
    cols = np.any(label, axis=0)
    rmin, rmax = np.where(rows)[0][[0, -1]]
    cmin, cmax = np.where(cols)[0][[0, -1]]
    rmax += 1
    cmax += 1
    b = min(max(rmax - rmin, cmax - cmin), min(img_width, img_length))
    center = [int((rmin + rmax) / 2), int((cmin + cmax) / 2)]
    rmin, rmax = center[0] - int(b / 2), center[0] + int(b / 2)
    cmin, cmax = center[1] - int(b / 2), center[1] + int(b / 2)
    if rmin < 0:
        rmax += -rmin
        rmin = 0
    if cmin < 0:
        cmax += -cmin
        cmin = 0
    if rmax > img_width:
        rmin -= rmax - img_width
        rmax = img_width
    if cmax > img_length:
        cmin -= cmax - img_length
        cmax = img_length
    return [rmin, rmax, cmin, cmax]


def backproject(depth, K, bbox=None):
    """Organised cloud (H, W, 3) of a depth map, optionally cropped (data_utils.py:216-229)."""
    H, W = depth.shape
    X, Y = np.meshgrid(np.asarray(range(W)) - K[0, 2], np.asarray(range(H)) - K[1, 2])
    cloud = np.stack((X * depth / K[0, 0], Y * depth / K[1, 1], depth), axis=2)
    if bbox is not None:
        rmin, rmax, cmin, cmax = bbox
        return cloud[rmin:rmax, cmin:cmax]
    return cloud


def get_resize_rgb_choose(choose, bbox, img_size):
    """Flat index into the crop -> flat index into the img_size x img_size resized crop
    (data_utils.py:232-246; the crop is square, and the reference splits rows AND columns by crop_h)."""
    rmin, rmax, cmin, cmax = bbox
    crop_h = rmax - rmin
    ratio_h = img_size / crop_h
    crop_w = cmax - cmin
    ratio_w = img_size / crop_w
    row_idx = choose // crop_h
    col_idx = choose % crop_h
    return (np.floor(row_idx * ratio_w) * img_size + np.floor(col_idx * ratio_h)).astype(np.int64)


# ------------------------------------------------------------------------------------------------
# image helpers
# ------------------------------------------------------------------------------------------------
def _linear_taps(dst, src):
    """OpenCV's per-axis source index and 11-bit weights for INTER_LINEAR."""
    scale = float(src) / float(dst)  # double, as 1 / inv_scale
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    return s, f


def resize_bilinear_u8(img, size):
    """cv2.resize(img, (size, size), interpolation=cv2.INTER_LINEAR) for uint8 HxWxC images.

    OpenCV computes 8-bit bilinear resizes in fixed point: horizontal pass into int32 with weights
    round(w * 2048) (borders: a source index left of 0 or at/after the last column collapses to that
    single pixel), vertical pass ``((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2`` with
    the source rows clamped; an exact 2x shrink in both axes takes the area shortcut
    ``(a + b + c + d + 2) >> 2``."""
    img = np.ascontiguousarray(img)
    assert img.dtype == np.uint8 and img.ndim in (2, 3)
    squeeze = img.ndim == 2
    if squeeze:
        img = img[:, :, None]
    sh, sw = img.shape[:2]
    dh = dw = int(size)
    if sh == dh and sw == dw:
        out = img.copy()
    elif sh == 2 * dh and sw == 2 * dw:
        a = img.astype(np.int32)
        out = ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    else:
        sx, fx = _linear_taps(dw, sw)
        lo, hi = sx < 0, sx >= sw - 1
        fx = np.where(lo | hi, np.float32(0), fx)
        sx = np.where(lo, 0, np.where(hi, sw - 1, sx))
        a0 = np.rint((np.float32(1) - fx) * np.float32(2048)).astype(np.int64)
        a1 = np.rint(fx * np.float32(2048)).astype(np.int64)
        sx1 = np.minimum(sx + 1, sw - 1)
        sy, fy = _linear_taps(dh, sh)
        b0 = np.rint((np.float32(1) - fy) * np.float32(2048)).astype(np.int64)
        b1 = np.rint(fy * np.float32(2048)).astype(np.int64)
        sy0 = np.clip(sy, 0, sh - 1)
        sy1 = np.clip(sy + 1, 0, sh - 1)
        src = img.astype(np.int64)
        rows = src[:, sx, :] * a0[None, :, None] + src[:, sx1, :] * a1[None, :, None]  # (sh, dw, C) x 2048
        r0, r1 = rows[sy0], rows[sy1]
        out = ((((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2)
        out = np.clip(out, 0, 255).astype(np.uint8)
    return out[:, :, 0] if squeeze else out


def to_tensor_normalize(rgb_u8, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """torchvision ToTensor (HWC uint8 -> CHW float32 / 255) followed by Normalize(mean, std)."""
    t = torch.from_numpy(np.ascontiguousarray(rgb_u8.transpose(2, 0, 1))).to(torch.float32).div(255)
    m = torch.as_tensor(mean, dtype=torch.float32).view(-1, 1, 1)
    s = torch.as_tensor(std, dtype=torch.float32).view(-1, 1, 1)
    return t.sub_(m).div_(s)


def get_bop_depth_map(inst):
    """Depth in metres before `depth_scale` (data_utils.py:339-351): png first, tif as the fallback."""
    folder = osp.join(inst["data_folder"], f"{inst['scene_id']:06d}", "depth")
    png = osp.join(folder, f"{inst['img_id']:06d}.png")
    path = png if osp.exists(png) else osp.join(folder, f"{inst['img_id']:06d}.tif")
    return read_image(path) / 1000.0


def get_bop_image(inst, bbox, img_size, mask=None, rgb_to_bgr=False):
    """Crop -> mask -> bilinear resize of the colour (or gray) image (data_utils.py:404-430)."""
    rmin, rmax, cmin, cmax = bbox
    folder = osp.join(inst["data_folder"], f"{inst['scene_id']:06d}")
    img_path = folder + "/"
    for s in (f"rgb/{inst['img_id']:06d}.jpg", f"rgb/{inst['img_id']:06d}.png", f"gray/{inst['img_id']:06d}.tif"):
        if osp.exists(osp.join(folder, s)):
            img_path = osp.join(folder, s)
            break
    rgb = read_image(img_path).astype(np.uint8)
    if rgb.ndim == 2:
        rgb = np.concatenate([rgb[:, :, None]] * 3, axis=2)
    rgb = rgb[..., ::-1][rmin:rmax, cmin:cmax, :3] if rgb_to_bgr else rgb[rmin:rmax, cmin:cmax, :3]
    if mask is not None:
        rgb = rgb * (mask[:, :, None] > 0).astype(np.uint8)
    return resize_bilinear_u8(rgb, img_size)


# ------------------------------------------------------------------------------------------------
class BOPTestsetOneRef:
    """One item = one test image: all detections above `seg_filter_score` (or the best one), each paired
    with its reference view from `ref_targets_name` (pfoneref_bop_test_dataset_v2.py:33-354).

    cfg fields (attribute or key access): data_dir, ref_targets_name, rgb_mask_flag, img_size,
    n_sample_observed_point, n_sample_template_point, minimum_n_point, seg_filter_score; optional
    rgb_to_bgr, obj_idxs, oneref_percat (+ targets_name, ref_scene_ims)."""

    def __init__(self, cfg, eval_dataset_name="lmo", detetion_path=None):
        assert detetion_path is not None
        get = (lambda k, d=None: cfg.get(k, d)) if hasattr(cfg, "get") else (lambda k, d=None: getattr(cfg, k, d))
        self.cfg = cfg
        self.dataset = eval_dataset_name
        self.data_dir = get("data_dir")
        self.ref_targets_name = get("ref_targets_name")
        self.rgb_mask_flag = get("rgb_mask_flag")
        self.img_size = get("img_size")
        self.n_sample_observed_point = get("n_sample_observed_point")
        self.n_sample_template_point = get("n_sample_template_point")
        self.minimum_n_point = get("minimum_n_point")
        self.seg_filter_score = get("seg_filter_score")
        self.rgb_to_bgr = get("rgb_to_bgr", False)
        obj_idxs = get("obj_idxs", None)
        if obj_idxs is None:
            obj_idxs = {obj_id: i for i, obj_id in enumerate(DATASET_OBJ_IDS[eval_dataset_name])}
        self.obj_idxs = obj_idxs
        self.data_folder = osp.join(self.data_dir, eval_dataset_name, "test")
        if get("oneref_percat", False):
            assert get("targets_name") is not None
            self.test_ref_target = self.load_single_ref_per_dset(
                osp.join(self.data_dir, eval_dataset_name, get("targets_name")), get("ref_scene_ims"))
        else:
            self.test_ref_target = self.load_ref(osp.join(self.data_dir, eval_dataset_name, self.ref_targets_name))
        self.det_keys, self.dets = [], {}
        for det in load_json(detetion_path):
            key = str(det["scene_id"]).zfill(6) + "_" + str(det["image_id"]).zfill(6)
            if key not in self.dets:
                self.det_keys.append(key)
                self.dets[key] = []
            self.dets[key].append(det)

    def __len__(self):
        return len(self.det_keys)

    def __getitem__(self, index):
        dets = self.dets[self.det_keys[index]]
        instances, inst_ids = [], []
        for det_i, det in enumerate(dets):
            if det["score"] > self.seg_filter_score:
                instance = self.get_instance(det)
                if instance is not None:
                    instances.append(instance)
                    inst_ids.append(det_i)
        if len(instances) == 0:  # keep at least the best detection (:113-124)
            scores = [det["score"] for det in dets]
            best = scores.index(max(scores))
            instance = self.get_instance(dets[best])
            if instance is None:
                raise ValueError(f"no qulified instance in {self.det_keys[index]}")
            instances.append(instance)
            inst_ids.append(best)
        ret = {}
        for key in instances[0].keys():
            if key == "ref_key":  # this provider's addition: identifies the reference view (runner.ReferenceCache)
                ret["ref_keys"] = [inst[key] for inst in instances]
            else:
                ret[key] = torch.stack([inst[key] for inst in instances])
        ret["scene_id"] = torch.IntTensor([int(self.det_keys[index][0:6])])
        ret["img_id"] = torch.IntTensor([int(self.det_keys[index][7:13])])
        ret["inst_ids"] = torch.IntTensor(inst_ids)
        ret["seg_time"] = torch.FloatTensor([dets[0]["time"]])
        return ret

    def get_instance(self, data):
        scene_id, img_id, obj_id = data["scene_id"], data["image_id"], data["category_id"]
        seg, score = data["segmentation"], data["score"]
        scene_camera = load_json(osp.join(self.data_folder, f"{scene_id:06d}", "scene_camera.json"))
        K = np.array(scene_camera[str(img_id)]["cam_K"]).reshape((3, 3)).copy()
        depth_scale = scene_camera[str(img_id)]["depth_scale"]
        inst = dict(scene_id=scene_id, img_id=img_id, data_folder=self.data_folder)
        obj_idx = self.obj_idxs[obj_id]
        depth = get_bop_depth_map(inst) * depth_scale
        mask = np.logical_and(rle_decode(seg) > 0, depth > 0)  # segmentation restricted to valid depth
        if not np.sum(mask) > self.minimum_n_point:
            return None
        y1, y2, x1, x2 = get_bbox(mask)
        mask = mask[y1:y2, x1:x2]
        choose = mask.astype(np.float32).flatten().nonzero()[0]
        cloud = backproject(depth, K, [y1, y2, x1, x2]).reshape(-1, 3)[choose, :]
        tmp_cloud = cloud - np.mean(cloud, axis=0)[None, :]

        tem_rgb, tem_choose, tem_pts, pose_camref_obj, ref_key = self._get_ref_instance(scene_id, img_id, obj_id)
        if tem_rgb is None:
            return None
        # outlier filter: keep points within 1.2 reference radii of the observed centroid (:187-198)
        radius = np.max(np.linalg.norm(tem_pts - np.mean(tem_pts, axis=0).reshape(1, 3), axis=1))
        flag = np.linalg.norm(tmp_cloud, axis=1) < 1.2 * radius
        if np.sum(flag) < self.minimum_n_point:
            return None
        choose, cloud = choose[flag], cloud[flag]
        n = self.n_sample_observed_point
        choose_idx = np.random.choice(np.arange(len(choose)), size=n, replace=len(choose) <= n)
        choose, cloud = choose[choose_idx], cloud[choose_idx]

        rgb = get_bop_image(inst, [y1, y2, x1, x2], self.img_size, mask if self.rgb_mask_flag else None,
                            rgb_to_bgr=self.rgb_to_bgr)
        rgb = to_tensor_normalize(np.array(rgb))
        rgb_choose = get_resize_rgb_choose(choose, [y1, y2, x1, x2], self.img_size)
        return {
            "pts": torch.FloatTensor(cloud),
            "rgb": torch.FloatTensor(rgb),
            "rgb_choose": torch.IntTensor(rgb_choose).long(),
            "obj": torch.IntTensor([obj_idx]).long(),
            "obj_id": torch.IntTensor([obj_id]),
            "score": torch.FloatTensor([score]),
            "tem1_rgb": torch.FloatTensor(tem_rgb),
            "tem1_choose": torch.IntTensor(tem_choose).long(),
            "tem1_pts": torch.FloatTensor(tem_pts),
            "tem1_pose": torch.FloatTensor(pose_camref_obj),
            "ref_key": ref_key,
        }

    def _ref_data_folder(self, ref_scene_id):
        if self.dataset == "ycbv":  # references outside the 12 test scenes live in train_real (:246-251)
            return self.data_folder if ref_scene_id in range(48, 60) else osp.join(self.data_dir, self.dataset, "train_real")
        if self.dataset == "tudl":
            return osp.join(self.data_dir, self.dataset, "train_real")
        return self.data_folder

    def _get_ref_instance(self, scene_id, img_id, obj_id):
        none = (None, None, None, None, None)
        key = f"{scene_id}_{img_id}_{obj_id}"
        if key not in self.test_ref_target:
            return none
        ref_scene_id, ref_im_id = (int(v) for v in self.test_ref_target[key].split("_"))
        data_folder = self._ref_data_folder(ref_scene_id)
        scene_folder = osp.join(data_folder, f"{ref_scene_id:06d}")
        scene_camera = load_json(osp.join(scene_folder, "scene_camera.json"))
        K = np.array(scene_camera[str(ref_im_id)]["cam_K"]).reshape((3, 3)).copy()
        pose_camref_obj = ref_mask_path = None
        for i, gt in enumerate(load_json(osp.join(scene_folder, "scene_gt.json"))[str(ref_im_id)]):
            if gt["obj_id"] == obj_id:
                ref_mask_path = osp.join(data_folder, f"{ref_scene_id:06d}/mask_visib/{ref_im_id:06d}_{i:06d}.png")
                pose_camref_obj = np.eye(4, dtype=np.float32)
                pose_camref_obj[:3, :3] = np.array(gt["cam_R_m2c"], dtype=np.float32).reshape(3, 3)
                pose_camref_obj[:3, 3] = np.array(gt["cam_t_m2c"], dtype=np.float32).reshape(3) * 0.001
                break
        if pose_camref_obj is None:
            return none
        depth_scale = scene_camera[str(ref_im_id)]["depth_scale"]
        inst = dict(scene_id=ref_scene_id, img_id=ref_im_id, data_folder=data_folder)
        depth = (get_bop_depth_map(inst) * depth_scale).astype("float32")
        mask = np.array(read_image(ref_mask_path)).astype(bool)
        bbox = get_bbox(mask)
        y1, y2, x1, x2 = bbox
        mask = mask[y1:y2, x1:x2]
        ref_xyz = backproject(depth, K, bbox)
        ref_xyz *= mask.astype("float32")[:, :, None]
        ref_rgb = get_bop_image(inst, [y1, y2, x1, x2], self.img_size, mask if self.rgb_mask_flag else None,
                                rgb_to_bgr=self.rgb_to_bgr)
        ref_rgb = to_tensor_normalize(np.array(ref_rgb))
        ref_choose = (mask > 0).astype(np.float32).flatten().nonzero()[0]
        n = self.n_sample_template_point
        if len(ref_choose) <= n:
            choose_idx = np.random.choice(np.arange(len(ref_choose)), n)
        else:
            choose_idx = np.random.choice(np.arange(len(ref_choose)), n, replace=False)
        ref_choose = ref_choose[choose_idx]
        ref_xyz = ref_xyz.reshape(-1, 3)[ref_choose, :]
        ref_rgb_choose = get_resize_rgb_choose(ref_choose, [y1, y2, x1, x2], self.img_size)
        return ref_rgb, ref_rgb_choose, ref_xyz, pose_camref_obj, (ref_scene_id, ref_im_id, obj_id)

    @staticmethod
    def load_ref(path):
        """[{scene_id, im_id, obj_id, ref_scene_id, ref_im_id}] -> {"scene_im_obj": "refscene_refim"} (:314-331)."""
        return {f"{t['scene_id']}_{t['im_id']}_{t['obj_id']}": f"{t['ref_scene_id']}_{t['ref_im_id']}"
                for t in load_json(path)}

    def load_single_ref_per_dset(self, test_target_path, ref_scene_ims):
        """One fixed reference view per object id (:333-354)."""
        assert len(ref_scene_ims) == len(self.obj_idxs)
        ref_dict = {i + 1: tuple(int(v) for v in s.split("_")) for i, s in enumerate(ref_scene_ims)}
        out = {}
        for t in load_json(test_target_path):
            rs, ri = ref_dict[t["obj_id"]]
            out[f"{t['scene_id']}_{t['im_id']}_{t['obj_id']}"] = f"{rs}_{ri}"
        return out


def collate_image(item):
    """DataLoader(batch_size=1) view of one item: tensors get the leading image dimension the runner
    indexes with [0] (oneref_inference_utils_v1.py:36-66)."""
    return {k: (v[None] if torch.is_tensor(v) else v) for k, v in item.items()}
