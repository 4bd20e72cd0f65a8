"""BOP test-set data provider for the one-reference setting: the producer side of UNOPose.forward's
input dict.

Contract = the items of ``BOPTestsetPoseFreeOneRefv2`` (core/unopose/provider/pfoneref_bop_test_dataset_v2.py:33-354;
geometry conventions of core/unopose/utils/data_utils.py:216-283): same constructor fields, same per-image dict
(keys, dtypes, shapes), same ``np.random`` stream consumption, so a seeded run reproduces the reference's samples
bit for bit (tests/golden/make_provider_golden.py runs the reference class on the same files).

Structure is this repo's own, built around what is invariant per IMAGE and per REFERENCE VIEW instead of per
detection:
  * :class:`SceneFiles` reads every json / depth / colour file at most once (small LRU): the reference re-opens
    scene_camera.json, the depth map and the colour image for every detection of an image;
  * :class:`ReferenceViews` keeps the deterministic part of a reference view (window, masked cloud, normalised crop,
    pose) across the many query instances that share it -- only the 5000-point draw is per instance (it must be:
    it consumes the random stream);
  * :class:`Window` carries the square crop geometry (construction from a mask, cropping, window-pixel ->
    resized-crop index map) as one value object.

Third-party pieces the reference calls that this image lacks are restated here:
  * pycocotools ``frPyObjects`` / ``decode``  -> :func:`rle_decode` (COCO RLE, compressed or not);
  * ``cv2.resize(..., INTER_LINEAR)`` on uint8 -> :func:`resize_bilinear_u8` (OpenCV's 11-bit fixed-point
    bilinear, incl. its exact-2x shortcut).  PARITY UNPINNED: cv2 is not importable in the build
    container, the restatement follows OpenCV's published algorithm (modules/imgproc/src/resize.cpp);
  * torchvision ``ToTensor`` + ``Normalize`` -> :func:`to_tensor_normalize`;
  * imageio -> Pillow.
"""
import json
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
# crop geometry
# ------------------------------------------------------------------------------------------------
class Window:
    """Square pixel window [y0, y1) x [x0, x1) around a mask (the reference's crop convention,
    data_utils.py:249-283): side = min(longer extent of the mask's bounding box, shorter image side), rounded
    down to even by the half-side arithmetic, centred on the box and pushed back inside the image."""

    __slots__ = ("y0", "y1", "x0", "x1")

    def __init__(self, y0, y1, x0, x1):
        self.y0, self.y1, self.x0, self.x1 = int(y0), int(y1), int(x0), int(x1)

    @classmethod
    def around(cls, mask):
        H, W = mask.shape
        ys = np.flatnonzero(mask.any(axis=1))
        xs = np.flatnonzero(mask.any(axis=0))
        top, bottom, left, right = ys[0], ys[-1] + 1, xs[0], xs[-1] + 1  # half-open extents
        half = int(min(max(bottom - top, right - left), min(H, W)) / 2)
        cy, cx = int((top + bottom) / 2), int((left + right) / 2)
        y0, x0 = cy - half, cx - half
        y0 -= min(y0, 0)  # left / top overflow: shift right / down
        x0 -= min(x0, 0)
        y0 -= max(y0 + 2 * half - H, 0)  # bottom / right overflow: shift back
        x0 -= max(x0 + 2 * half - W, 0)
        return cls(y0, y0 + 2 * half, x0, x0 + 2 * half)

    def as_list(self):
        return [self.y0, self.y1, self.x0, self.x1]

    @property
    def side(self):
        return self.y1 - self.y0

    def crop(self, arr):
        return arr[self.y0:self.y1, self.x0:self.x1]

    def to_resized(self, flat_idx, img_size):
        """Flat index into the window (row-major) -> flat index into the img_size x img_size resized crop.
        The reference decomposes with the window HEIGHT for rows and columns alike (data_utils.py:243-244;
        windows are square) and scales by img_size / side in float64 before flooring."""
        h, w = self.y1 - self.y0, self.x1 - self.x0
        r, c = np.divmod(flat_idx, h)
        return (np.floor(r * (img_size / w)) * img_size + np.floor(c * (img_size / h))).astype(np.int64)


def lift_depth(depth, K, window=None):
    """Pinhole back-projection of a depth map to an organised (h, w, 3) float64 cloud, optionally only the
    window: X = (u - cx) d / fx, Y = (v - cy) d / fy, Z = d  (evaluation order of data_utils.py:216-229, so the
    float64 results are identical)."""
    H, W = depth.shape
    y0, y1, x0, x1 = (0, H, 0, W) if window is None else window.as_list()
    d = depth[y0:y1, x0:x1]
    u = (np.arange(x0, x1, dtype=np.int64) - K[0, 2])[None, :]
    v = (np.arange(y0, y1, dtype=np.int64) - K[1, 2])[:, None]
    return np.stack((u * d / K[0, 0], v * d / K[1, 1], d), axis=2)


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


def _normalised_crop(rgb_u8, window, img_size, mask=None, bgr=False):
    """Window of the colour image -> (masked) -> OpenCV-style bilinear resize -> ImageNet-normalised CHW tensor."""
    if rgb_u8.ndim == 2:
        rgb_u8 = np.repeat(rgb_u8[:, :, None], 3, axis=2)
    patch = window.crop(rgb_u8[..., ::-1] if bgr else rgb_u8)[:, :, :3]
    if mask is not None:
        patch = patch * (mask[:, :, None] > 0).astype(np.uint8)
    return to_tensor_normalize(resize_bilinear_u8(patch, img_size))


class SceneFiles:
    """Read-once access to a BOP split folder: per-scene json (camera, ground truth) and per-image depth (metres,
    depth_scale applied) / colour arrays, each behind a small LRU."""

    def __init__(self, max_scenes=64, max_images=8):
        from collections import OrderedDict

        self._json, self._img = OrderedDict(), OrderedDict()
        self._max_json, self._max_img = 2 * max_scenes, 2 * max_images

    @staticmethod
    def _lru(store, key, limit, make):
        if key in store:
            store.move_to_end(key)
            return store[key]
        val = store[key] = make()
        while len(store) > limit:
            store.popitem(last=False)
        return val

    def scene_json(self, folder, scene_id, name):
        path = osp.join(folder, f"{scene_id:06d}", name)
        return self._lru(self._json, path, self._max_json, lambda: load_json(path))

    def camera(self, folder, scene_id, im_id):
        cam = self.scene_json(folder, scene_id, "scene_camera.json")[str(im_id)]
        return np.array(cam["cam_K"]).reshape((3, 3)).copy(), cam["depth_scale"]

    def depth_m(self, folder, scene_id, im_id, depth_scale):
        """Depth in metres: file value / 1000 * depth_scale (png first, tif as the fallback, data_utils.py:339-351)."""
        def make():
            base = osp.join(folder, f"{scene_id:06d}", "depth", f"{im_id:06d}")
            return read_image(base + ".png" if osp.exists(base + ".png") else base + ".tif") / 1000.0 * depth_scale

        return self._lru(self._img, ("d", folder, scene_id, im_id), self._max_img, make)

    def colour(self, folder, scene_id, im_id):
        def make():
            scene = osp.join(folder, f"{scene_id:06d}")
            for rel in (f"rgb/{im_id:06d}.jpg", f"rgb/{im_id:06d}.png", f"gray/{im_id:06d}.tif"):
                if osp.exists(osp.join(scene, rel)):
                    return read_image(osp.join(scene, rel)).astype(np.uint8)
            raise FileNotFoundError(f"no colour image for scene {scene_id} image {im_id} under {folder}")

        return self._lru(self._img, ("c", folder, scene_id, im_id), self._max_img, make)


class ReferenceViews:
    """The per-view, draw-independent half of a reference instance: visible-mask window, masked organised cloud
    (float32, reference camera frame, metres), flat indices of the mask pixels inside the window, normalised crop,
    object pose in that camera.  Built once per (scene, image, object) and reused by every query paired with it."""

    def __init__(self, files, img_size, rgb_mask_flag, bgr, max_views=64):
        from collections import OrderedDict

        self.files, self.img_size, self.rgb_mask_flag, self.bgr = files, img_size, rgb_mask_flag, bgr
        self._views, self._max = OrderedDict(), max_views

    def get(self, folder, scene_id, im_id, obj_id):
        key = (folder, scene_id, im_id, obj_id)
        if key in self._views:
            self._views.move_to_end(key)
            return self._views[key]
        view = self._views[key] = self._build(folder, scene_id, im_id, obj_id)
        while len(self._views) > self._max:
            self._views.popitem(last=False)
        return view

    def _build(self, folder, scene_id, im_id, obj_id):
        gts = self.files.scene_json(folder, scene_id, "scene_gt.json")[str(im_id)]
        slot = next((i for i, gt in enumerate(gts) if gt["obj_id"] == obj_id), None)
        if slot is None:
            return None
        pose = np.eye(4, dtype=np.float32)
        pose[:3, :3] = np.array(gts[slot]["cam_R_m2c"], dtype=np.float32).reshape(3, 3)
        pose[:3, 3] = np.array(gts[slot]["cam_t_m2c"], dtype=np.float32).reshape(3) * 0.001
        K, depth_scale = self.files.camera(folder, scene_id, im_id)
        depth = self.files.depth_m(folder, scene_id, im_id, depth_scale).astype("float32")
        visible = np.array(read_image(osp.join(folder, f"{scene_id:06d}", "mask_visib", f"{im_id:06d}_{slot:06d}.png"))).astype(bool)
        win = Window.around(visible)
        inside = win.crop(visible)
        cloud = lift_depth(depth, K, win)
        cloud *= inside.astype("float32")[:, :, None]
        crop = _normalised_crop(self.files.colour(folder, scene_id, im_id), win, self.img_size,
                                inside if self.rgb_mask_flag else None, self.bgr)
        return dict(window=win, cloud=cloud.reshape(-1, 3), pixels=np.flatnonzero(inside), crop=crop, pose=pose)


# ------------------------------------------------------------------------------------------------
class BOPTestsetOneRef:
    """One item = one test image: all detections above `seg_filter_score` (or the best one), each paired
    with its reference view from `ref_targets_name` (pfoneref_bop_test_dataset_v2.py:33-354).

    cfg fields (attribute or key access): data_dir, ref_targets_name, rgb_mask_flag, img_size,
    n_sample_observed_point, n_sample_template_point, minimum_n_point, seg_filter_score; optional
    rgb_to_bgr, obj_idxs, oneref_percat (+ targets_name, ref_scene_ims).

    `dets` maps "<scene:06d>_<image:06d>" to the image's detection dicts in file order (the runner deep-copies it
    for the detections json, like ``data_loader.dataset.dets``)."""

    def __init__(self, cfg, eval_dataset_name="lmo", detetion_path=None):
        assert detetion_path is not None
        opt = (lambda k, d=None: cfg.get(k, d)) if hasattr(cfg, "get") else (lambda k, d=None: getattr(cfg, k, d))
        self.cfg, self.dataset = cfg, eval_dataset_name
        for name in ("data_dir", "ref_targets_name", "rgb_mask_flag", "img_size", "n_sample_observed_point",
                     "n_sample_template_point", "minimum_n_point", "seg_filter_score"):
            setattr(self, name, opt(name))
        self.rgb_to_bgr = opt("rgb_to_bgr", False)
        self.obj_idxs = opt("obj_idxs") or {obj_id: i for i, obj_id in enumerate(DATASET_OBJ_IDS[eval_dataset_name])}
        self.data_folder = osp.join(self.data_dir, eval_dataset_name, "test")
        self.files = SceneFiles()
        self.ref_views = ReferenceViews(self.files, self.img_size, self.rgb_mask_flag, self.rgb_to_bgr)
        root = osp.join(self.data_dir, eval_dataset_name)
        if opt("oneref_percat", False):
            assert opt("targets_name") is not None
            self.test_ref_target = self.load_single_ref_per_dset(osp.join(root, opt("targets_name")), opt("ref_scene_ims"))
        else:
            self.test_ref_target = self.load_ref(osp.join(root, self.ref_targets_name))
        self.dets = {}
        for det in load_json(detetion_path):
            self.dets.setdefault(f"{int(det['scene_id']):06d}_{int(det['image_id']):06d}", []).append(det)
        self.det_keys = list(self.dets)  # first-appearance order of the images in the detection file

    def __len__(self):
        return len(self.det_keys)

    # ---- one image --------------------------------------------------------------------------------------------
    def __getitem__(self, index):
        key = self.det_keys[index]
        dets = self.dets[key]
        picked = [(i, inst) for i, inst in ((i, self.get_instance(d)) for i, d in enumerate(dets) if d["score"] > self.seg_filter_score)
                  if inst is not None]
        if not picked:  # nothing passed the score filter (or survived it): fall back to the best-scored detection
            best = max(range(len(dets)), key=lambda i: (dets[i]["score"], -i))
            inst = self.get_instance(dets[best])
            if inst is None:
                raise ValueError(f"no qulified instance in {key}")
            picked = [(best, inst)]
        insts = [inst for _, inst in picked]
        item = {k: torch.stack([inst[k] for inst in insts]) for k in insts[0] if k != "ref_key"}
        item["ref_keys"] = [inst["ref_key"] for inst in insts]  # this provider's addition (runner.ReferenceCache)
        item["scene_id"] = torch.IntTensor([int(key[:6])])
        item["img_id"] = torch.IntTensor([int(key[7:13])])
        item["inst_ids"] = torch.IntTensor([i for i, _ in picked])
        item["seg_time"] = torch.FloatTensor([dets[0]["time"]])
        return item

    # ---- one detection ----------------------------------------------------------------------------------------
    def get_instance(self, det):
        """Query side of one (detection, reference) pair, or None when the detection has too little valid depth, no
        reference target, or too few points near the reference's size.  Consumes the global ``np.random`` stream
        exactly like the reference: the reference-view draw first, then the query draw (:187-203, :296-304)."""
        scene_id, img_id, obj_id = det["scene_id"], det["image_id"], det["category_id"]
        K, depth_scale = self.files.camera(self.data_folder, scene_id, img_id)
        depth = self.files.depth_m(self.data_folder, scene_id, img_id, depth_scale)
        valid = np.logical_and(rle_decode(det["segmentation"]) > 0, depth > 0)  # segmentation restricted to measured depth
        if not np.sum(valid) > self.minimum_n_point:
            return None
        win = Window.around(valid)
        inside = win.crop(valid)
        pixels = np.flatnonzero(inside)
        cloud = lift_depth(depth, K, win).reshape(-1, 3)[pixels]
        centred = cloud - np.mean(cloud, axis=0)[None, :]

        ref = self._reference_instance(scene_id, img_id, obj_id)
        if ref is None:
            return None
        ref_crop, ref_choose, ref_pts, ref_pose, ref_key = ref
        # keep what lies within 1.2 reference radii of the observed centroid (drops background leaking into the mask)
        radius = np.max(np.linalg.norm(ref_pts - np.mean(ref_pts, axis=0).reshape(1, 3), axis=1))
        near = np.linalg.norm(centred, axis=1) < 1.2 * radius
        if np.sum(near) < self.minimum_n_point:
            return None
        pixels, cloud = pixels[near], cloud[near]
        n = self.n_sample_observed_point
        take = np.random.choice(np.arange(len(pixels)), size=n, replace=len(pixels) <= n)
        pixels, cloud = pixels[take], cloud[take]
        crop = _normalised_crop(self.files.colour(self.data_folder, scene_id, img_id), win, self.img_size,
                                inside if self.rgb_mask_flag else None, self.rgb_to_bgr)
        return {
            "pts": torch.FloatTensor(cloud),
            "rgb": torch.FloatTensor(crop),
            "rgb_choose": torch.IntTensor(win.to_resized(pixels, self.img_size)).long(),
            "obj": torch.IntTensor([self.obj_idxs[obj_id]]).long(),
            "obj_id": torch.IntTensor([obj_id]),
            "score": torch.FloatTensor([det["score"]]),
            "tem1_rgb": torch.FloatTensor(ref_crop),
            "tem1_choose": torch.IntTensor(ref_choose).long(),
            "tem1_pts": torch.FloatTensor(ref_pts),
            "tem1_pose": torch.FloatTensor(ref_pose),
            "ref_key": ref_key,
        }

    def _ref_split_folder(self, ref_scene_id):
        """Where a reference scene lives: ycbv references outside the 12 test scenes and all tudl references come
        from train_real (:246-251)."""
        train_real = osp.join(self.data_dir, self.dataset, "train_real")
        if self.dataset == "ycbv":
            return self.data_folder if 48 <= ref_scene_id < 60 else train_real
        return train_real if self.dataset == "tudl" else self.data_folder

    def _reference_instance(self, scene_id, img_id, obj_id):
        target = self.test_ref_target.get(f"{scene_id}_{img_id}_{obj_id}")
        if target is None:
            return None
        ref_scene, ref_im = (int(v) for v in target.split("_"))
        view = self.ref_views.get(self._ref_split_folder(ref_scene), ref_scene, ref_im, obj_id)
        if view is None:
            return None
        pixels, n = view["pixels"], self.n_sample_template_point
        if len(pixels) <= n:  # same two call forms as the reference, so the random stream advances identically
            take = np.random.choice(np.arange(len(pixels)), n)
        else:
            take = np.random.choice(np.arange(len(pixels)), n, replace=False)
        pixels = pixels[take]
        return (view["crop"], view["window"].to_resized(pixels, self.img_size), view["cloud"][pixels, :], view["pose"],
                (ref_scene, ref_im, obj_id))

    # ---- target lists -----------------------------------------------------------------------------------------
    @staticmethod
    def load_ref(path):
        """[{scene_id, im_id, obj_id, ref_scene_id, ref_im_id}] -> {"scene_im_obj": "refscene_refim"} (:314-331)."""
        return {f"{t['scene_id']}_{t['im_id']}_{t['obj_id']}": f"{t['ref_scene_id']}_{t['ref_im_id']}"
                for t in load_json(path)}

    def load_single_ref_per_dset(self, test_target_path, ref_scene_ims):
        """One fixed reference view per object id, `ref_scene_ims[i]` = "scene_im" of object i + 1 (:333-354)."""
        assert len(ref_scene_ims) == len(self.obj_idxs)
        return {f"{t['scene_id']}_{t['im_id']}_{t['obj_id']}": ref_scene_ims[t["obj_id"] - 1] for t in load_json(test_target_path)}


def collate_image(item):
    """DataLoader(batch_size=1) view of one item: tensors get the leading image dimension the runner
    indexes with [0] (oneref_inference_utils_v1.py:36-66)."""
    return {k: (v[None] if torch.is_tensor(v) else v) for k, v in item.items()}
