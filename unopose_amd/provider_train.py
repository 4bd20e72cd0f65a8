"""MegaPose training-pair provider for the one-reference setting: the producer side of the TRAINING step's input dict
(BASELINE configs[3]).

Contract = the items of ``DatasetPoseFreeOneRefv2`` (core/unopose/provider/pfoneref_training_dataset_v2.py:75-590): the same
directory layout (MegaPose-GSO / MegaPose-ShapeNetCore ``train_pbr_web`` shards, the two "valid instances" and "reference
candidates" json files), the same constructor fields, the same item dict (keys, dtypes, shapes: ``pts``, ``rgb``, ``rgb_choose``,
``translation_label``, ``rotation_label``, ``tem1_rgb``, ``tem1_choose``, ``tem1_pts``, ``K``) and the same consumption of the
global ``np.random`` stream (instance pick, reference pick, colour-augmentation coin, reference point draw, dilation coin,
query point draw, colour-augmentation coin, random rotation, shift, per-point noise), so a seeded epoch reproduces the
reference's samples bit for bit when the colour augmentation is the identity (tests/golden/make_provider_train_golden.py runs the
reference class on the same files).

The structure is this repo's own: one :class:`ViewFiles` value per (shard, key) gives lazily loaded, once-parsed access to the six
files of a MegaPose view; the window / back-projection / resize / normalisation helpers are those of ``provider.py``.

Third-party pieces the reference calls that this image lacks, restated here:
  * ``cv2.dilate(mask, MORPH_CROSS 3x3, iterations=4)`` -> :func:`dilate_cross` (exact: binary dilation by the 4-neighbourhood, four
    times, nothing enters from outside the image; checked against scipy.ndimage.binary_dilation in the tests);
  * ``cv2.resize`` -> ``provider.resize_bilinear_u8`` (see the note there);
  * the imgaug colour-augmentation chain (``aug_code``, :158-176) -> :class:`ColorAugmentor`: the same thirteen operators with the
    same probabilities and parameter ranges, drawn from an own ``np.random.Generator`` (imgaug keeps its own random state too, so the
    global stream is untouched either way).  PARITY UNPINNED for this piece: imgaug is not importable in the build container; the
    operators follow imgaug's documented semantics and the sample statistics are tested, not the pixels."""
import os.path as osp

import numpy as np
import torch

from .provider import Window, lift_depth, load_json, read_image, resize_bilinear_u8, to_tensor_normalize

VIEW_SUFFIXES = (".camera.json", ".depth.png", ".gt_info.json", ".gt.json", ".mask_visib.json", ".rgb.jpg")  # :451-458
SUBSETS = (("GSO", osp.join("MegaPose-GSO", "train_pbr_web"), "megapose_gso_fixed", "gso_models.json"),
           ("ShapeNetCore", osp.join("MegaPose-ShapeNetCore", "train_pbr_web"), "megapose_shapenetcore_fixed", "shapenet_models.json"))


def dilate_cross(mask, iterations=4):
    """cv2.dilate(mask_u8, cv2.getStructuringElement(cv2.MORPH_CROSS, (3, 3)), iterations=n) for a 0/1 mask -> uint8 0/1."""
    m = np.asarray(mask) > 0
    for _ in range(int(iterations)):
        g = m.copy()
        g[1:, :] |= m[:-1, :]
        g[:-1, :] |= m[1:, :]
        g[:, 1:] |= m[:, :-1]
        g[:, :-1] |= m[:, 1:]
        m = g
    return m.astype(np.uint8)


def rle_list_to_mask(rle):
    """MegaPose's mask files hold UNCOMPRESSED COCO run lengths (a list of ints, column-major, starting with a 0-run):
    data_utils.py:168-185."""
    h, w = rle["size"]
    counts = np.asarray(rle["counts"], np.int64)
    runs = np.repeat((np.arange(len(counts)) % 2).astype(bool), counts)[: h * w]
    flat = np.zeros(h * w, dtype=bool)
    flat[: len(runs)] = runs
    return flat.reshape(w, h).T  # column-major, as reshape(h, w, order="F")


def random_rotation_xyz():
    """data_utils.py:286-296: Rx(a0) Ry(a1) Rz(a2) with three angles `np.random.rand(3) * 2 pi` (float64)."""
    a = np.random.rand(3) * 2 * np.pi
    c, s = np.cos(a), np.sin(a)
    rx = np.array([[1, 0, 0], [0, c[0], -s[0]], [0, s[0], c[0]]])
    ry = np.array([[c[1], 0, s[1]], [0, 1, 0], [-s[1], 0, c[1]]])
    rz = np.array([[c[2], -s[2], 0], [s[2], c[2], 0], [0, 0, 1]])
    return rx @ ry @ rz


class ColorAugmentor:
    """The reference's colour augmentation (pfoneref_training_dataset_v2.py:158-176; "gdrnpp aug"): thirteen `Sometimes(p, op)`
    applied in a random order per image.  uint8 HxWx3 in, uint8 out.  Own random generator (see the module docstring)."""

    def __init__(self, seed=None):
        self.rng = np.random.default_rng(seed)
        self.ops = [(0.5, self.coarse_dropout), (0.4, self.gaussian_blur), (0.3, self.sharpness), (0.3, self.contrast),
                    (0.5, self.brightness), (0.3, self.color), (0.5, self.add), (0.3, self.invert), (0.5, self.multiply_pc),
                    (0.5, self.multiply), (0.1, self.gaussian_noise), (0.5, self.linear_contrast), (0.5, self.grayscale)]

    def augment_image(self, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        if img.size == 0:
            return img
        for i in self.rng.permutation(len(self.ops)):
            p, op = self.ops[i]
            if self.rng.random() < p:
                img = op(img)
        return img

    __call__ = augment_image

    @staticmethod
    def _u8(x):
        return np.clip(np.rint(x), 0, 255).astype(np.uint8)

    def _enhance(self, img, name, lo, hi):
        from PIL import Image, ImageEnhance

        return np.asarray(getattr(ImageEnhance, name)(Image.fromarray(img)).enhance(float(self.rng.uniform(lo, hi))))

    def coarse_dropout(self, img):  # CoarseDropout(p=0.2, size_percent=0.05): a coarse drop mask, nearest-upsampled, pixels -> 0
        h, w = img.shape[:2]
        ch, cw = max(3, int(round(h * 0.05))), max(3, int(round(w * 0.05)))
        drop = self.rng.random((ch, cw)) < 0.2
        yy = np.minimum((np.arange(h) * ch) // max(h, 1), ch - 1)
        xx = np.minimum((np.arange(w) * cw) // max(w, 1), cw - 1)
        return img * (~drop[yy][:, xx])[:, :, None].astype(np.uint8)

    def gaussian_blur(self, img):  # GaussianBlur(sigma in (0, 3))
        from PIL import Image, ImageFilter

        sigma = float(self.rng.uniform(0.0, 3.0))
        return img if sigma < 1e-3 else np.asarray(Image.fromarray(img).filter(ImageFilter.GaussianBlur(sigma)))

    def sharpness(self, img):
        return self._enhance(img, "Sharpness", 0.0, 50.0)

    def contrast(self, img):
        return self._enhance(img, "Contrast", 0.2, 50.0)

    def brightness(self, img):
        return self._enhance(img, "Brightness", 0.1, 6.0)

    def color(self, img):
        return self._enhance(img, "Color", 0.0, 20.0)

    def _per_channel(self, p, draw):
        return draw(3).reshape(1, 1, 3) if self.rng.random() < p else draw(1).reshape(1, 1, 1)

    def add(self, img):  # Add((-25, 25), per_channel=0.3): integer offsets
        return self._u8(img.astype(np.float32) + self._per_channel(0.3, lambda n: self.rng.integers(-25, 26, n).astype(np.float32)))

    def invert(self, img):  # Invert(0.2, per_channel=True)
        flip = self.rng.random(3) < 0.2
        return np.where(flip.reshape(1, 1, 3), 255 - img, img).astype(np.uint8)

    def multiply_pc(self, img):  # Multiply((0.6, 1.4), per_channel=0.5)
        return self._u8(img.astype(np.float32) * self._per_channel(0.5, lambda n: self.rng.uniform(0.6, 1.4, n).astype(np.float32)))

    def multiply(self, img):
        return self._u8(img.astype(np.float32) * np.float32(self.rng.uniform(0.6, 1.4)))

    def gaussian_noise(self, img):  # AdditiveGaussianNoise(scale=10, per_channel=True)
        return self._u8(img.astype(np.float32) + self.rng.normal(0.0, 10.0, img.shape).astype(np.float32))

    def linear_contrast(self, img):  # LinearContrast((0.5, 2.2), per_channel=0.3): 128 + alpha (v - 128)
        a = self._per_channel(0.3, lambda n: self.rng.uniform(0.5, 2.2, n).astype(np.float32))
        return self._u8(128.0 + a * (img.astype(np.float32) - 128.0))

    def grayscale(self, img):  # Grayscale(alpha in (0, 1)): blend with the luma image
        alpha = np.float32(self.rng.uniform(0.0, 1.0))
        g = img.astype(np.float32) @ np.array([0.299, 0.587, 0.114], np.float32)
        return self._u8((1 - alpha) * img.astype(np.float32) + alpha * g[:, :, None])


class ViewFiles:
    """The six files of one MegaPose view ``<root>/<subset>/<shard:06d>/<key>.*``, each parsed at most once."""

    def __init__(self, head):
        self.head = head
        self._camera = self._gt = self._masks = None

    def complete(self):
        return all(osp.exists(self.head + s) for s in VIEW_SUFFIXES)

    @property
    def camera(self):
        if self._camera is None:
            self._camera = load_json(self.head + ".camera.json")
        return self._camera

    @property
    def K(self):
        return np.array(self.camera["cam_K"]).reshape(3, 3).astype(np.float32)

    def pose(self, inst):
        """4 x 4 float32 object-to-camera pose of instance `inst` (translation in metres) and its obj_id."""
        if self._gt is None:
            self._gt = load_json(self.head + ".gt.json")
        g = self._gt[inst]
        T = np.eye(4, dtype=np.float32)
        T[:3, :3] = np.array(g["cam_R_m2c"], np.float32).reshape(3, 3)
        T[:3, 3] = np.array(g["cam_t_m2c"], np.float32).reshape(3) / 1000.0
        return T, g["obj_id"]

    def mask(self, inst_id=None, position=None):
        """Visible mask (bool H x W) of the instance with id `inst_id`, or of the `position`-th instance in id order (the
        reference indexes the stacked, id-sorted masks with the valid-instance number: :247)."""
        if self._masks is None:
            self._masks = {int(k): v for k, v in load_json(self.head + ".mask_visib.json").items()}
        if inst_id is None:
            inst_id = sorted(self._masks)[position]
        return rle_list_to_mask(self._masks[inst_id])

    def depth_m(self):
        return read_image(self.head + ".depth.png").astype(np.float32) * self.camera["depth_scale"] / 1000.0

    def colour(self):
        return read_image(self.head + ".rgb.jpg").astype(np.uint8)


class MegaPoseOneRefTrainSet:
    """cfg: data_dir, img_size, n_sample_observed_point, n_sample_model_point, n_sample_template_point, min_px_count_visib,
    min_visib_fract, dilate_mask, rgb_mask_flag, shift_range, optional rgb_to_bgr (attribute or key access).
    ``color_augmentor``: "default" -> :class:`ColorAugmentor`; None -> identity; or any object with ``augment_image``.
    Usage as the reference's (build_data_loader.py): ``ds.reset()`` once per epoch, then ``ds[i]`` for i < len(ds)."""

    def __init__(self, cfg, num_img_per_epoch=-1, color_augmentor="default", seed=None):
        get = (lambda k, *d: cfg.get(k, *d)) if hasattr(cfg, "get") else (lambda k, *d: getattr(cfg, k, *d))
        self.data_dir = get("data_dir")
        self.num_img_per_epoch = num_img_per_epoch
        self.dilate_mask, self.rgb_mask_flag = get("dilate_mask"), get("rgb_mask_flag")
        self.shift_range, self.img_size = get("shift_range"), get("img_size")
        self.n_obs, self.n_tpl = get("n_sample_observed_point"), get("n_sample_template_point")
        self.bgr = get("rgb_to_bgr", False)
        self.views, self.subset_dir, self.references, self.valid = [], {}, {}, {}
        for name, rel, prefix, _models in SUBSETS:
            self.subset_dir[name] = rel
            shards = load_json(osp.join(self.data_dir, rel, "key_to_shard.json"))
            self.views += [(name, f"{shards[k]:06d}", k) for k in shards]  # key order of the json file (:119-124)
            self.references[name] = load_json(osp.join(self.data_dir, prefix + "_obj_id_to_visib0_8_scene_im_inst_ids.json"))
            self.valid[name] = load_json(osp.join(self.data_dir, prefix + "_valid_inst_ids.json"))
        self.length = len(self.views)
        self.color_augmentor = ColorAugmentor(seed) if isinstance(color_augmentor, str) else color_augmentor
        self.img_idx = None

    def __len__(self):
        return self.length if self.num_img_per_epoch == -1 else self.num_img_per_epoch

    def reset(self):
        """Draw the epoch's view indices (:182-191)."""
        if self.num_img_per_epoch == -1:
            self.num_img_per_epoch = self.length
        self.img_idx = np.random.choice(self.length, self.num_img_per_epoch, replace=self.length <= self.num_img_per_epoch)

    def __getitem__(self, index):
        while True:  # a view without a usable sample is replaced by another position of the epoch (:193-204)
            item = self.read_data(self.img_idx[index])
            if item is not None:
                return item
            index = np.random.choice([i for i in range(len(self)) if i != index])

    def _files(self, subset, shard, key):
        return ViewFiles(osp.join(self.data_dir, self.subset_dir[subset], shard, key))

    def _crop_tensor(self, files, win, mask):
        """colour -> window -> (80 %: colour augmentation) -> foreground only -> resize -> ImageNet-normalised CHW."""
        rgb = files.colour()
        rgb = win.crop(rgb[..., ::-1] if self.bgr else rgb)
        if np.random.rand() < 0.8 and self.color_augmentor is not None:
            rgb = self.color_augmentor.augment_image(rgb)
        if self.rgb_mask_flag:
            rgb = rgb * (mask[:, :, None] > 0).astype(np.uint8)
        return to_tensor_normalize(resize_bilinear_u8(np.ascontiguousarray(rgb), self.img_size))

    @staticmethod
    def _draw(n_have, n_want):
        return np.random.choice(np.arange(n_have), n_want, replace=n_have <= n_want)

    def reference_view(self, subset, obj_id):
        """One random reference view of the object (:399-449) -> (rgb CHW, choose (n_tpl,), camera-space points (n_tpl,3) float64,
        4 x 4 pose) or None."""
        cands = self.references[subset][str(obj_id)]
        if len(cands) == 0:
            return None
        shard, key, inst = cands[np.random.choice(list(range(len(cands))))]
        files = self._files(subset, f"{shard:06d}", key)
        mask = files.mask(inst_id=inst)
        if mask.sum() == 0:
            return None
        win = Window.around(mask)
        mask = win.crop(mask)
        if mask.sum() == 0:
            return None
        rgb = self._crop_tensor(files, win, mask)
        choose = mask.astype(np.float32).flatten().nonzero()[0]
        choose = choose[self._draw(len(choose), self.n_tpl)]
        K = files.K
        xyz = lift_depth(files.depth_m(), K, win).reshape(-1, 3)[choose]
        return rgb, win.to_resized(choose, self.img_size), xyz, files.pose(inst)[0]

    def read_data(self, index):
        subset, shard, key = self.views[index]
        files = self._files(subset, shard, key)
        if not files.complete():
            return None
        insts = self.valid[subset].get(f"{shard}/{key}", [])
        if len(insts) == 0:
            return None
        inst = insts[np.random.randint(0, len(insts))]  # one valid instance per visit (:208-210)
        pose_q, obj_id = files.pose(inst)
        assert len(self.references[subset][str(obj_id)]) != 0
        K = files.K
        ref = self.reference_view(subset, obj_id)
        if ref is None:
            return None
        tem_rgb, tem_choose, tem_pts, pose_r = ref
        rel = pose_q @ np.linalg.inv(pose_r)  # reference camera -> query camera (float32, :242)
        radius = np.max(np.linalg.norm(tem_pts - np.mean(tem_pts, axis=0).reshape(1, 3), axis=1))

        mask = files.mask(position=inst)
        if np.sum(mask) == 0:
            return None
        if self.dilate_mask and np.random.rand() < 0.5:
            mask = dilate_cross(mask, 4)
        win = Window.around(mask > 0)
        mask = win.crop(mask)
        if np.sum(mask) == 0:
            return None
        choose = mask.astype(np.float32).flatten().nonzero()[0]
        pts = lift_depth(files.depth_m(), K, win).reshape(-1, 3)[choose]
        keep = np.linalg.norm(pts - np.mean(pts, axis=0).reshape(1, 3), axis=1) < 1.2 * radius  # outliers w.r.t. the reference's extent
        pts, choose = pts[keep], choose[keep]
        if len(choose) < 32:
            return None
        sel = self._draw(len(choose), self.n_obs)
        choose, pts = choose[sel], pts[sel]
        rgb = self._crop_tensor(files, win, mask)
        rgb_choose = win.to_resized(choose, self.img_size)

        # rotation augmentation of the reference cloud, translation shift + per-point noise on the query cloud (:336-356)
        spin = np.eye(4, dtype=np.float32)
        spin[:3, :3] = random_rotation_xyz()
        tem_pts = tem_pts @ spin[:3, :3]
        target = rel @ spin
        shift = np.random.uniform(-self.shift_range, self.shift_range, (1, 3))
        target_t = target[:3, 3] + shift[0]
        pts = np.add(pts, shift + 0.001 * np.random.randn(pts.shape[0], 3))
        return {"pts": torch.FloatTensor(pts), "rgb": torch.FloatTensor(rgb), "rgb_choose": torch.IntTensor(rgb_choose).long(),
                "translation_label": torch.FloatTensor(target_t), "rotation_label": torch.FloatTensor(target[:3, :3]),
                "tem1_rgb": torch.FloatTensor(tem_rgb), "tem1_choose": torch.IntTensor(tem_choose).long(),
                "tem1_pts": torch.FloatTensor(tem_pts), "K": torch.FloatTensor(K)}


def collate_pairs(items):
    """Default collation of the training loader: stack every key (all items have the same shapes)."""
    return {k: torch.stack([it[k] for it in items]) for k in items[0]}
