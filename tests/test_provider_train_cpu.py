"""unopose_amd/provider_train.py (the training-dataset item contract) against a fixture produced by the reference's
own ``DatasetPoseFreeOneRefv2`` on the synthetic MegaPose tree (tests/golden/make_provider_train_golden.py), plus known-answer /
property tests for the pieces restated there (cross dilation, uncompressed RLE, the colour-augmentation chain)."""
import os

import numpy as np
import pytest
import torch

import megapose_synth
from unopose_amd import provider_train as T

GOLD = os.path.join(os.path.dirname(__file__), "golden")
EPOCH, SEED = 14, 77


@pytest.fixture(scope="module")
def tree(tmp_path_factory):
    return megapose_synth.build(str(tmp_path_factory.mktemp("megapose")))


@pytest.mark.parametrize("tag,over", [("a", {}), ("b", dict(dilate_mask=False, rgb_mask_flag=False, rgb_to_bgr=True))])
def test_seeded_epoch_reproduces_the_reference_items(tree, tag, over):
    """Same files, same seed -> the same epoch order, the same 9-key items bit for bit (dtypes included) and the same position of
    the global np.random stream afterwards; the epoch walks over a view without valid instances, a view whose files are missing
    and an instance that is too small, so the retry path (`_rand_another`) is part of what is pinned."""
    z = np.load(os.path.join(GOLD, "provider_train.npz"))
    ds = T.MegaPoseOneRefTrainSet(dict(tree, **over), num_img_per_epoch=EPOCH, color_augmentor=None)
    assert len(ds) == EPOCH and ds.length == 14  # 2 subsets x (2 shards x 3 views + 1 missing key)
    np.random.seed(SEED)
    ds.reset()
    assert np.array_equal(ds.img_idx, z[f"{tag}__img_idx"])
    for i in range(EPOCH):
        item = ds[i]
        assert sorted(item) == sorted(["pts", "rgb", "rgb_choose", "translation_label", "rotation_label", "tem1_rgb", "tem1_choose", "tem1_pts", "K"])
        for k, v in item.items():
            want = z[f"{tag}__item{i}__{k}"]
            assert v.numpy().dtype == want.dtype and v.shape == want.shape, (i, k)
            assert np.array_equal(v.numpy(), want), (i, k, np.abs(v.numpy().astype(np.float64) - want).max())
    assert np.array_equal(np.random.rand(2), z[f"{tag}__next_random"])


def test_items_are_consistent_pairs(tree):
    """Property of every item (no fixture): the label maps the rotated reference cloud onto the query cloud --
    rotation_label @ p_ref + translation_label lies on the query surface up to the shift, the noise and the sampling."""
    ds = T.MegaPoseOneRefTrainSet(tree, num_img_per_epoch=6, color_augmentor=None)
    np.random.seed(3)
    ds.reset()
    for i in range(6):
        it = ds[i]
        R, t = it["rotation_label"].double(), it["translation_label"].double()
        assert torch.allclose(R @ R.T, torch.eye(3, dtype=torch.float64), atol=1e-5) and abs(float(torch.det(R)) - 1) < 1e-5
        assert it["rgb"].shape == (3, 56, 56) and it["rgb_choose"].max() < 56 * 56 and it["tem1_choose"].max() < 56 * 56
        moved = it["tem1_pts"].double() @ R.T + t
        # the synthetic instances are blobs on tilted planes seen by one camera; the same object id in two views shares
        # neither shape nor pose model, so only scale agreement is meaningful: both clouds are within one object radius
        rad = (it["tem1_pts"] - it["tem1_pts"].mean(0)).norm(dim=1).max()
        assert (it["pts"] - it["pts"].mean(0)).norm(dim=1).max() < 1.2 * rad + 0.02
        assert torch.isfinite(moved).all()


def test_collate_and_model_input_contract(tree):
    ds = T.MegaPoseOneRefTrainSet(tree, num_img_per_epoch=4, color_augmentor="default", seed=0)
    np.random.seed(1)
    ds.reset()
    batch = T.collate_pairs([ds[i] for i in range(4)])
    assert batch["pts"].shape == (4, 192, 3) and batch["tem1_pts"].shape == (4, 300, 3) and batch["rotation_label"].shape == (4, 3, 3)
    assert batch["rgb"].dtype == torch.float32 and batch["rgb_choose"].dtype == torch.int64


def test_dilate_cross_equals_scipy_binary_dilation():
    from scipy import ndimage

    rs = np.random.RandomState(0)
    cross = ndimage.generate_binary_structure(2, 1)
    for _ in range(10):
        m = rs.rand(rs.randint(5, 40), rs.randint(5, 40)) < 0.08
        m[0, 0] = True  # touches the border
        for it in (1, 4):
            assert np.array_equal(T.dilate_cross(m, it).astype(bool), ndimage.binary_dilation(m, cross, iterations=it))
    one = np.zeros((11, 11), bool)
    one[5, 5] = True
    d = T.dilate_cross(one, 4)  # known answer: the diamond |dy| + |dx| <= 4 (41 pixels)
    yy, xx = np.mgrid[0:11, 0:11]
    assert d.dtype == np.uint8 and np.array_equal(d.astype(bool), np.abs(yy - 5) + np.abs(xx - 5) <= 4) and d.sum() == 41


def test_rle_list_round_trip_and_known_answer():
    m = np.zeros((3, 4), bool)
    m[1, 0] = m[2, 0] = m[0, 1] = True  # column-major flat: 0 1 1 | 1 0 0 | ... -> runs 1, 3, 8
    assert megapose_synth._rle_list(m)["counts"] == [1, 3, 8]
    assert np.array_equal(T.rle_list_to_mask({"size": [3, 4], "counts": [1, 3, 8]}), m)
    rs = np.random.RandomState(1)
    for _ in range(10):
        m = rs.rand(rs.randint(2, 30), rs.randint(2, 30)) < 0.5
        assert np.array_equal(T.rle_list_to_mask(megapose_synth._rle_list(m)), m)
    assert not T.rle_list_to_mask({"size": [4, 4], "counts": [16]}).any()  # empty mask: one zero-run


def test_color_augmentor_statistics():
    """PARITY UNPINNED piece (no imgaug here): what can be tested is the contract -- uint8 in / out, same shape, deterministic per
    seed, every operator alone keeps the image valid, and the documented effect of the simple operators."""
    rs = np.random.RandomState(2)
    img = rs.randint(0, 256, size=(40, 52, 3)).astype(np.uint8)
    a, b = T.ColorAugmentor(5), T.ColorAugmentor(5)
    outs = [a.augment_image(img) for _ in range(20)]
    assert all(o.dtype == np.uint8 and o.shape == img.shape for o in outs)
    assert all(np.array_equal(o, b.augment_image(img)) for o in outs)  # same seed, same stream
    assert sum(not np.array_equal(o, img) for o in outs) >= 18  # P(no operator fires) = 0.5^7 0.6 0.7^4 0.9 < 0.1 %
    aug = T.ColorAugmentor(9)
    for p, op in aug.ops:
        o = op(img)
        assert o.dtype == np.uint8 and o.shape == img.shape, op.__name__
    flat = np.full((30, 30, 3), 100, np.uint8)
    m = aug.multiply(flat)
    assert len(np.unique(m)) == 1 and 60 <= int(m[0, 0, 0]) <= 140
    inv = [aug.invert(flat) for _ in range(200)]
    frac = np.mean([(o[0, 0] == 155).mean() for o in inv])
    assert 0.12 < frac < 0.28  # each channel flips with probability 0.2
    drops = np.mean([(aug.coarse_dropout(flat) == 0).mean() for _ in range(200)])
    assert 0.12 < drops < 0.28  # 20 % of the coarse cells
    g = aug.grayscale(img)
    assert np.abs(g.astype(int).std(axis=2)).mean() < img.astype(int).std(axis=2).mean()  # channels pulled together
    assert aug.augment_image(np.zeros((0, 5, 3), np.uint8)).size == 0
