"""unopose_amd/provider.py against fixtures produced by the reference's own provider
and helpers (tests/golden/make_provider_golden.py), plus known-answer / property tests for the two
third-party calls restated there (COCO RLE, OpenCV's fixed-point bilinear resize)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import bop_synth
from unopose_amd import provider as P

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_helpers_match_reference_golden():
    z = np.load(os.path.join(GOLD, "provider_helpers.npz"))
    for i in range(12):
        m = z[f"bbox_mask_{i}"]
        assert P.Window.around(m).as_list() == z[f"bbox_out_{i}"].tolist()
        rle = {"size": list(m.shape), "counts": z[f"rle_counts_{i}"].tolist()}
        assert np.array_equal(P.rle_decode(rle).astype(bool), z[f"rle_mask_{i}"])  # data_utils.rle_to_binary_mask
        assert np.array_equal(P.rle_decode(rle).astype(bool), m)
    win = P.Window(3, 33, 10, 40)
    assert np.array_equal(P.lift_depth(z["bp_depth"], z["bp_K"]), z["bp_full"])
    assert np.array_equal(P.lift_depth(z["bp_depth"], z["bp_K"], win), z["bp_crop"])
    for size in (224, 56, 518):
        assert np.array_equal(win.to_resized(z["rc_choose"], size), z[f"rc_out_{size}"])


def test_bbox_edge_cases():
    m = np.zeros((40, 60), bool)
    m[0:3, 55:60] = True  # corner: the square (side 2*int(5/2) = 4, the reference's rounding) is pushed back inside
    y1, y2, x1, x2 = P.Window.around(m).as_list()
    assert (y1, y2, x1, x2) == (0, 4, 55, 59)
    m[:] = True  # whole image: side clipped to min(H, W)
    y1, y2, x1, x2 = P.Window.around(m).as_list()
    assert y2 - y1 == 40 and x2 - x1 == 40 and P.Window.around(m).side == 40


def test_rle_string_known_answers_and_round_trip():
    # maskApi.c rleToString by hand: 5 -> '5'; 40 = 8 + 1*32 -> chr(48 + (8|0x20)) chr(48 + 1) = 'X1'
    assert P.rle_counts_to_string([5]) == "5" and P.rle_counts_to_string([40]) == "X1"
    assert P.rle_counts_from_string("5X1") == [5, 40]
    rs = np.random.RandomState(0)
    for _ in range(20):
        n = rs.randint(1, 30)
        counts = rs.randint(0, 5000, size=n).tolist()  # deltas vs counts[i-2] go negative: sign handling
        assert P.rle_counts_from_string(P.rle_counts_to_string(counts)) == counts
    m = rs.rand(33, 47) < 0.4
    m[0, 0] = True  # mask starting with a 1-run -> leading zero-length 0-run
    rle = P.rle_encode(m)
    assert rle["counts"][0] == 0 and np.array_equal(P.rle_decode(rle).astype(bool), m)
    packed = {"size": rle["size"], "counts": P.rle_counts_to_string(rle["counts"])}
    assert np.array_equal(P.rle_decode(packed).astype(bool), m)


def test_resize_bilinear_u8_properties():
    rs = np.random.RandomState(1)
    img = rs.randint(0, 256, size=(37, 37, 3)).astype(np.uint8)
    assert np.array_equal(P.resize_bilinear_u8(img, 37), img)
    big = rs.randint(0, 256, size=(48, 48, 3)).astype(np.uint8)
    a = big.astype(np.int32)
    box = (a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2
    assert np.array_equal(P.resize_bilinear_u8(big, 24), box.astype(np.uint8))  # OpenCV's exact-2x shortcut
    const = np.full((19, 19, 3), 173, np.uint8)
    for size in (7, 19, 56, 224):
        assert np.array_equal(P.resize_bilinear_u8(const, size), np.full((size, size, 3), 173, np.uint8))
    # 2 -> 4 upscale of [0, 200]: taps at -0.25, 0.25, 0.75, 1.25 -> weights (2048,0) (1536,512) (512,1536) (2048,0)
    ramp = np.array([[0, 200], [0, 200]], np.uint8)
    row = [0, (((2048 * ((200 * 512) >> 4)) >> 16) + 2) >> 2, (((2048 * ((200 * 1536) >> 4)) >> 16) + 2) >> 2, 200]
    assert P.resize_bilinear_u8(ramp, 4)[0].tolist() == row == [0, 50, 150, 200]
    # against float bilinear (half-pixel centres, no antialias) within one grey level, up and down
    for size in (56, 224, 30):
        ref = F.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None].float(), size=(size, size), mode="bilinear",
                            align_corners=False)[0].permute(1, 2, 0).numpy()
        assert np.abs(P.resize_bilinear_u8(img, size).astype(np.float32) - ref).max() <= 1.0
    gray = img[:, :, 0]
    assert np.array_equal(P.resize_bilinear_u8(gray, 56), P.resize_bilinear_u8(img, 56)[:, :, 0])


def test_to_tensor_normalize():
    img = np.random.RandomState(2).randint(0, 256, size=(5, 6, 3)).astype(np.uint8)
    t = P.to_tensor_normalize(img)
    assert t.shape == (3, 5, 6) and t.dtype == torch.float32
    ref = (img.transpose(2, 0, 1).astype(np.float32) / np.float32(255) - np.array(P.IMAGENET_MEAN, np.float32)[:, None, None]) \
        / np.array(P.IMAGENET_STD, np.float32)[:, None, None]
    assert np.abs(t.numpy() - ref).max() < 1e-6


@pytest.fixture(scope="module")
def dataset(tmp_path_factory):
    root = str(tmp_path_factory.mktemp("bop"))
    cfg, det_path = bop_synth.build(root)
    return P.BOPTestsetOneRef(cfg, "ycbv", det_path)


def test_dataset_matches_reference_provider_golden(dataset):
    """Same seed -> the reference class (run on the same files in the build container) and this provider
    return identical items: every key, dtype, shape and value (ints bit-exact, floats equal)."""
    z = np.load(os.path.join(GOLD, "provider_dataset.npz"))
    assert len(dataset) == int(z["n_items"]) == 2
    np.random.seed(2024)
    items = [dataset[i] for i in range(len(dataset))]
    for i, it in enumerate(items):
        keys = {k.split("__", 1)[1] for k in z.files if k.startswith(f"item{i}__")}
        assert keys == set(it.keys()) - {"ref_keys"}
        for k in keys:
            got, want = it[k].numpy(), z[f"item{i}__{k}"]
            assert got.dtype == want.dtype and got.shape == want.shape, k
            assert np.array_equal(got, want), (i, k, np.abs(got.astype(np.float64) - want).max())
    assert items[0]["ref_keys"] == [(10, 5, 2), (49, 7, 5)] and items[1]["ref_keys"] == [(10, 5, 2)]
    # the low-score duplicate detection was dropped; instance ids index the detection list of the image
    assert items[0]["inst_ids"].tolist() == [0, 1] and items[0]["score"].flatten().tolist() == pytest.approx([0.9, 0.7])


def test_dataset_invariants(dataset):
    np.random.seed(5)
    it = dataset[0]
    n, s = bop_synth.CFG["n_sample_observed_point"], bop_synth.CFG["img_size"]
    assert it["pts"].shape == (2, n, 3) and it["rgb"].shape == (2, 3, s, s) and it["tem1_pts"].shape[1] == 400
    assert it["rgb_choose"].dtype == torch.int64 and int(it["rgb_choose"].max()) < s * s and int(it["rgb_choose"].min()) >= 0
    assert (it["pts"][..., 2] > 0).all()  # only valid-depth pixels are back-projected
    # reference points are in the reference camera frame, metres: z = depth_mm * depth_scale / 1000
    assert 0.5 < float(it["tem1_pts"][0, :, 2].mean()) < 1.0
    assert torch.allclose(it["tem1_pose"][:, 3], torch.tensor([0.0, 0, 0, 1]).expand(2, 4))


def test_best_detection_kept_when_all_scores_low(dataset):
    dets = dataset.dets[dataset.det_keys[1]]
    old = dets[0]["score"]
    dets[0]["score"] = 0.05
    try:
        np.random.seed(0)
        it = dataset[1]
    finally:
        dets[0]["score"] = old
    assert it["pts"].shape[0] == 1 and it["inst_ids"].tolist() == [0]


def test_provider_feeds_runner(dataset, tmp_path):
    """provider item -> collate -> runner.inference_and_save with a stand-in model: CSV rows carry the
    image's ids, one row per kept instance, translations in millimetres."""
    from unopose_amd.runner import inference_and_save

    class Identity(torch.nn.Module):
        def forward(self, ep):
            B = ep["pts"].shape[0]
            ep["pred_R"] = torch.eye(3).expand(B, 3, 3).clone()
            ep["pred_t"] = ep["pts"].mean(1)
            ep["pred_pose_score"] = torch.ones(B)
            return ep

    np.random.seed(3)
    images = [P.collate_image(dataset[i]) for i in range(len(dataset))]
    lines = inference_and_save(Identity(), images, str(tmp_path / "res.csv"), instance_batch_size=1)
    assert len(lines) == 3
    f = lines[0].split(",")
    assert f[:3] == ["48", "1", "2"] and abs(float(f[3]) - 0.9) < 1e-6 and len(f[4].split()) == 9 and len(f[5].split()) == 3
    assert os.path.exists(str(tmp_path / "res.json"))
