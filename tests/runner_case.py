"""Deterministic inputs + stand-in model for the runner capture (tests/golden/make_runner_golden.py runs the
REFERENCE loop on them; tests/test_runner_cpu.py runs unopose_amd.runner on the same and compares)."""
import torch


class StubModel:
    """UNOPose.forward's output contract as a pure function of the inputs (fp32, irrational-looking values so the
    str() formatting of every digit is exercised)."""

    def __call__(self, ep):
        B = ep["pts"].shape[0]
        c = ep["pts"].mean(dim=(1, 2))
        ang = c * 3.0
        R = torch.zeros(B, 3, 3)
        R[:, 0, 0] = torch.cos(ang)
        R[:, 0, 1] = -torch.sin(ang)
        R[:, 1, 0] = torch.sin(ang)
        R[:, 1, 1] = torch.cos(ang)
        R[:, 2, 2] = 1.0
        ep["pred_R"] = R
        ep["pred_t"] = torch.stack([c * 0.37, c * -1.21 + 0.05, 0.8 + c], 1)
        ep["pred_pose_score"] = torch.sigmoid(ep["tem1_pts"].mean(dim=(1, 2)) + c)
        return ep


def make_case(n_img=5, seed=11):
    """-> (images: list of batch-dim-1 dicts as the DataLoader yields them, dets: {"<scene>_<img>": [detection dicts]})."""
    g = torch.Generator().manual_seed(seed)
    images, dets = [], {}
    for i in range(n_img):
        n_det = 2 + (i * 3) % 7  # detections in the file for this image
        keep = [k for k in range(n_det) if (k + i) % 3 != 0] or [0]  # the provider dropped some (score / depth filters)
        n = len(keep)
        scene_id, img_id = 48 + i // 2, 10 * i + 1
        pose = torch.eye(4).repeat(n, 1, 1)
        pose[:, :3, 3] = torch.randn(n, 3, generator=g) * 0.1
        pose[:, 0, 1], pose[:, 1, 0] = 0.6, -0.6
        pose[:, 0, 0] = pose[:, 1, 1] = 0.8
        images.append(dict(
            pts=torch.randn(1, n, 32, 3, generator=g) * 0.1, rgb=torch.zeros(1, n, 3, 4, 4), rgb_choose=torch.zeros(1, n, 32, dtype=torch.long),
            tem1_rgb=torch.zeros(1, n, 3, 4, 4), tem1_choose=torch.zeros(1, n, 40, dtype=torch.long),
            tem1_pts=torch.randn(1, n, 40, 3, generator=g) * 0.1, tem1_pose=pose[None], score=torch.rand(1, n, 1, generator=g),
            obj_id=torch.randint(1, 22, (1, n, 1), generator=g, dtype=torch.int32), scene_id=torch.IntTensor([scene_id]),
            img_id=torch.IntTensor([img_id]), inst_ids=torch.IntTensor([keep]), seg_time=torch.FloatTensor([0.25 + 0.01 * i])))
        dets[f"{scene_id:06d}_{img_id:06d}"] = [
            dict(scene_id=scene_id, image_id=img_id, category_id=int(1 + (k * 5 + i) % 21), score=round(0.3 + 0.1 * k, 3), time=0.25 + 0.01 * i,
                 bbox=[k, 2 * k, 10 + k, 12 + k], segmentation={"size": [4, 6], "counts": [3, 5, 16]}) for k in range(n_det)]
    return images, dets
