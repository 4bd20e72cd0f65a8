"""Depth renderer for BOP's VSD error on the HIP rasteriser (csrc/raster.hip, C ABI unopose_render_depth).

The reference's evaluation (bop_toolkit `eval_bop19_pose.py`, tabulated by core/unopose/engine/bop_eval_utils.py:340-454) hands
`pose_error.vsd` a renderer object with `render_object(obj_id, R, t, fx, fy, cx, cy) -> {"depth": (H,W)}`; this class offers that
call (so it can also be passed to the toolkit itself) plus a batched form for many poses of one object."""
import numpy as np
import torch

from ._lib import call, on_device, ptr, stream_ptr


class HipDepthRenderer:
    def __init__(self, width, height, device="cuda:0"):
        self.W, self.H, self.device = int(width), int(height), torch.device(device)
        self.models = {}

    def add_object(self, obj_id, verts, faces):
        """verts (V,3) in model units (mm for BOP), faces (F,3) vertex indices."""
        v = torch.as_tensor(np.asarray(verts, np.float32)).contiguous().to(self.device)
        f = torch.as_tensor(np.asarray(faces, np.int32)).contiguous().to(self.device)
        assert v.dim() == 2 and v.shape[1] == 3 and f.dim() == 2 and f.shape[1] == 3 and int(f.max()) < v.shape[0]
        self.models[obj_id] = (v, f)

    def render_batch(self, obj_id, Rs, ts, K4):
        """Rs (P,3,3), ts (P,3), K4 (P,4) or (4,) = fx, fy, cx, cy -> depth (P,H,W) float32 tensor on the device."""
        v, f = self.models[obj_id]
        Rs = torch.as_tensor(np.asarray(Rs, np.float32)).reshape(-1, 9)
        ts = torch.as_tensor(np.asarray(ts, np.float32)).reshape(-1, 3)
        P = Rs.shape[0]
        Rt = torch.cat([Rs, ts], 1).contiguous().to(self.device)
        K4 = torch.as_tensor(np.asarray(K4, np.float32)).reshape(-1, 4)
        K4 = (K4.expand(P, 4) if K4.shape[0] == 1 else K4).contiguous().to(self.device)
        depth = torch.empty(P, self.H, self.W, dtype=torch.float32, device=self.device)
        with on_device(self.device):
            call("unopose_render_depth", ptr(v), v.shape[0], ptr(f), f.shape[0], ptr(Rt), ptr(K4), P, self.H, self.W, ptr(depth),
                 stream_ptr(self.device))
        return depth

    def render_object(self, obj_id, R, t, fx, fy, cx, cy):
        """bop_toolkit's renderer interface."""
        d = self.render_batch(obj_id, np.asarray(R).reshape(1, 3, 3), np.asarray(t).reshape(1, 3), [fx, fy, cx, cy])
        return {"depth": d[0].cpu().numpy()}
