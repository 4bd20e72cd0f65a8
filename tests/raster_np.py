"""Test infrastructure: a numpy z-buffer rasteriser with the arithmetic of csrc/raster.hip in float32, operation for operation
(same expression order, no fused multiply-adds; division correctly rounded on both sides) -- the HIP rasteriser must reproduce its depth
maps bit for bit.  It is also the `renderer` handed to bop_toolkit_lib.pose_error.vsd when the VSD golden values are generated
(tests/golden/make_bop_eval_golden.py), so that the toolkit and unopose_amd.bop_eval score the SAME rendered depth."""
import numpy as np

f32 = np.float32


def render_depth(verts, faces, R, t, fx, fy, cx, cy, H, W):
    verts, R, t = np.asarray(verts, f32), np.asarray(R, f32).reshape(3, 3), np.asarray(t, f32).reshape(3)
    fx, fy, cx, cy = f32(fx), f32(fy), f32(cx), f32(cy)
    q = verts
    X = R[0, 0] * q[:, 0] + R[0, 1] * q[:, 1] + R[0, 2] * q[:, 2] + t[0]
    Y = R[1, 0] * q[:, 0] + R[1, 1] * q[:, 1] + R[1, 2] * q[:, 2] + t[1]
    Z = R[2, 0] * q[:, 0] + R[2, 1] * q[:, 1] + R[2, 2] * q[:, 2] + t[2]
    with np.errstate(divide="ignore", invalid="ignore"):
        U = fx * X / Z + cx
        V = fy * Y / Z + cy
    zbuf = np.full((H, W), np.inf, f32)
    one = f32(1.0)
    for tri in np.asarray(faces):
        z3 = Z[tri]
        if not (z3 > f32(1e-6)).all():
            continue
        u, v = U[tri], V[tri]
        area = (u[1] - u[0]) * (v[2] - v[0]) - (u[2] - u[0]) * (v[1] - v[0])
        if abs(area) < f32(1e-12):
            continue
        x0, x1 = max(0, int(np.ceil(u.min()))), min(W - 1, int(np.floor(u.max())))
        y0, y1 = max(0, int(np.ceil(v.min()))), min(H - 1, int(np.floor(v.max())))
        if x1 < x0 or y1 < y0:
            continue
        inv = one / area
        iz = one / z3
        px, py = np.meshgrid(np.arange(x0, x1 + 1, dtype=f32), np.arange(y0, y1 + 1, dtype=f32))
        b0 = ((u[1] - px) * (v[2] - py) - (u[2] - px) * (v[1] - py)) * inv
        b1 = ((u[2] - px) * (v[0] - py) - (u[0] - px) * (v[2] - py)) * inv
        b2 = one - b0 - b1
        inside = (b0 >= 0) & (b1 >= 0) & (b2 >= 0)
        with np.errstate(divide="ignore"):
            z = one / (b0 * iz[0] + b1 * iz[1] + b2 * iz[2])
        sub = zbuf[y0:y1 + 1, x0:x1 + 1]
        upd = inside & (z > 0) & (z < sub)
        sub[upd] = z[upd]
    zbuf[np.isinf(zbuf)] = 0
    return zbuf


class NumpyRenderer:
    """bop_toolkit renderer interface on top of `render_depth`."""

    def __init__(self, width, height):
        self.W, self.H, self.models = width, height, {}

    def add_object(self, obj_id, verts, faces):
        self.models[obj_id] = (np.asarray(verts, f32), np.asarray(faces, np.int32))

    def render_object(self, obj_id, R, t, fx, fy, cx, cy):
        v, f = self.models[obj_id]
        return {"depth": render_depth(v, f, R, np.asarray(t).reshape(3), fx, fy, cx, cy, self.H, self.W)}
