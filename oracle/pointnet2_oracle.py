"""ctypes front-end of ``libunopose_oracle.so`` -- TEST INFRASTRUCTURE ONLY.

Presents the nine operators of the reference's pybind module
``core.unopose.model.pointnet2._ext`` (``_ext_src/src/bindings.cpp:11-24``) on
CPU torch tensors, with the reference host wrappers' allocation semantics
(zero-filled outputs, ``tmp`` = 1e10 for FPS: ``sampling.cpp:70-91``,
``ball_query.cpp:13-37``, ``group_points.cpp:17-65``, ``interpolate.cpp``).
The object ``ext`` can be bound as ``pointnet2_utils._ext`` when the reference
Python is imported to generate golden fixtures.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libunopose_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "pointnet2_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libunopose_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _fp(t):
    assert t.dtype == torch.float32 and t.is_contiguous() and t.device.type == "cpu"
    return ctypes.cast(t.data_ptr(), ctypes.POINTER(ctypes.c_float))


def _ip(t):
    assert t.dtype == torch.int32 and t.is_contiguous() and t.device.type == "cpu"
    return ctypes.cast(t.data_ptr(), ctypes.POINTER(ctypes.c_int))


class _Ext:
    """Same names / argument order / return types as the reference ``_ext``."""

    @staticmethod
    def furthest_point_sampling(points, nsamples):
        b, n, _ = points.shape
        out = torch.zeros(b, nsamples, dtype=torch.int32)
        tmp = torch.full((b, n), 1e10, dtype=torch.float32)
        lib().oracle_furthest_point_sampling(b, n, int(nsamples), _fp(points), _fp(tmp), _ip(out))
        return out

    @staticmethod
    def gather_points(points, idx):
        b, c, n = points.shape
        m = idx.shape[1]
        out = torch.zeros(b, c, m, dtype=torch.float32)
        lib().oracle_gather_points(b, c, n, m, _fp(points), _ip(idx), _fp(out))
        return out

    @staticmethod
    def gather_points_grad(grad_out, idx, n):
        b, c, m = grad_out.shape
        out = torch.zeros(b, c, int(n), dtype=torch.float32)
        lib().oracle_gather_points_grad(b, c, int(n), m, _fp(grad_out), _ip(idx), _fp(out))
        return out

    @staticmethod
    def ball_query(new_xyz, xyz, radius, nsample):
        b, m, _ = new_xyz.shape
        n = xyz.shape[1]
        idx = torch.zeros(b, m, int(nsample), dtype=torch.int32)
        lib().oracle_ball_query(b, n, m, ctypes.c_float(radius), int(nsample), _fp(new_xyz), _fp(xyz), _ip(idx))
        return idx

    @staticmethod
    def group_points(points, idx):
        b, c, n = points.shape
        _, npoints, nsample = idx.shape
        out = torch.zeros(b, c, npoints, nsample, dtype=torch.float32)
        lib().oracle_group_points(b, c, n, npoints, nsample, _fp(points), _ip(idx), _fp(out))
        return out

    @staticmethod
    def group_points_grad(grad_out, idx, n):
        b, c, npoints, nsample = grad_out.shape
        out = torch.zeros(b, c, int(n), dtype=torch.float32)
        lib().oracle_group_points_grad(b, c, int(n), npoints, nsample, _fp(grad_out), _ip(idx), _fp(out))
        return out

    @staticmethod
    def three_nn(unknown, known):
        b, n, _ = unknown.shape
        m = known.shape[1]
        dist2 = torch.zeros(b, n, 3, dtype=torch.float32)
        idx = torch.zeros(b, n, 3, dtype=torch.int32)
        lib().oracle_three_nn(b, n, m, _fp(unknown), _fp(known), _fp(dist2), _ip(idx))
        return [dist2, idx]

    @staticmethod
    def three_interpolate(points, idx, weight):
        b, c, m = points.shape
        n = idx.shape[1]
        out = torch.zeros(b, c, n, dtype=torch.float32)
        lib().oracle_three_interpolate(b, c, m, n, _fp(points), _ip(idx), _fp(weight), _fp(out))
        return out

    @staticmethod
    def three_interpolate_grad(grad_out, idx, weight, m):
        b, c, n = grad_out.shape
        out = torch.zeros(b, c, int(m), dtype=torch.float32)
        lib().oracle_three_interpolate_grad(b, c, n, int(m), _fp(grad_out), _ip(idx), _fp(weight), _fp(out))
        return out


ext = _Ext()
