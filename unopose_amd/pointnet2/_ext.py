"""Drop-in for the reference's pybind module ``core.unopose.model.pointnet2._ext``.

Same nine functions, argument order, dtypes, shapes and error behaviour as
``_ext_src/src/bindings.cpp:11-24`` and the host wrappers
(``sampling.cpp``, ``ball_query.cpp``, ``group_points.cpp``, ``interpolate.cpp``):
inputs must be contiguous float32 / int32 device tensors (else ``RuntimeError``),
CPU tensors raise ``RuntimeError("CPU not supported")``, outputs are freshly
allocated (zero-filled where the reference zero-fills and the kernel does not
write every element).  Kernels run on torch's current stream through the C ABI
of ``libunopose_hip.so``.
"""
import torch

from .._lib import call, check_f32, check_i32, on_device, ptr, stream_ptr


def furthest_point_sampling(points, nsamples):
    check_f32(points, "points")
    B, N, _ = points.shape
    if B == 0 or nsamples == 0 or N == 0:  # nothing to launch (the reference returns zeros)
        return torch.zeros(B, nsamples, dtype=torch.int32, device=points.device)
    # (the kernel writes every index, out[0] = 0 included: no zero fill -- a 67-us launch in front of each FPS chain when the GPU is busy)
    out = torch.empty(B, nsamples, dtype=torch.int32, device=points.device)
    with on_device(points.device):
        call("unopose_furthest_point_sampling", ptr(points), B, N, int(nsamples), ptr(out), stream_ptr())
    return out


def gather_points(points, idx):
    check_f32(points, "points")
    check_i32(idx, "idx")
    B, C, N = points.shape
    M = idx.shape[1]
    out = torch.empty(B, C, M, dtype=torch.float32, device=points.device)
    if out.numel() == 0:  # empty batch: nothing to launch
        return out
    with on_device(points.device):
        call("unopose_gather_points", ptr(points), ptr(idx), B, C, N, M, ptr(out), stream_ptr())
    return out


def gather_points_grad(grad_out, idx, n):
    check_f32(grad_out, "grad_out")
    check_i32(idx, "idx")
    B, C, M = grad_out.shape
    out = torch.zeros(B, C, int(n), dtype=torch.float32, device=grad_out.device)
    if out.numel() == 0:  # empty batch: nothing to launch
        return out
    with on_device(grad_out.device):
        call("unopose_gather_points_grad", ptr(grad_out), ptr(idx), B, C, int(n), M, ptr(out), stream_ptr())
    return out


def ball_query(new_xyz, xyz, radius, nsample):
    check_f32(new_xyz, "new_xyz")
    check_f32(xyz, "xyz")
    B, M, _ = new_xyz.shape
    N = xyz.shape[1]
    idx = torch.empty(B, M, int(nsample), dtype=torch.int32, device=new_xyz.device)
    if idx.numel() == 0:  # empty batch: nothing to launch
        return idx
    with on_device(new_xyz.device):
        call("unopose_ball_query", ptr(new_xyz), ptr(xyz), B, N, M, float(radius), int(nsample), ptr(idx),
             stream_ptr())
    return idx


def group_points(points, idx):
    check_f32(points, "points")
    check_i32(idx, "idx")
    B, C, N = points.shape
    _, M, S = idx.shape
    out = torch.empty(B, C, M, S, dtype=torch.float32, device=points.device)
    if out.numel() == 0:  # empty batch: nothing to launch
        return out
    with on_device(points.device):
        call("unopose_group_points", ptr(points), ptr(idx), B, C, N, M, S, ptr(out), stream_ptr())
    return out


def group_points_grad(grad_out, idx, n):
    check_f32(grad_out, "grad_out")
    check_i32(idx, "idx")
    B, C, M, S = grad_out.shape
    out = torch.zeros(B, C, int(n), dtype=torch.float32, device=grad_out.device)
    if out.numel() == 0:  # empty batch: nothing to launch
        return out
    with on_device(grad_out.device):
        call("unopose_group_points_grad", ptr(grad_out), ptr(idx), B, C, int(n), M, S, ptr(out), stream_ptr())
    return out


def three_nn(unknowns, knows):
    check_f32(unknowns, "unknowns")
    check_f32(knows, "knows")
    B, n, _ = unknowns.shape
    m = knows.shape[1]
    dist2 = torch.zeros(B, n, 3, dtype=torch.float32, device=unknowns.device)
    idx = torch.zeros(B, n, 3, dtype=torch.int32, device=unknowns.device)
    if dist2.numel() == 0:  # empty batch / no query points: nothing to launch
        return [dist2, idx]
    with on_device(unknowns.device):
        call("unopose_three_nn", ptr(unknowns), ptr(knows), B, n, m, ptr(dist2), ptr(idx), stream_ptr())
    return [dist2, idx]


def three_interpolate(points, idx, weight):
    check_f32(points, "points")
    check_i32(idx, "idx")
    check_f32(weight, "weight")
    B, c, m = points.shape
    n = idx.shape[1]
    out = torch.empty(B, c, n, dtype=torch.float32, device=points.device)
    if out.numel() == 0:  # empty batch: nothing to launch
        return out
    with on_device(points.device):
        call("unopose_three_interpolate", ptr(points), ptr(idx), ptr(weight), B, c, m, n, ptr(out), stream_ptr())
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    check_f32(grad_out, "grad_out")
    check_i32(idx, "idx")
    check_f32(weight, "weight")
    B, c, n = grad_out.shape
    out = torch.zeros(B, c, int(m), dtype=torch.float32, device=grad_out.device)
    if out.numel() == 0:  # empty batch: nothing to launch
        return out
    with on_device(grad_out.device):
        call("unopose_three_interpolate_grad", ptr(grad_out), ptr(idx), ptr(weight), B, c, n, int(m), ptr(out),
             stream_ptr())
    return out
