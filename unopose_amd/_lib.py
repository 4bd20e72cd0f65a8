"""ctypes binding of ``libunopose_hip.so`` (the C ABI in ``include/unopose_hip.h``).

The product path has NO fallback: if the shared library is missing or an entry
point returns non-zero, a ``RuntimeError`` is raised (the reference's wrappers
raise ``RuntimeError`` through TORCH_CHECK, ``_ext_src/include/utils.h:10-30``).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# UNOPOSE_LIB: an alternative build of the SAME library (scripts/build_variant.py: same sources, other compiler flags) for same-box A/Bs
SO_PATH = os.environ.get("UNOPOSE_LIB") or os.path.join(_HERE, "libunopose_hip.so")

_lib = None

_P = ctypes.c_void_p
_I = ctypes.c_int
_F = ctypes.c_float
_L = ctypes.c_long

# name -> argtypes (return type is always int).  Must list every symbol that
# include/unopose_hip.h declares; tests/test_abi.py checks the two agree.
SIGNATURES = {
    "unopose_furthest_point_sampling": [_P, _I, _I, _I, _P, _P],
    "unopose_gather_points": [_P, _P, _I, _I, _I, _I, _P, _P],
    "unopose_gather_points_grad": [_P, _P, _I, _I, _I, _I, _P, _P],
    "unopose_ball_query": [_P, _P, _I, _I, _I, _F, _I, _P, _P],
    "unopose_group_points": [_P, _P, _I, _I, _I, _I, _I, _P, _P],
    "unopose_group_points_grad": [_P, _P, _I, _I, _I, _I, _I, _P, _P],
    "unopose_three_nn": [_P, _P, _I, _I, _I, _P, _P, _P],
    "unopose_three_interpolate": [_P, _P, _P, _I, _I, _I, _I, _P, _P],
    "unopose_three_interpolate_grad": [_P, _P, _P, _I, _I, _I, _I, _P, _P],
    "unopose_lrf_global": [_P, _I, _I, _I, _P, _P],
    "unopose_query_lrf_group": [_P, _I, _I, _F, _I, _P, _P],
    "unopose_lrf_group_idx": [_P, _P, _P, _I, _I, _F, _I, _P, _P],
    "unopose_weighted_procrustes": [_P, _P, _P, _I, _I, _F, _F, _P, _P, _P],
    "unopose_assign_labels": [_P, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "unopose_linear_bf16_ld": [_P, _I, _P, _I, _P, _P, _I, ctypes.c_long, _I, _I, _I, _P],
    "unopose_scale_residual_layernorm_f32": [_P, _P, _P, _P, _P, ctypes.c_long, _I, _F, _P, _L, _P],
    "unopose_gather_rows": [_P, _I, _I, _I, _P, _I, _I, _I, _P, _L, _I, _P, _P],
    "unopose_softmax_stats": [_P, _I, _I, _I, _P, _P],
    "unopose_infonce_grad": [_P, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "unopose_fine_correspondences": [_P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "unopose_upproj_plan": [_P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    "unopose_linear_add_layernorm_bf16": [_P, _P, _P, _P, _P, _P, _F, _P, ctypes.c_long, _I, _P],
    "unopose_linear_bf16_kv_vt": [_P, _P, _P, _P, _P, ctypes.c_long, _I, _I, _I, _I, _P],
    "unopose_linear_bf16_gather": [_P, ctypes.c_long, _I, _P, _I, _P, _P, _P, _I, _P, _P],
    "unopose_bilinear_sample_compact": [_P, _P, _P, _I, _I, _I, _I, _I, _P, _I, _P],
    "unopose_fine_assign": [_P, _P, _I, _I, _I, _I, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "unopose_min_dist": [_P, _P, _I, _I, _I, _P, _P, _I, _P, _P],
    "unopose_coarse_hypotheses": [_P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P],
    "unopose_coarse_scores": [_P, _P, _I, _I, _I, _P, _P, _I, _P, _I, _P, _P, _P],
    "unopose_topk_smallest": [_P, _I, _I, _I, _P, _P],
    "unopose_coarse_pick": [_P, _P, _I, _I, _P, _P, _I, _P, _P, _P, _P],
    "unopose_token_attention": [_P, _I, _P, _I, _P, _P, _I, _P, _I, _I, _I, _F, _P, _P],
    "unopose_token_attention_key_pad": [],
    "unopose_token_attention_f32": [_P, _I, _P, _I, _P, _P, _I, _P, _I, _I, _I, _F, _P, _P],
    "unopose_vit_attention_f32": [_P, _I, _I, _I, _P, _P],
    "unopose_vit_attention_f32_split": [_P, _I, _I, _I, _P, _P],
    "unopose_vit_attention_f32_ss": [_P, _I, _I, _I, _P, _P],
    "unopose_bn_train_chunk": [],
    "unopose_bn_relu_train_forward": [_P, _I, _I, ctypes.c_long, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P],
    "unopose_bn_relu_train_backward": [_P, _P, _I, _I, ctypes.c_long, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "unopose_bn_relu_maxpool_train_forward": [_P, _I, _I, _I, _I, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P],
    "unopose_bn_relu_maxpool_train_backward": [_P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "unopose_nearest_partner": [_P, _P, _I, _I, _I, _I, _F, _P, _P, _P, _P],
    "unopose_saliency_train_forward": [_P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P],
    "unopose_saliency_train_backward": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P],
    "unopose_conv1x1_train_wgrad_blocks": [],
    "unopose_conv1x1_train_forward": [_P, _I, _I, ctypes.c_long, _P, _I, _P, _P],
    "unopose_conv1x1_train_wgrad": [_P, _P, _I, _I, _I, ctypes.c_long, _P, _P, _P],
    "unopose_linear_wgrad_f32_splits": [ctypes.c_long, _I, _I],
    "unopose_linear_wgrad_f32": [_P, _P, ctypes.c_long, _I, _I, _P, _P, _P],
    "unopose_vit_attention": [_P, _I, _I, _I, _P, _P],
    "unopose_add_layernorm": [_P, _I, _P, _I, _P, _P, ctypes.c_long, _I, _F, _P, _I, _P],
    "unopose_add_layernorm_strided": [_P, _I, _P, _I, _P, _P, ctypes.c_long, _I, _F, _P, _I, ctypes.c_long, _P],
    "unopose_bilinear_sample": [_P, _I, _P, _I, _I, _I, _I, _I, _P, _P],
    "unopose_bilinear_sample_tokens": [_P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P],
    "unopose_scale_residual_layernorm": [_P, _P, _P, _P, _P, ctypes.c_long, _I, _F, _P, _P],
    "unopose_scale_residual": [_P, _P, _P, ctypes.c_long, _I, _P],
    "unopose_linear_attention": [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P],
    "unopose_linear_attention_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P],
    "unopose_linear_attention_kv_state": [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P],
    "unopose_linear_bf16": [_P, _P, _P, _P, ctypes.c_long, _I, _I, _I, _P],
    "unopose_gemm_bf16_tile": [],
    "unopose_gemm_fold_stagger": [_I],
    "unopose_linear_bf16_residual": [_P, _P, _P, _P, _P, _P, ctypes.c_long, _I, _I, _P],
    "unopose_linear_bf16_lnfold": [_P, _P, _P, _P, _P, _I, _F, _P, ctypes.c_long, _I, _I, _I, _P],
    "unopose_patchify_bf16": [_P, _I, _P, _I, _I, _I, _P, _P],
    "unopose_patchify_split": [_P, _I, _P, _I, _I, _I, _P, _P],
    "unopose_vit_tokens_layernorm": [_P, _P, _P, _I, _I, _I, _I, _P, _P, _F, _P, _P, _P],
    "unopose_vit_tokens_layernorm_f32": [_P, _P, _P, _I, _I, _I, _I, _P, _P, _F, _P, _P, _P],
    "unopose_row_dot": [_P, _I, _P, _F, ctypes.c_long, _I, _P, _I, _P],
    "unopose_normalize_rows_bf16": [_P, _I, ctypes.c_long, _I, _F, _P, _I, _P],
    "unopose_transpose_pad_bf16": [_P, ctypes.c_long, _I, _I, _I, _I, _P, _P],
    "unopose_transpose_pad_f32": [_P, _L, _I, _I, _I, _I, _P, _P],
    "unopose_cloud_radius": [_P, _I, _I, _P, _P],
    "unopose_scale_by_radius": [_P, _I, _I, _P, _F, _I, _P, _P],
    "unopose_overlap_scores": [_P, _I, _I, _I, _I, _I, _P, _P],
    "unopose_copy_rows": [_P, _L, _P, _L, _I, _I, _P],
    "unopose_nan_to_num_multi": [_P, _P, _I, _L, _F, _F, _F, _P],
    "unopose_rigid_rows_bf16": [_P, _I, _I, _P, _P, _P, _P],
    "unopose_token_sum_bf16": [_P, _I, _I, _I, _P, _P],
    "unopose_pose_score": [_P, _P, _I, _I, _F, _P, _P],
    "unopose_render_depth": [_P, _I, _P, _I, _P, _P, _I, _I, _I, _P, _P],
    "unopose_bmm_f32": [_P, ctypes.c_long, ctypes.c_long, ctypes.c_long, ctypes.c_long, _P, ctypes.c_long, ctypes.c_long, ctypes.c_long, ctypes.c_long, _P, _I, _I, _I, _I, _I, _F, _P],
    "unopose_split_bf16x2": [_P, ctypes.c_long, _I, _P, _P],
    "unopose_linear_f32x3": [_P, _P, _P, _P, _P, ctypes.c_long, _I, _I, _I, _P],
    "unopose_pe_image_bytes": [],
    "unopose_pe_pack_weights": [_P, _P, _P, _P, _P, _P, _P, _P],
    "unopose_pe_group_mlp_max_packed": [_P, _I, _I, _F, _I, _P, _P, _P],
    "unopose_pe_group_mlp_max_packed_cand": [_P, _I, _I, _F, _I, _P, _P, _P, _I, _P, _P, _P, _P],
    "unopose_pe_group_mlp_max_packed_out": [_P, _I, _I, _F, _I, _P, _P, _P, _I, _P, _P, _P, _I, _I, _P],
    "unopose_linear_f32x3_bf16": [_P, _P, _P, _P, _P, ctypes.c_long, _I, _I, _P],
    "unopose_pe_group_mlp_max": [_P, _I, _I, _F, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P],
    "unopose_geo_embedding": [_P, _I, _I, _P, _P, _P, _P, _P, _P, _F, _F, _I, _I, _I, _P, _P, _P],
    "unopose_geo_embedding_train_workgroups": [_I, _I],
    "unopose_geo_embedding_train_forward": [_P, _I, _I, _P, _I, _P, _I, _P, _P, _P, _I, _F, _F, _I, _P, _P, _P, _P],
    "unopose_geo_embedding_train_backward": [_P, _P, _I, _I, _I, _I, _I, _F, _F, _I, _P, _P, _P, _P, _P, _P],
    "unopose_geo_embedding_table": [_P, _I, _I, _P, _I, _P, _I, _P, _P, _P, _I, _I, _F, _F, _I, _I, _P, _P, _P],
}


def lib():
    """Load the shared library once; raise loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                f"{SO_PATH} is missing: the HIP extension has not been built "
                "(run `python -m unopose_amd.build`). There is no CPU fallback."
            )
        L = ctypes.CDLL(SO_PATH)
        L.unopose_abi_version.restype = _I
        L.unopose_last_error.restype = ctypes.c_char_p
        for name, argtypes in SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = argtypes
            fn.restype = _I
        _lib = L
    return _lib


def stream_ptr(device=None):
    """Raw hipStream_t of torch's current stream on `device` (default: the current device).  Through torch's C hook: `torch.cuda.current_stream()`
    builds a Stream object and resolves the device index in Python, ~4.5 us per call -- 1.3 ms of the 7.3 ms a forward takes to ENQUEUE at the
    reference's contract shape (16 x 224 x 224), where the host bounds the pipelined step (scripts/host_profile.py, round 6)."""
    if device is None:
        idx = torch._C._cuda_getDevice()
    else:
        idx = device if isinstance(device, int) else torch.device(device).index
        if idx is None:
            idx = torch._C._cuda_getDevice()
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(idx))


class _NullContext:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NULL_CONTEXT = _NullContext()


def on_device(dev):
    """`with on_device(t.device):` = `with torch.cuda.device(t.device):` when that device is not the current one, nothing otherwise (the common
    case: one process per GPU).  The torch context manager costs ~8 us per use on the host; the wrappers enter it once per kernel call."""
    idx = dev.index if isinstance(dev, torch.device) else dev
    if idx is None or idx == torch._C._cuda_getDevice():
        return _NULL_CONTEXT
    return torch.cuda.device(idx)


def call(name, *args):
    L = lib()
    rc = getattr(L, name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed (code {rc}): {L.unopose_last_error().decode()}")


def ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def check_f32(x, name):
    if not x.is_cuda:
        raise RuntimeError(f"{name}: CPU not supported")
    if not x.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")
    if x.dtype != torch.float32:
        raise RuntimeError(f"{name} must be a float tensor")


def check_i32(x, name):
    if not x.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if not x.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")
    if x.dtype != torch.int32:
        raise RuntimeError(f"{name} must be an int tensor")
