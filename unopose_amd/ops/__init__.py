"""Fused operators of the MI355X UNOPose forward (C ABI part 2) on torch tensors, by family:
    dense      nn.Linear on the hand-written GEMMs, LayerNorm / LayerScale glue, the residual + LayerNorm fold, the trainable form
    attention  ViT attention, RPE / cross token attention, focused linear attention
    geometry   frames, Procrustes, radius normalisation, gathers, geometric embedding, positional encoding
    sampling   pixel sampling and the sparse up-projection
    pose       similarity, assignment, coarse / fine pose heads
    train      autograd Functions of the training step
Each function cites the reference Python it replaces.  Inputs must be CUDA tensors; there is no CPU path (RuntimeError).
`from unopose_amd import ops; ops.linear(...)` keeps working as before the split; the A/B switches (ops.USE_LN_FOLD = False, ...) live in
`ops._state` and assignments on the package are forwarded there."""
import sys
import types

from . import _state
from .common import (  # noqa: F401
    _c, differentiable, is_differentiable, _fallbacks_seen, note_fallback, _params_key, _SPLIT_MEMO, _MUTATION_EPOCH, note_mutation,
    clear_split_memo, _aligned16, _no_autograd, _f32_path, _own_f32, _own_glue,
)
from .dense import (  # noqa: F401
    linear_backend, _bf16_weights, linear_bf16_hip, own_gemm_ok, bf16_linear_2d, f32x3_ok, split_f32, _f32x3_weights, linear_f32x3,
    linear_f32x3_bf16, linear_f32_raw, _transposed_weights, _LinearFn, _ZERO_BIAS, _zero_bias, linear_train, _lin, mlp, linear,
    ffn_add_layernorm, linear_add_layernorm, patch_embed, vit_prologue_ok, vit_prologue, vit_prologue_f32_ok, vit_prologue_f32, bmm_nt_f32, score_head, add_layernorm,
    scale_residual_, scale_residual_layernorm_f32_, vit_f32_fused_ok, scale_residual_layernorm_, ln_fold_ok, _fold_producer_weights,
    _fold_consumer_weights, linear_residual_, linear_lnfold, _into,
)
from .attention import (  # noqa: F401
    vit_attention, vit_attention_f32_split, vit_attention_f32_ss, vit_attention_torch, _KEY_PAD, token_attention,
    _token_attention_hip_f32, _attn_weights, _token_attention_hip, token_attention_torch, focused_linear_attention,
    _focused_linear_attention_hip_f32, _focused_linear_attention_hip, focused_linear_attention_torch,
)
from .geometry import (  # noqa: F401
    lrf_global, query_lrf_group, lrf_group_idx, weighted_procrustes, cloud_radius, scale_by_radius, gather_rows, pairwise_distance,
    _bf16_split, _mfma_fragment_order, _GEO_HINV, _GEO_D_RANGE, _GEO_TABLE_UNAVAILABLE, _geo_tables, geo_embedding, geo_embedding_torch,
    pe_group_mlp_max, pe_group_mlp_max_unfused, furthest_point_sample, _geo_grid, _GeoEmbedFn, geo_embedding_train_ok,
)
from .sampling import (  # noqa: F401
    bilinear_sample_native, sparse_upproj_ok, upproj_plan, sparse_pixel_features, bilinear_sample_pixels,
)
from .pose import (  # noqa: F401
    overlap_scores, set_first_rows_, pose_score, rigid_rows, feature_similarity, soft_assignment, coarse_pose_torch, fine_pose_torch, _assign_labels,
    coarse_pose, fine_pose, fine_pose_fused_ok, normalize_rows_bf16, normalize_rows_f32, fine_pose_from_features,
)
from .train import (  # noqa: F401
    _InfoNCEFn, infonce_two_way, _BNReLUTrain, bn_relu, _BNReLUMaxPoolTrain, bn_relu_maxpool, _SaliencyFn, saliency_pair,
    nearest_partner, _CONV_FWD_PAIRS, _CONV_WGRAD_PAIRS, _conv1x1_pair_ok, _conv1x1_wgrad_ok, _Conv1x1Fn, conv1x1,
)

_SWITCHES = frozenset(['FORBID_LIBRARY_BF16_GEMM', 'GEO_TABLE', 'GEO_TABLE_F32', 'HIP_GEMM_ALL', 'PIXEL_FEATS_BF16', 'TRAIN_FUSED_SALIENCY', 'TRAIN_OWN_CONV', 'TRAIN_OWN_GEMM', 'TRAIN_OWN_GEMM_MIN_FLOP', 'TRAIN_OWN_GEO', 'TRAIN_OWN_WGRAD', 'TRAIN_OWN_WGRAD_MIN_ROWS', 'USE_F32X3', 'USE_FUSED_BN_RELU', 'USE_FUSED_FINE', 'USE_FUSED_INFONCE', 'USE_FUSED_LINEAR_LN', 'USE_HIP_GEMM', 'USE_KV_VT', 'USE_LA_KV_STATE', 'USE_LN_FOLD', 'USE_OWN_TOPK', 'USE_SPARSE_UPPROJ', 'USE_STACKED_OUT', '_DIFF'])


class _OpsModule(types.ModuleType):
    """`ops.SWITCH` reads and `ops.SWITCH = value` writes go to `ops._state`, where the family modules read them at call time."""

    def __getattr__(self, name):
        if name in _SWITCHES:
            return getattr(_state, name)
        raise AttributeError(f"module {self.__name__!r} has no attribute {name!r}")

    def __setattr__(self, name, value):
        if name in _SWITCHES:
            setattr(_state, name, value)
        else:
            super().__setattr__(name, value)

    def __delattr__(self, name):
        if name in _SWITCHES:
            raise AttributeError(f"{name} is a switch of ops._state and cannot be deleted")
        super().__delattr__(name)


sys.modules[__name__].__class__ = _OpsModule
