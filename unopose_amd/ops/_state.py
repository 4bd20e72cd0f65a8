"""The A/B switches and the mode flag of `unopose_amd.ops`: plain module attributes (no environment routes in the product), read by the family
modules as `st.NAME` at call time and set from outside as `ops.NAME = value` (the package forwards such assignments here).  Every default is the
product path; the off-positions are same-box A/B references for scripts/ and tests/ (library GEMM, torch composite) -- each default is
pinned by a test (tests/test_pipeline_gpu.py: no library GEMM on the eval path; tests/test_train_gpu.py: the training guards)."""

# ---- differentiable mode (training): inside `with ops.differentiable():` every dispatcher takes its autograd-recorded form
# (own kernels behind autograd Functions where they exist, the op-by-op torch composite otherwise); index-producing kernels (FPS, ball query,
# frames), whose outputs carry no gradient in the reference either, keep running on HIP.  UNOPose.forward enters it when `model.training`.
_DIFF = False

# ---- linear layers (ops/dense.py)
USE_HIP_GEMM = True  # own bf16 GEMM (csrc/gemm.hip + gemm_small.hip) for the linears; False: hipBLASLt through torch
HIP_GEMM_ALL = True  # every shape the tiling admits on the own GEMM: no library stream-K kernel (inter-workgroup waits) on the path, which
                     # is what makes two forwards in flight safe (pipeline.py)
USE_F32X3 = True  # fp32 linears on csrc/gemm_f32.hip (hi / lo-split bf16 operands, 3 MFMAs per product); False: the library SGEMM
USE_FUSED_LINEAR_LN = True  # linear + residual + LayerNorm of the matcher's layers in one GEMM epilogue
USE_LN_FOLD = True  # round 6: the ViT's residual + LayerNorm passes folded into the GEMM epilogues (csrc/gemm_kernel.h EPI 5 / 6 / 7);
                    # False: scale_residual_layernorm_ between the GEMMs (round 5's path).  A/B: profiles/r06_ln_fold_ab.txt
FORBID_LIBRARY_BF16_GEMM = False  # set by pipeline.PipelinedForward around forwards it overlaps: the library fall-back of `linear` (bf16 and fp32) raises

# ---- geometry / sampling / pose heads
GEO_TABLE = True      # bf16 geometric embedding through the 4-point table kernel; False: the matrix-core kernel
GEO_TABLE_F32 = True  # fp32 result through the 6-point table kernel; False: the split-operand matrix-core kernel
USE_SPARSE_UPPROJ = True  # only the map cells the chosen pixels' bilinear taps read are up-projected (csrc/upproj.hip)
PIXEL_FEATS_BF16 = True  # round 6: the sparse up-projection's pixel features leave the sampling kernel in bf16 (their consumers are autocast Linears); False: fp32 + a cast each
USE_KV_VT = True  # round 6: the token attention's V^T operand written by the k | v projection's epilogue (csrc/gemm_small.hip EPI 4); False: a transpose launch
USE_LA_KV_STATE = True  # round 6: the linear attention's key / value state (focused keys, their sum, k^T v) in one launch (csrc/linattn.hip); False: 7 launches
USE_STACKED_OUT = True  # round 6: a matcher block's two cross layers write the halves of ONE stacked tensor, the dense layer reads its keys / values
                        # behind the background row in place; False: concatenations / slices copied out (4 more launches per block)
USE_OWN_TOPK = True  # round 6: the coarse head's top-k of the hypothesis residuals and the pick of the winner on csrc/posehead.hip; False: torch.topk + max + 3 gathers
USE_FUSED_FINE = True  # bf16 fine stage without the (B, N1 + 1, N2 + 1) similarity (csrc/fineassign.hip)

# ---- training step (ops/train.py, ops/dense.py `_LinearFn`, ops/geometry.py `_GeoEmbedFn`)
USE_FUSED_INFONCE = True  # False: two F.cross_entropy calls
USE_FUSED_BN_RELU = True  # False: nn.BatchNorm2d (MIOpen) + F.relu in the PE's SharedMLP under train()
TRAIN_FUSED_SALIENCY = True  # False: the two softmax + matmul pairs of the reference through torch
TRAIN_OWN_CONV = True  # False: nn.Conv2d (MIOpen) for the PE's 1 x 1 convolutions under train()
TRAIN_OWN_WGRAD = True  # False: the linears' weight gradients through the library (dY^T @ X)
TRAIN_OWN_WGRAD_MIN_ROWS = 16384
TRAIN_OWN_GEMM = True  # False: nn.Linear through the library
# The persistent 256 x 256-tile kernels pay off from a few GFLOP per launch: below this many flops the training step keeps nn.Linear
# (round 6 same-box A/B of the training step: 2e10 121-124 ms, 1e10 119-121, 4e9 118.2, 1e9 117-120: scripts/ubench/train_ab.py)
TRAIN_OWN_GEMM_MIN_FLOP = 4e9
TRAIN_OWN_GEO = True  # round 6: the geometric embedding under autograd on the table kernels; False: the op-by-op composite
