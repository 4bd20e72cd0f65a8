/*
 * unopose_hip.h -- C ABI of libunopose_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the native operators on UNOPose's forward hot path.
 * Every entry point takes plain DEVICE pointers + sizes + a HIP stream handle
 * (hipStream_t passed as void*; NULL = the null stream) and returns 0 on
 * success or a UNOPOSE_E* code.  No torch types cross this boundary.  Kernels
 * are enqueued on `stream`; nothing here synchronises the host.
 *
 * Part 1 replaces, one for one, the nine functions of the reference's pybind
 * module core.unopose.model.pointnet2._ext
 * (core/unopose/model/pointnet2/_ext_src/src/bindings.cpp:11-24).  Unlike the
 * reference host wrappers these do NOT allocate: the caller passes the output
 * buffer (the Python shim unopose_amd/pointnet2/_ext.py allocates it exactly
 * as the reference wrappers do, zero-filled where the reference zero-fills).
 *
 * Part 2 are the fused operators the reference expresses as PyTorch op
 * sequences; each cites the Python it replaces.
 *
 * Layouts are the reference's: row-major, float32 data, int32 indices.
 */
#ifndef UNOPOSE_HIP_H
#define UNOPOSE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UNOPOSE_OK 0
#define UNOPOSE_EINVAL 1   /* bad size / null pointer */
#define UNOPOSE_ELAUNCH 2  /* hipGetLastError() != hipSuccess after launch */
#define UNOPOSE_ENOMEM 3   /* workspace too small */

typedef void *unopose_stream_t;

/* Library / device identification (no GPU work). */
int unopose_abi_version(void);
const char *unopose_last_error(void);

/* ------------------------------------------------------------------ Part 1 */

/* furthest_point_sampling(points (B,N,3), nsamples) -> idx (B,M) int32.
 * Replaces sampling.cpp:70-91 + sampling_gpu.cu:74-234.  idx[b,0] = 0; ties
 * are broken exactly as the reference's 512-lane shared-memory tree does
 * (argmax d, then min (bitrev(k mod bs), k), bs = min(512, 2^floor(log2 N))).
 * The running min-distance array stays on chip; no `tmp` buffer is needed. */
int unopose_furthest_point_sampling(const float *xyz, int B, int N, int M,
                                    int32_t *idx, unopose_stream_t stream);

/* gather_points(points (B,C,N), idx (B,M)) -> out (B,C,M).
 * Replaces sampling.cpp:20-44 + sampling_gpu.cu:13-36. */
int unopose_gather_points(const float *points, const int32_t *idx, int B, int C,
                          int N, int M, float *out, unopose_stream_t stream);

/* gather_points_grad(grad_out (B,C,M), idx (B,M), N) -> grad_points (B,C,N),
 * which MUST be zero-filled by the caller (the reference wrapper allocates
 * zeros).  Replaces sampling.cpp:46-69 + sampling_gpu.cu:39-62. */
int unopose_gather_points_grad(const float *grad_out, const int32_t *idx, int B,
                               int C, int N, int M, float *grad_points,
                               unopose_stream_t stream);

/* ball_query(new_xyz (B,M,3), xyz (B,N,3), radius, nsample) -> idx (B,M,S).
 * Replaces ball_query.cpp:13-37 + ball_query_gpu.cu:14-59.  Every element of
 * idx is written (rows with no hit are written as zeros, which is what the
 * reference's zero-initialised output holds). */
int unopose_ball_query(const float *new_xyz, const float *xyz, int B, int N,
                       int M, float radius, int nsample, int32_t *idx,
                       unopose_stream_t stream);

/* group_points(points (B,C,N), idx (B,M,S)) -> out (B,C,M,S).
 * Replaces group_points.cpp:17-40 + group_points_gpu.cu:13-44. */
int unopose_group_points(const float *points, const int32_t *idx, int B, int C,
                         int N, int M, int S, float *out,
                         unopose_stream_t stream);

/* group_points_grad(grad_out (B,C,M,S), idx, N) -> grad_points (B,C,N),
 * zero-filled by the caller.  Replaces group_points.cpp:42-65 +
 * group_points_gpu.cu:48-80. */
int unopose_group_points_grad(const float *grad_out, const int32_t *idx, int B,
                              int C, int N, int M, int S, float *grad_points,
                              unopose_stream_t stream);

/* three_nn(unknown (B,n,3), known (B,m,3)) -> dist2 (B,n,3), idx (B,n,3).
 * Replaces interpolate.cpp:20-45 + interpolate_gpu.cu:14-74. */
int unopose_three_nn(const float *unknown, const float *known, int B, int n,
                     int m, float *dist2, int32_t *idx, unopose_stream_t stream);

/* three_interpolate(points (B,c,m), idx (B,n,3), weight (B,n,3)) -> (B,c,n).
 * Replaces interpolate.cpp:47-74 + interpolate_gpu.cu:77-116. */
int unopose_three_interpolate(const float *points, const int32_t *idx,
                              const float *weight, int B, int c, int m, int n,
                              float *out, unopose_stream_t stream);

/* three_interpolate_grad(grad_out (B,c,n), idx, weight, m) -> (B,c,m),
 * zero-filled by the caller.  Replaces interpolate.cpp:76-104 +
 * interpolate_gpu.cu:121-159. */
int unopose_three_interpolate_grad(const float *grad_out, const int32_t *idx,
                                   const float *weight, int B, int c, int n,
                                   int m, float *grad_points,
                                   unopose_stream_t stream);

/* ------------------------------------------------------------------ Part 2 */

/* Global local-reference-frame coordinates of a cloud: pts (B,N,3) -> out (B,N,3).
 * Replaces UNOPose.get_batch_lrf + LRF.forward
 * (core/unopose/model/oneref_grf_predator_pose_estimation_model.py:78-93,
 *  core/unopose/utils/model_utils.py:766-823).  use_ref_rad != 0 -> r = 1. */
int unopose_lrf_global(const float *pts, int B, int N, int use_ref_rad,
                       float *out, unopose_stream_t stream);

/* QueryAndLRFGroup.forward with use_xyz=True, use_feature=False, new_xyz == xyz:
 * xyz (B,N,3) -> out (B,6,N,S), channels [p_k - c (3), R^T (p_k - c)/radius (3)].
 * Fuses ball_query + group_points + LRF_batch
 * (core/unopose/model/pointnet2/pointnet2_utils.py:429-481, 522-584). */
int unopose_query_lrf_group(const float *xyz, int B, int N, float radius,
                            int nsample, float *out, unopose_stream_t stream);

/* The general form of QueryAndLRFGroup.forward (new_xyz != xyz and/or sample_uniformly):
 * the neighbour lists idx (B,N,S) int32 are given (ball_query around new_xyz (B,N,3));
 * out (B,6,N,S): channels 0-2 = p_k - new_xyz[j], 3-5 = R^T (p_k - xyz[j]) / radius with
 * the frame of LRF_batch(xyz, grouped) -- the reference passes xyz, not new_xyz, as the
 * frame centres (core/unopose/model/pointnet2/pointnet2_utils.py:548-565), so npoint == N. */
int unopose_lrf_group_idx(const float *xyz, const float *new_xyz, const int *idx,
                          int B, int N, float radius, int nsample, float *out,
                          unopose_stream_t stream);

/* weighted_procrustes(src (M,N,3), ref (M,N,3), w (M,N) or NULL) -> R (M,3,3),
 * t (M,3) with ref ~ R src + t.  Replaces
 * core/unopose/utils/model_utils.py:667-743 (torch.svd -> register Jacobi). */
int unopose_weighted_procrustes(const float *src, const float *ref,
                                const float *w, int M, int N, float thresh,
                                float eps, float *R, float *t,
                                unopose_stream_t stream);

/* Same operator on the bf16 matrix cores with hi/lo-split operands (3 MFMAs per product, ~2^-16
 * relative error).  The folded weights are packed ONCE into an LDS image of unopose_pe_image_bytes()
 * bytes (bf16 hi/lo, k permuted to the MFMA C/D register map, bank-swizzled) that every workgroup
 * copies verbatim. */
int unopose_pe_image_bytes(void);
int unopose_pe_pack_weights(const float *w1, const float *b1, const float *w2, const float *b2,
                            const float *w3, const float *b3, void *image, unopose_stream_t stream);
int unopose_pe_group_mlp_max_packed(const float *xyz, int B, int N, float radius, int nsample,
                                    const void *image, float *out, unopose_stream_t stream);

/* Same with a neighbour-list hand-off between the two scales of PositionalEncoding: a pass may write, per
 * centre, its (padded) neighbour list cand_out (B,N,nsample) int32 and cand_cnt_out (B,N) = number of points
 * inside the radius, or -1 if that exceeded nsample; a later pass over the SAME cloud with a SMALLER radius
 * may read them (cand_in with row stride cand_stride, cand_cnt_in) and test only those candidates instead of
 * scanning all N points.  Results are identical to the full scan.  Any of the pairs may be NULL. */
int unopose_pe_group_mlp_max_packed_cand(const float *xyz, int B, int N, float radius, int nsample,
                                         const void *image, const int *cand_in, const int *cand_cnt_in,
                                         int cand_stride, int *cand_out, int *cand_cnt_out, float *out,
                                         unopose_stream_t stream);
/* The same launch with the output placed by the caller: row stride out_ld (in 4-byte units, >= 128; the row of centre (b, j) starts
 * at out + (b N + j) out_ld * 4 bytes) and out_split = 1 writing the 128 channels in the split layout of unopose_linear_f32x3
 * (4 blocks of [hi | lo] bf16) instead of float32 -- both scales of the positional encoding then land side by side in the
 * (B, N, 256)-wide operand of its Conv1d (oneref_predator_fine_point_matching.py:174) without a concatenation or a split pass. */
int unopose_pe_group_mlp_max_packed_out(const float *xyz, int B, int N, float radius, int nsample, const void *image,
                                        const int *cand_in, const int *cand_cnt_in, int cand_stride, int *cand_out,
                                        int *cand_cnt_out, void *out, int out_ld, int out_split, unopose_stream_t stream);

/* GeometricStructureEmbedding.forward (core/unopose/model/transformer.py:303-350):
 * points (B,n,3) -> out (B,n,n,256), float32 or bfloat16 (out_bf16).  hidden_dim = 256,
 * angle_k = 3.  Weights are passed as bfloat16 bit patterns in MFMA-fragment order
 * [k/16][out/32][(k%16)/8][out%32][k%8] (so one wave-wide operand load is 1 KiB contiguous):
 * w*_hi = bf16(W), w*_lo = bf16(W - float(w*_hi)) (only read when split != 0: hi/lo split
 * keeps fp32-class accuracy on the bf16 matrix cores).  bias_sum = proj_d.bias + proj_a.bias,
 * div_term = the 128 sinusoid frequencies, knn_ws = B*n*3 int32 of scratch. */
int unopose_geo_embedding(const float *points, int B, int n, const void *wd_hi,
                          const void *wd_lo, const void *wa_hi, const void *wa_lo,
                          const float *bias_sum, const float *div_term,
                          float sigma_d, float factor_a, int reduce_mean, int split,
                          int out_bf16, int32_t *knn_ws, void *out,
                          unopose_stream_t stream);

/* The same function (core/unopose/model/transformer.py:303-350) evaluated through tables: proj_d(sinus(x)) and proj_a(sinus(x))
 * are smooth functions of one scalar per output channel, tabulated by the caller -- tab_d (rows_d, 256) and tab_a (rows_a, 256)
 * float32 WITHOUT the biases, row r holding the value at x = (r - (npoint / 2 - 1)) / hinv (hinv must be 4) -- and interpolated with
 * an npoint-point Lagrange polynomial: npoint = 4 (error ~1e-4 of the weights' scale: the bfloat16 result of the autocast forward,
 * 3x faster than the matrix-core kernel above) or 6 (~1e-6: float32 class, the fp32 forward).  A distance index above
 * (rows_d - npoint) / hinv is evaluated from its defining sum with w_d (proj_d.weight, (256, 256) float32 row-major) and div_term;
 * rows_a must cover pi * factor_a (rows_a >= floor(pi factor_a hinv) + npoint + 1). */
int unopose_geo_embedding_table(const float *points, int B, int n, const float *tab_d, int rows_d, const float *tab_a, int rows_a,
                                const float *bias_sum, const float *w_d, const float *div_term, int hinv, int npoint, float sigma_d,
                                float factor_a, int reduce_mean, int out_bf16, int32_t *knn_ws, void *out,
                                unopose_stream_t stream);

/* The table form under autograd (the training step; gradients for proj_d / proj_a of transformer.py:303-350, the points are data):
 * _train_forward = the 6-point float32 evaluation above, which also records per output element WHICH of the three angle terms was the
 * maximum (amax: B*n*n*256 bytes, first maximum on ties; unused with reduce_mean) and keeps the neighbour lists in `knn`;
 * _train_backward scatters dE (B,n,n,256) into table-shaped gradients: ws (workgroups, min(rows_d, 69) + rows_a, 256) float32, one slab per
 * workgroup (unopose_geo_embedding_train_workgroups of them; the caller sums them: rows [0, min(rows_d, 69)) are distance rows, the rest
 * angle rows), full_d (rows_d, 256) float32, ZEROED by the caller, for distance indices past the LDS-resident rows, and *past_table set to
 * 1 if a distance index lies past the table altogether (its gradient is then missing).  dW = dT^T sinus(grid), db = sum_r dT[r]. */
int unopose_geo_embedding_train_workgroups(int B, int n);
int unopose_geo_embedding_train_forward(const float *points, int B, int n, const float *tab_d, int rows_d, const float *tab_a, int rows_a,
                                        const float *bias_sum, const float *w_d, const float *div_term, int hinv, float sigma_d, float factor_a,
                                        int reduce_mean, int32_t *knn, float *out, void *amax, unopose_stream_t stream);
int unopose_geo_embedding_train_backward(const float *points, const int32_t *knn, int B, int n, int rows_d, int rows_a, int hinv, float sigma_d,
                                         float factor_a, int reduce_mean, const float *dE, const void *amax, float *ws, float *full_d,
                                         int *past_table, unopose_stream_t stream);

/* One scale of the fine matcher's PositionalEncoding: QueryAndLRFGroup(radius, nsample,
 * use_xyz) -> SharedMLP[6,32,64,128] (1x1 conv + eval BatchNorm + ReLU) -> max over neighbours
 * (core/unopose/model/oneref_predator_fine_point_matching.py:167-174).  xyz (B,N,3) ->
 * out (B,N,128) float32.  w1 (32,6), w2 (64,32), w3 (128,64) row-major [out][in] with
 * BatchNorm already folded in, b1/b2/b3 the folded biases.  nsample % 32 == 0.
 * Exact fp32 matrix cores (v_mfma_f32_32x32x2_f32); bf16x3 must be 0 (the split-precision form is
 * unopose_pe_group_mlp_max_packed below). */
int unopose_pe_group_mlp_max(const float *xyz, int B, int N, float radius, int nsample,
                             const float *w1, const float *b1, const float *w2,
                             const float *b2, const float *w3, const float *b3,
                             int bf16x3, float *out, unopose_stream_t stream);

/* ---- pose heads (core/unopose/utils/model_utils.py:411-490 coarse, :527-566 fine) ----
 * atten (B,R,C) float32 similarity with background row/col 0; score1 (B,R-1), score2 (B,C-1).
 * a_ij = softmax_row * softmax_col * s1_i * s2_j (s*_0 = 1).
 *
 * assign_labels: streaming softmax statistics into stats_ws (2*B*(R+C) floats:
 * [rmax B*R | 1/rsum B*R | cmax B*C | 1/csum B*C], the sums of exp(x - max) stored as reciprocals) and the foreground masks
 * w1 (B,R-1) = [argmax_j a_ij > 0], w2 (B,C-1) = [argmax_i a_ij > 0]  (:444-447, :542-545). */
/* Training (core/unopose/utils/loss_utils.py:181-187, the two-way InfoNCE "atten" loss of compute_overlap_loss):
 * softmax_stats writes the statistics alone (same stats_ws layout); infonce_grad writes, for labels label1 (B,R-1) / label2 (B,C-1)
 * (int64; 0 = background column / row) and the upstream gradient g (B) of
 *   L_b = 0.5 (mean_{i>=1} CE(row i over all columns, label1) + mean_{j>=1} CE(column j over all rows, label2)),
 * dL/dx into grad (B,R,C) in one pass over x. */
int unopose_softmax_stats(const float *x, int B, int R, int C, float *stats_ws, unopose_stream_t stream);
int unopose_infonce_grad(const float *x, int B, int R, int C, const float *stats_ws, const long long *label1,
                         const long long *label2, const float *g, float *grad, unopose_stream_t stream);

/* Training-mode BatchNorm2d + ReLU of the PE's SharedMLP (pytorch_utils.py:25-132 under train(): nn.BatchNorm2d with batch
 * statistics followed by ReLU), x / y / dy / dx (B, C, L) fp32 with L = N * S a multiple of 4.  forward: batch mean / rstd (biased
 * variance) into `mean`, `rstd`, running statistics updated as nn.BatchNorm2d does (unbiased variance; both NULL: not tracked),
 * y = relu(bn(x)).  backward: dgamma, dbeta, dx from x, dy and the forward's mean / rstd (the ReLU mask is recomputed).
 * workspace: 2 * B * C * ceil(L / unopose_bn_train_chunk()) floats. */
int unopose_bn_train_chunk(void);
int unopose_bn_relu_train_forward(const float *x, int B, int C, long L, const float *gamma, const float *beta, float eps, float momentum,
                                  float *running_mean, float *running_var, float *workspace, float *mean, float *rstd, float *y,
                                  unopose_stream_t stream);
int unopose_bn_relu_train_backward(const float *x, const float *dy, int B, int C, long L, const float *gamma, const float *beta,
                                   const float *mean, const float *rstd, float *workspace, float *dgamma, float *dbeta, float *dx,
                                   unopose_stream_t stream);

/* The LAST SharedMLP layer fused with the max over the S neighbours that follows it (oneref_predator_fine_point_matching.py:167-174
 * in train()): out[b, c, n] = max_s relu(batch_norm(x))[b, c, n, s], idx = the first arg max; x (B, C, N, S) float32, S in {32, 64, 128, 256}.
 * The normalised activation is never written.  Backward takes the POOLED gradient g (B, C, N): dgamma / dbeta from the B N winners,
 * dx dense.  Workspaces: forward as unopose_bn_relu_train_forward (L = N S), backward 2 * B * C floats. */
int unopose_bn_relu_maxpool_train_forward(const float *x, int B, int C, int N, int S, const float *gamma, const float *beta, float eps,
                                          float momentum, float *running_mean, float *running_var, float *workspace, float *mean, float *rstd,
                                          float *out, int32_t *idx, unopose_stream_t stream);
int unopose_bn_relu_maxpool_train_backward(const float *x, const float *g, const int32_t *idx, int B, int C, int N, int S, const float *gamma,
                                           const float *beta, const float *mean, const float *rstd, float *workspace, float *dgamma, float *dbeta,
                                           float *dx, unopose_stream_t stream);

/* Saliency of the matchers' training branch (oneref_predator_coarse_point_matching.py:68-76 / _fine_point_matching.py:91-99 under
 * train()): with inner = atten[:, 1:, 1:] of the (B, n1 + 1, n2 + 1) float32 similarity,
 *   m1 = softmax(inner, dim=2) @ s2   (B, n1),     m2 = softmax(inner^T, dim=2) @ s1   (B, n2),
 * forward (row / column softmax statistics rmax, rsum (B, n1) and cmax, csum (B, n2) are returned for backward) and backward
 * (d_atten (B, n1 + 1, n2 + 1) with zero first row / column, ds1 (B, n1), ds2 (B, n2) from the gradients g1, g2 of m1, m2).
 * Nothing of the matrix's size is written forward; deterministic reductions. */
int unopose_saliency_train_forward(const float *atten, const float *s1, const float *s2, int B, int n1, int n2, float *m1, float *m2, float *rmax,
                                   float *rsum, float *cmax, float *csum, unopose_stream_t stream);
int unopose_saliency_train_backward(const float *atten, const float *s1, const float *s2, const float *m1, const float *m2, const float *rmax,
                                    const float *rsum, const float *cmax, const float *csum, const float *g1, const float *g2, int B, int n1, int n2,
                                    float *d_atten, float *ds1, float *ds2, unopose_stream_t stream);

/* Training labels (core/unopose/utils/loss_utils.py:150-176): for clouds a (B, n, 3) and b (B, m, 3), float32, the nearest partner of
 * every point of one cloud in the other and whether ANY partner lies within thr, without forming the (B, n, m) distance matrix.
 * over_b != 0: for every a_i over j (outputs of length n); else for every b_j over i (length m).  Entry values follow the reference's
 * expansion sqrt(max(|a_i|^2 - 2 a_i.b_j + |b_j|^2, 0)) in both directions; ties: the first index.  any_close is one byte per point. */
int unopose_nearest_partner(const float *a, const float *b, int B, int n, int m, int over_b, float thr, float *dmin, int32_t *arg,
                            uint8_t *any_close, unopose_stream_t stream);

/* nn.Conv2d(cin, cout, 1, bias=False) of the same SharedMLP under autograd (pytorch_utils.py:25-132 in train()), on (B, C, L) float32
 * slabs with L = N * S a multiple of 64 (v_mfma_f32_32x32x2_f32: fp32 products and accumulation).
 *   forward:  y[b, m, l] = sum_k w[m, k] x[b, k, l], w (cout, cin) row-major.  The input gradient is the same call on the transposed
 *             weights (x := dy, w := w^T).  Built channel pairs: cin <= 8 -> 32; <= 32 -> 32, 64; <= 64 -> 32, 64, 128; <= 128 -> 64, 128.
 *   wgrad:    dw[m, k] = sum_{b, l} dy[b, m, l] x[b, k, l]; workspace = unopose_conv1x1_train_wgrad_blocks() * cout * 128 floats
 *             (per-workgroup partial sums, combined in double: deterministic).  Built: cout in {32, 64, 128} with cin <= 32 / 64 / 128. */
int unopose_conv1x1_train_wgrad_blocks(void);
int unopose_conv1x1_train_forward(const float *x, int B, int cin, long L, const float *w, int cout, float *y, unopose_stream_t stream);
int unopose_conv1x1_train_wgrad(const float *dy, const float *x, int B, int cout, int cin, long L, float *workspace, float *dw,
                                unopose_stream_t stream);

/* Weight gradient of a trainable nn.Linear of the matcher (the `grad_output^T @ input` of torch.nn.functional.linear's backward):
 * dw[n, k] = sum_r g[r, n] x[r, k], g (rows, N) and x (rows, K) float32 row-major, N and K multiples of 128; fp32 products and
 * accumulation on v_mfma_f32_32x32x2_f32 over slices of the rows, slices combined in double (deterministic).
 * workspace = unopose_linear_wgrad_f32_splits(rows, N, K) * N * K floats. */
int unopose_linear_wgrad_f32_splits(long rows, int N, int K);
int unopose_linear_wgrad_f32(const float *g, const float *x, long rows, int N, int K, float *workspace, float *dw, unopose_stream_t stream);
int unopose_assign_labels(const float *atten, int B, int R, int C, const float *score1,
                          const float *score2, float *stats_ws, float *w1, float *w2,
                          unopose_stream_t stream);

/* fine stage (:547-553): weight_i = sum_j a_ij w1_i w2_j, pred_i = sum_j (..) pts2_j / (weight_i + 1e-6). */
int unopose_fine_correspondences(const float *atten, int B, int R, int C, const float *score1,
                                 const float *score2, const float *stats_ws, const float *w1,
                                 const float *w2, const float *pts2, float *weight, float *pred,
                                 unopose_stream_t stream);

/* ---- pixel features without the dense up-projection (ViT_AE, oneref_feature_extraction.py:200-236 + get_chosen_pixel_feats,
 * model_utils.py:215-227): only the <= 4 cells of the (4 side)^2 x 256 map a chosen pixel's bilinear tap reads are computed.
 * upproj_plan: choose (B2,Np) int64 pixel indices of the (H,W) crops -> row_list (cap_rows) activation row of every needed
 * cell, grouped by sub-position s = 4 (Y & 3) + (X & 3) (groups padded to 256-row tiles with -1), cellmap (B2, 16 side^2)
 * compact row of a cell, tile_info[18] = {tiles, first tile of group 0..16}.  Activation row = crop * tok_stride + tok_offset
 * + token.  ws: B2 * (16 side^2 + 32) ints; cap_rows: a multiple of 256, >= B2 * min(4 Np, 16 side^2) + 4096. */
int unopose_upproj_plan(const long long *choose, int B2, int Np, int H, int W, int side, int tok_offset, int tok_stride,
                        int cap_rows, int *ws, int *row_list, int *cellmap, int *tile_info, unopose_stream_t stream);

/* linear_bf16 with explicit row strides (in elements, multiples of 8): A rows lda apart, W rows ldw apart, C rows ldc apart -- an
 * nn.Linear (oneref_feature_extraction.py:24-42, transformer.py:151-193) over a column slice of a wider activation, or into one,
 * without a copy. */
int unopose_linear_bf16_ld(const void *A, int lda, const void *W, int ldw, const float *bias, void *C, int ldc, long M, int N, int K,
                           int epilogue, unopose_stream_t stream);

/* C = LayerNorm(A W^T + bias + resid) * ln_w + ln_b for 256-wide layers (N = 256 = one tile: a row's statistics stay inside the
 * workgroup): the output projection / FFN squeeze of the matcher's transformer layers with the residual add and the post-LN
 * (core/unopose/model/transformer.py:151-193) in the GEMM epilogue, on the fp32 accumulators.  A (M,K), W (256,K), resid and C
 * (M,256) bfloat16; bias, ln_w, ln_b float32 (256). */
int unopose_linear_add_layernorm_bf16(const void *A, const void *W, const float *bias, const void *resid, const float *ln_w,
                                      const float *ln_b, float eps, void *C, long M, int K, unopose_stream_t stream);

/* linear_bf16 (bias only) for the token attention's fused projections (q | q W_p | k | v or k | v; model/transformer.py:130-148, 386-405): the
 * LAST 256 output columns -- V -- are not written to C but TRANSPOSED and key-padded into vt (M / tokens clouds, 256, key_pad) bf16, keys
 * tokens .. key_pad - 1 zero: the V^T operand of unopose_token_attention, without a transpose launch.  M = whole clouds of `tokens` rows;
 * N % 128 == 0, K % 64 == 0; C (M, N) keeps its other columns. */
int unopose_linear_bf16_kv_vt(const void *A, const void *W, const float *bias, void *C, void *vt, long M, int N, int K, int tokens, int key_pad,
                              unopose_stream_t stream);

/* Row-gathered grouped form of linear_bf16: C[r] = A[row_list[r]] . W[g(r)*256 .. +255]^T + bias, g(r) = the group of r's tile
 * (tile_info as written by upproj_plan; N / 256 groups).  A (M,K), W (N,K), C (max_tiles * 256, 256) bf16; bias fp32 (N).
 * The tile count is read on the device. */
int unopose_linear_bf16_gather(const void *A, long M, int K, const void *W, int N, const float *bias, const int *row_list,
                               const int *tile_info, int max_tiles, void *C, unopose_stream_t stream);

/* out (B2,Np,256) = bilinear blend (fp32 arithmetic) of the 4 compact rows of each chosen pixel (torch upsample_bilinear2d,
 * align_corners=False, then the pixel gather); float32, or with out_bf16 rounded to bf16 (under autocast the reference's features are
 * half precision from the up-projection on, and the next consumer is an autocast Linear). */
int unopose_bilinear_sample_compact(const void *Cc, const int *cellmap, const long long *choose, int B2, int side, int Np,
                                    int H, int W, void *out, int out_bf16, unopose_stream_t stream);

/* fine stage without the similarity matrix (bf16 / autocast path; replaces compute_feature_similarity :260-282 followed by
 * assign_labels + fine_correspondences): f1 (B,R,D) and f2 (B,C,D) are the L2-normalised out_proj features as bf16, f1
 * already multiplied by 1/temp; x_ij = f1_i . f2_j is recomputed tile by tile on the matrix cores in each of the three
 * reduction passes instead of being stored (16.8 MB per pair) and re-read.  shift = 1/temp (any bound of |x|: the softmax is
 * evaluated as exp(x - shift) / sum, no running maxima).  D must be 256.  ws: B*(R+C) + B*(ceil((R-1)/256) + ceil((C-1)/256))
 * floats of scratch.  Outputs as assign_labels / fine_correspondences: w1 (B,R-1), w2 (B,C-1), weight (B,R-1), pred (B,R-1,3). */
int unopose_fine_assign(const void *f1, const void *f2, int B, int R, int C, int D, float shift, const float *score1,
                        const float *score2, const float *pts2, float *ws, float *w1, float *w2, float *weight,
                        float *pred, unopose_stream_t stream);

/* out[bc,i] = min_j |p'_i - q_j| with p' = (p_i - t_bc) R_bc when R/t are given (row-vector
 * convention of the reference, :481, :559), bc = b*cand_per_b + c.  p (B,N,3), q (B,M,3). */
int unopose_min_dist(const float *p, const float *q, int B, int N, int M, const float *R,
                     const float *t, int cand_per_b, float *out, unopose_stream_t stream);

/* coarse stage (:449-474): CDF of (a_ij w1_i w2_j)^1.5 into cdf_ws (B,(R-1)*(C-1)), then for
 * each of nprop hypotheses three searchsorted look-ups with rand (B,3*nprop), a 3-point
 * Procrustes (register Jacobi SVD) and the mean residual: Rout (B,nprop,9), tout (B,nprop,3),
 * dis (B,nprop). */
int unopose_coarse_hypotheses(const float *atten, int B, int R, int C, const float *score1,
                              const float *score2, const float *stats_ws, const float *w1,
                              const float *w2, const float *rand, int nprop, const float *pts1,
                              const float *pts2, float *cdf_ws, float *Rout, float *tout,
                              float *dis, unopose_stream_t stream);

/* coarse pose selection (:480-485): score[b,c] = sum(w1) / (sum_i w1_i min_j |(p1_i - t) R - p2_j| + 1e-8)
 * for the hypotheses top[b,c] (int64 indices into the nprop hypotheses). */
int unopose_coarse_scores(const float *pts1, const float *pts2, int B, int n1, int n2,
                          const float *Rall, const float *tall, int nprop, const int64_t *top,
                          int ncand, const float *w1, float *score, unopose_stream_t stream);

/* idx (B,k) int64 = the k smallest of x (B,n) float32 per row in ascending order, ties by index, NaN as the largest value:
 * torch.topk(dis, n_proposal2, dim=1, largest=False)[1] of compute_coarse_Rt_overlap (model_utils.py:476).  n <= 16384. */
int unopose_topk_smallest(const float *x, int B, int n, int k, int64_t *idx, unopose_stream_t stream);
/* The winning hypothesis of every pair (model_utils.py:486-490): best = first maximum of score (B,ncand) (a NaN wins, as in torch.max),
 * R (B,3,3) / t (B,3) = Rall / tall (B,nprop,...) at hypothesis top[b,best], best_score (B) = score[b,best]. */
int unopose_coarse_pick(const float *score, const int64_t *top, int B, int ncand, const float *Rall, const float *tall, int nprop,
                        float *R, float *t, float *best_score, unopose_stream_t stream);

/* Attention core of the 4-head x 64 token transformers (core/unopose/model/transformer.py:130-148
 * cross, :386-405 RPE self): out (B,n,256) = softmax((q k^T [+ qp . E]) * scale) v, all tensors
 * bfloat16 bit patterns.  q (B,n,256) with row stride ldq elements, k (B,m,256) with row stride ldk
 * (so both can be read in place from a fused projection output), channel = head*64 + c; vt (B,256,KP)
 * = v transposed to channel-major and zero-padded to KP = unopose_token_attention_key_pad() keys;
 * RPE only: qp (B,n,4,256) with row stride ldqp = q_h W_p,h (the folded proj_p), E (B,n,m,256)
 * contiguous, the geometric embedding; pass qp = E = NULL for plain (cross) attention.  m <= KP. */
int unopose_token_attention(const void *q, int ldq, const void *k, int ldk, const void *vt,
                            const void *qp, int ldqp, const void *E, int B, int n, int m,
                            float scale, void *out, unopose_stream_t stream);
int unopose_token_attention_key_pad(void);

/* float32 twins of unopose_token_attention / unopose_vit_attention (same layouts, float32 data; vt zero-
 * padded to unopose_token_attention_key_pad() keys): operands are split on the fly into hi + lo bfloat16
 * and every product issued as 3 bf16 MFMAs (fp32 accumulation, ~2^-16 relative error), so the fp32
 * configuration of the reference (configs/main_cfg.py:87-89) also runs on hand-written kernels. */
int unopose_token_attention_f32(const float *q, int ldq, const float *k, int ldk, const float *vt,
                                const float *qp, int ldqp, const float *E, int B, int n, int m,
                                float scale, float *out, unopose_stream_t stream);
int unopose_vit_attention_f32(const float *qkv, int B, int T, int H, float *out, unopose_stream_t stream);
/* The same with the output in the split layout of unopose_linear_f32x3 ((B,T,2 H 64) bf16: the operand of the projection that
 * follows, timm Attention.proj). */
int unopose_vit_attention_f32_split(const float *qkv, int B, int T, int H, void *out_split, unopose_stream_t stream);
/* Split layout IN and OUT (csrc/vit_attn_f32s.hip): qkv_split = the (B*T, 3 H 64) qkv in the split layout, as unopose_linear_f32x3
 * writes it (Cs); the kernel of the fp32 ViT blocks -- 8 waves x 32 queries, 128-key double-buffered chunks, no operand is split
 * inside the kernel. */
int unopose_vit_attention_f32_ss(const void *qkv_split, int B, int T, int H, void *out_split, unopose_stream_t stream);

/* ViT attention core (timm Attention as driven by core/unopose/model/oneref_feature_extraction.py:38-41):
 * out (B,T,H*64) = softmax(q k^T / 8) v per head, flash-style.  qkv (B,T,3,H,64) = the fused qkv Linear
 * output; bfloat16 bit patterns. */
int unopose_vit_attention(const void *qkv, int B, int T, int H, void *out, unopose_stream_t stream);

/* out = LayerNorm(a (+ b)) * w + bias over the last dimension C (<= 1024), rows x C row-major.
 * a / b / out are float32 or bfloat16 (flags); b may be NULL.  One pass instead of the reference's
 * add + layer_norm (+ autocast casts): transformer.py:151-193, timm Block norm1/norm2. */
int unopose_add_layernorm(const void *a, int a_bf16, const void *b, int b_bf16, const float *w,
                          const float *bias, long rows, int C, float eps, void *out, int out_bf16,
                          unopose_stream_t stream);

/* Same with a row stride ld_out >= C (elements) on the OUTPUT: lets several LayerNorms write side by side into
 * one wider row-major buffer (the four ViT taps that oneref_feature_extraction.py:213 concatenates). */
int unopose_add_layernorm_strided(const void *a, int a_bf16, const void *b, int b_bf16, const float *w,
                                  const float *bias, long rows, int C, float eps, void *out, int out_bf16,
                                  long ld_out, unopose_stream_t stream);

/* Pixel features at the chosen pixels only: bilinear resize (align_corners=False) of the 4x up-projected
 * ViT map to (H,W) fused with get_chosen_pixel_feats (oneref_feature_extraction.py:221-229,
 * utils/model_utils.py:215-227).  z (B,side,side,4,4,256) float32 or bfloat16 = the up-projection output
 * in its native order, choose (B,Np) int64 flat pixel indices, out (B,Np,256) float32. */
int unopose_bilinear_sample(const void *z, int z_bf16, const long long *choose, int B, int side, int Np,
                            int H, int W, float *out, unopose_stream_t stream);

/* Same on a token tensor that still carries prefix tokens: z (B, tok_stride, 4, 4, 256) with patch token p of
 * image b at row tok_offset + p (the ViT's 5 class / register tokens are skipped by the index math instead
 * of being sliced off with a copy). */
int unopose_bilinear_sample_tokens(const void *z, int z_bf16, const long long *choose, int B, int side, int Np,
                                   int H, int W, int tok_offset, int tok_stride, float *out,
                                   unopose_stream_t stream);

/* x (rows,C) float32 += gamma * y (bfloat16) in place AND out (bfloat16) = LayerNorm(x) * w + bias: the
 * LayerScale residual of one timm ViT branch fused with the LayerNorm that opens the next one. */
int unopose_scale_residual_layernorm(float *x, const void *y_bf16, const float *gamma, const float *w,
                                     const float *bias, long rows, int C, float eps, void *out_bf16,
                                     unopose_stream_t stream);
/* The same for fp32 data (no autocast: the reference's default precision): x += gamma * y with y float32 (y NULL: no update);
 * LayerNorm(x) is written in the split layout of unopose_linear_f32x3 (out_split NULL: residual update only).  C % 32 == 0.
 * out_ld_bytes: distance between output rows (0 = 4 C, dense): a column block of a wider split-layout matrix -- the four tap LayerNorms of
 * ViT_AE (oneref_feature_extraction.py:207-213) side by side as the up-projection's K = 4 C operand, no concatenation. */
int unopose_scale_residual_layernorm_f32(float *x, const float *y, const float *gamma, const float *w, const float *bias, long rows,
                                         int C, float eps, void *out_split, long out_ld_bytes, unopose_stream_t stream);

/* x (rows,C) float32 += gamma (C) * y (rows,C) bfloat16, in place: the LayerScale residual of a
 * timm ViT block (x = x + ls(branch(x))). */
int unopose_scale_residual(float *x, const void *y_bf16, const float *gamma, long rows, int C,
                           unopose_stream_t stream);

/* Focused linear attention core (core/unopose/model/transformer.py:533-568), 4 heads x 64.
 * x (B,N,256) bf16 = proj_q(dense) [mode 0] or proj_k(sparse) [mode 1]; inv_softplus_scale (256) fp32.
 * mode 1: out (B,N,256) bf16 = the focused features relu/scale/cube/renorm only.
 * mode 0: out = (q_h kv_h) / (q_h . ksum_h + 1e-6) per head with kvt (B,4,64 d,64 c) bf16 = kv_h^T and
 *         ksum (B,256) fp32 = sum_j focused k_j. */
int unopose_linear_attention(const void *x, const float *inv_softplus_scale, const void *kvt,
                             const float *ksum, int B, int N, int focus, int mode, void *out,
                             unopose_stream_t stream);
/* The key / value state of one layer in one launch (transformer.py:545-562: k's focusing, k.sum(dim=1) and the "bjhc,bjhd->bhcd"
 * einsum): ykv (B,rows_per_pair,512) bf16 = [k projection | v] rows as the fused k | v projection writes them, of which rows
 * [first_row, first_row + J) of every pair are the tokens (the sparse-to-dense block's tokens sit behind their background-token row,
 * transformer.py:655-668: first_row = 1); kvt (B,4,64 d,64 c) bf16 and ksum (B,256) float32 come out in the layouts
 * unopose_linear_attention(mode 0) reads.  The focused keys are rounded to bf16 before both sums (the values mode 1 writes). */
int unopose_linear_attention_kv_state(const void *ykv, const float *inv_softplus_scale, int B, int J, int rows_per_pair,
                                      int first_row, int focus, void *kvt, float *ksum, unopose_stream_t stream);
/* Same function on float32 data (the reference's default precision, configs/main_cfg.py:87-89): x, kvt
 * and out are float32; the per-head contraction runs as hi/lo-split bf16 MFMAs (fp32-class accuracy). */
int unopose_linear_attention_f32(const float *x, const float *inv_softplus_scale, const float *kvt,
                                 const float *ksum, int B, int N, int focus, int mode, float *out,
                                 unopose_stream_t stream);

/* Depth maps of a triangle mesh in P poses (BOP's VSD error renders the object in the estimated and the ground-truth pose:
 * third_party/bop_toolkit/bop_toolkit_lib/pose_error.py:17-101, renderer.render_object(...)["depth"]).  verts (V,3) float32, faces (F,3)
 * int32, Rt (P,12) = row-major R then t (model -> camera, camera looks along +z), K4 (P,4) = fx, fy, cx, cy; integer pixel coordinates
 * are pixel centres (misc.py:142-162); depth (P,H,W) float32 = z of the nearest surface, 0 where nothing projects. */
int unopose_render_depth(const float *verts, int V, const int *faces, int F, const float *Rt, const float *K4, int P, int H, int W,
                         float *depth, unopose_stream_t stream);

/* ---- data-movement glue of the forward as single-pass kernels (csrc/glue.hip) ---------------------------------------------
 * patchify: the ViT's 14 x 14 / 14 patch unfolding (timm PatchEmbed, oneref_feature_extraction.py:24-27) of two image batches
 * (na + nb crops of (3,S,S) float32; rgb_b may be NULL with nb = 0) into the bf16 matrix ((na + nb) (S/14)^2, Kp) the
 * patch-embedding GEMM reads: column = c * 196 + py * 14 + px, columns 588 .. Kp-1 zero. */
int unopose_patchify_bf16(const float *rgb_a, int na, const float *rgb_b, int nb, int S, int Kp, void *out,
                          unopose_stream_t stream);
/* tokens of timm's _pos_embed (no_embed_class) + the first block's LayerNorm in one pass: x (nimg, npre + P, 768) float32 =
 * [prefix (npre,768) | patch (nimg,P,768) bf16 + pos (P,768)], n1 = LayerNorm(x) * ln_w + ln_b as bf16. */
int unopose_vit_tokens_layernorm(const void *patch, const float *pos, const float *prefix, int npre, int P, int nimg, int C,
                                 const float *ln_w, const float *ln_b, float eps, float *x, void *n1, unopose_stream_t stream);
/* The two steps above for the no-autocast forward (the reference's default precision): the patch matrix written in the split layout of
 * unopose_linear_f32x3 (Kp % 32 == 0: 588 -> 608 zero-padded columns), and the token assembly on the float32 patch embedding with the
 * first LayerNorm in the split layout. */
int unopose_patchify_split(const float *rgb_a, int na, const float *rgb_b, int nb, int S, int Kp, void *out, unopose_stream_t stream);
int unopose_vit_tokens_layernorm_f32(const float *patch, const float *pos, const float *prefix, int npre, int P, int nimg, int C,
                                     const float *ln_w, const float *ln_b, float eps, float *x, void *n1_split, unopose_stream_t stream);
/* out[r] = x[r,:] . w + b for 256-wide rows: the overlap-score heads nn.Linear(256, 1)
 * (oneref_predator_coarse_point_matching.py:66, ..._fine_point_matching.py:89). */
int unopose_row_dot(const void *x, int x_bf16, const float *w, float b, long rows, int C, void *out, int out_bf16,
                    unopose_stream_t stream);
/* out[r,:] = bf16( x[r,:] / max(||x[r,:]||_2, 1e-12) / temp ) for 256-wide rows: the operands of compute_feature_similarity
 * (core/unopose/utils/model_utils.py:260-282) as the fine assignment reads them; out_f32 = 1: the same bf16-rounded values stored as
 * float32 (the operand type of unopose_bmm_f32, which forms the coarse similarity from them); out_f32 = 2: the unrounded float32 values
 * (F.normalize at the reference's default precision). */
int unopose_normalize_rows_bf16(const void *x, int x_bf16, long rows, int C, float temp, void *out, int out_f32,
                                unopose_stream_t stream);
/* vt (B, C, pad) bf16: vt[b,c,j] = v[b,j,c] (rows of v `ld` elements apart), zero for m <= j < pad: the value image of
 * unopose_token_attention. */
/* out (B, prepend + J, row) = rows of feats (B, N, row) picked by idx (B, J; int32 or int64): row idx - off, or alt (B rows,
 * alt_stride_bytes apart, 0 = one row for all: the background tokens may be row 0 of a (B, 1 + n, row) tensor) where idx - off < 0; with prepend = 1 row 0
 * of every batch is alt as well.  Rows are row_bytes long (a multiple of 4, any element
 * type).  The (B,N,C)-layout gather the model uses instead of gather_operation's transposes (model_utils.py:146-149) and the
 * background-token sampling of the sparse-to-dense block (transformer.py:655-662: index 0 = the background token). */
int unopose_gather_rows(const void *feats, int B, int N, int row_bytes, const void *idx, int idx_is_i64, int J, int off,
                        const void *alt, long alt_stride_bytes, int prepend, void *out, unopose_stream_t stream);
int unopose_transpose_pad_bf16(const void *v, long ld, int B, int m, int C, int pad, void *vt, unopose_stream_t stream);
/* The same on float32 data: the value image of unopose_token_attention_f32 (transformer.py:386-405 at the reference's default precision). */
int unopose_transpose_pad_f32(const float *v, long ld, int B, int m, int C, int pad, float *vt, unopose_stream_t stream);

/* Small fp32 glue of the forward (round 5: the last torch reductions / elementwise kernels of the eval path).
 * cloud_radius: radius[b] = max_i |p_i - mean(p)| of pts (B,N,3) (oneref_grf_predator_pose_estimation_model.py: the normalisation radius).
 * scale_by_radius: out (B,n) = x / (radius[b] + eps) (multiply = 0) or x * (radius[b] + eps) (1).
 * overlap_scores: clamp(sigmoid(.), 0, 1) of the score-head outputs (B, n_tot) of the two stacked clouds without their background tokens
 *   (positions 0 and n1 + 1) -> out (B, n_tot - 2) fp32 (oneref_predator_coarse_point_matching.py:68-76, fine: :91-99); with `halves` the
 *   head ran over the clouds as one batch of 2B: scores is (2B, n1 + 1), cloud 2 of pair b in batch B + b (n_tot = 2 (n1 + 1)).
 * copy_rows: dst row r = src row r for `rows` rows of row_bytes, the rows src_stride_bytes / dst_stride_bytes apart (src stride 0: one
 *   row for all) -- the background token written in front of every pair (`torch.cat([bg_token.expand(B, -1, -1), f], dim=1)`,
 *   oneref_predator_coarse_point_matching.py:52-54) into a tensor that already has the slot.
 * rigid_rows_bf16: bf16((p - t) @ R) with bf16-rounded operands and fp32 accumulation, p (B,N,3) fp32 (autocast's bmm of
 *   oneref_predator_fine_point_matching.py:69).
 * token_sum_bf16: out (B,C) fp32 = sum over the J tokens of x (B,J,C) bf16 (the k-sum of the focused linear attention, transformer.py:560-566).
 * pose_score: sum_i [dis_i < thr] w_i / (sum_i w_i + 1e-8) * mean_i w_i, dis / w (B,N) (model_utils.py:559-566). */
int unopose_cloud_radius(const float *pts, int B, int N, float *radius, unopose_stream_t stream);
int unopose_scale_by_radius(const float *x, int B, int n, const float *radius, float eps, int multiply, float *out, unopose_stream_t stream);
int unopose_overlap_scores(const void *scores, int x_bf16, int B, int n_tot, int n1, int halves, float *out, unopose_stream_t stream);
/* The gradient hygiene of the training loop (core/unopose/engine/engine_utils.py:14-18: torch.nan_to_num(p.grad, nan=0, posinf=1e5, neginf=-1e5) for every
 * parameter) over ALL gradients in one launch: ptrs_dev / sizes_dev are DEVICE arrays of n_tensors float32 pointers / int64 element counts, max_size the
 * largest count.  In place; finite values are not rewritten. */
int unopose_nan_to_num_multi(const void *ptrs_dev, const void *sizes_dev, int n_tensors, long max_size, float nan_value, float posinf_value,
                             float neginf_value, unopose_stream_t stream);
int unopose_copy_rows(const void *src, long src_stride_bytes, void *dst, long dst_stride_bytes, int rows, int row_bytes,
                      unopose_stream_t stream);
int unopose_rigid_rows_bf16(const float *p, int B, int N, const float *t, const float *R, void *out, unopose_stream_t stream);
int unopose_token_sum_bf16(const void *x, int B, int J, int C, float *out, unopose_stream_t stream);
int unopose_pose_score(const float *dis, const float *w, int B, int N, float thr, float *out, unopose_stream_t stream);

/* nn.Linear on bf16 data with a fused epilogue (timm ViT blocks: qkv / proj / fc1 + GELU / fc2, and the
 * up-projection of oneref_feature_extraction.py:221):
 *     C (M,N) bf16 = act( A (M,K) bf16 . W (N,K)^T bf16 + bias (N) fp32 ),  fp32 accumulation,
 * epilogue 0 = bias only, 1 = bias + GELU (erf form to 2.5e-5 absolute, far inside the bf16 result's resolution) evaluated
 * on the fp32 accumulators, 2 = bias + ReLU.
 * Requires N % 256 == 0 and K % 64 == 0 (unopose_gemm_bf16_tile() reports the 256); any M >= 1. */
int unopose_linear_bf16(const void *A, const void *W, const float *bias, void *C, long M, int N, int K,
                        int epilogue, unopose_stream_t stream);
int unopose_gemm_bf16_tile(void);

/* The residual + LayerNorm passes of a timm Block (x + ls(f(norm(x))), oneref_feature_extraction.py:24-42) folded into the GEMMs
 * around them (bf16 autocast forward).
 * _residual (proj / fc2, LayerScale folded into W and bias by the caller):  xres (M,N) fp32 += A (M,K) . W (N,K)^T + bias, in place;
 *     xb (M,N) bf16 = the updated rows; stats[(row * N/256 + t) * 2 + {0,1}] = (sum, sum of squares) of the row's columns 256 t .. 256 t + 255
 *     (stats holds ceil(M/256)*256 rows: whole tiles are written).
 * _lnfold (qkv / fc1 reading those un-normalised rows against W' = ln_weight (.) W):
 *     C (M,N) bf16 = act( rstd_r (A W'^T - mean_r cvec) + dvec ),  cvec[n] = sum_k W'[n][k],  dvec[n] = sum_k ln_bias[k] W[n][k] + b[n],
 *     mean_r / rstd_r from the `nparts` partial sums of row r (LayerNorm over the K columns, eps), gelu = 1: erf-class GELU.
 * Both need N % 256 == 0 and K % 64 == 0. */
int unopose_linear_bf16_residual(const void *A, const void *W, const float *bias, float *xres, void *xb, float *stats, long M, int N, int K,
                                 unopose_stream_t stream);
int unopose_linear_bf16_lnfold(const void *A, const void *W, const float *dvec, const float *cvec, const float *stats, int nparts, float eps,
                               void *C, long M, int N, int K, int gelu, unopose_stream_t stream);
/* Start offset of every second workgroup of `_residual`, in 1/8 ticks of the 100 MHz clock per 64-wide K step (0: all start together;
 * < 0 only queries); returns the previous value.  A tuning knob of the tile schedule: results do not depend on it. */
int unopose_gemm_fold_stagger(int eighth_ticks_per_ktile);

/* Small batched float32 contraction on the exact-fp32 matrix instruction (an fma chain, one rounding per product):
 *     C[(bo,bi)][i][j] = alpha * sum_k A[bo sab + bi sah + i sai + k sak] * Bm[bo sbb + bi sbh + j sbj + k sbk],  C (bo*bi, n, m) row-major.
 * The coarse feature similarity (model_utils.py:260-282), the focused linear attention's k^T v (transformer.py:560-566). */
int unopose_bmm_f32(const float *A, long sab, long sah, long sai, long sak, const float *Bm, long sbb, long sbh, long sbj, long sbk,
                    float *C, int bo, int bi, int n, int m, int K, float alpha, unopose_stream_t stream);

/* nn.Linear on float32 data (the reference's default precision, configs/main_cfg.py:87-89) with fp32-class accuracy on the
 * bf16 matrix cores: every operand is split into hi + lo bf16 parts and a product is three MFMAs (ah wh + ah wl + al wh,
 * fp32 accumulation; relative error ~2^-17 per product).  Operands come in the SPLIT layout -- row r, k-block j (32 k) is
 * one 128-byte line [hi (32 x bf16) | lo (32 x bf16)], rows K * 4 bytes apart, i.e. the size of the fp32 matrix --
 * written by unopose_split_bf16x2 (X (M,K) float32, K % 32 == 0) or by the GEMM itself:
 *     C (M,N) float32 and / or Cs (M,N) split = act( As (M,K) . Ws (N,K)^T + bias (N) float32 )
 * epilogue 0 = bias, 1 = bias + exact (erf) GELU, 2 = bias + ReLU; either of C / Cs may be NULL.
 * Requires N % 256 == 0, K % 32 == 0.
 * unopose_linear_f32x3_bf16: the same product rounded to bfloat16, optionally added to a bfloat16 residual (result rounded again):
 * Cb = bf16( resid + bf16(As Ws^T + bias) ) -- the fp32 island `d + PE(p).to(bf16)` of the fine matcher under autocast
 * (oneref_predator_fine_point_matching.py:163-165, 77-80). */
int unopose_split_bf16x2(const float *X, long M, int K, void *Xs, unopose_stream_t stream);
int unopose_linear_f32x3(const void *As, const void *Ws, const float *bias, float *C, void *Cs, long M, int N, int K,
                         int epilogue, unopose_stream_t stream);
int unopose_linear_f32x3_bf16(const void *As, const void *Ws, const float *bias, const void *resid, void *Cb, long M, int N, int K,
                              unopose_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* UNOPOSE_HIP_H */
