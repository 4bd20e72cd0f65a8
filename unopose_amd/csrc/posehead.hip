// Pose heads of UNOPose for gfx950 (C ABI part 2): the soft-assignment passes and the
// hypothesise-and-verify / weighted-Procrustes pose solvers.
//
// Replaces compute_coarse_Rt_overlap (core/unopose/utils/model_utils.py:411-490) and
// compute_fine_Rt_overlap (:527-566).  The reference makes ~10 element-wise passes over the
// (B,n1+1,n2+1) similarity (16.8 MB per pair in the fine stage) plus a (B,N,N) distance matrix and
// torch.svd on 6000*B 3x3 matrices; here the assignment is reduced in 4-5 streaming passes that only
// ever write O(n) statistics, and every SVD is a register Jacobi (jacobi3.h).
//
// Notation: x = atten (B,R,C) with R = n1+1, C = n2+1 (row/col 0 = background token),
//   a_ij = softmax_row(x)_ij * softmax_col(x)_ij * s1_i * s2_j          (s*_0 = 1)
//   w1_i = [argmax_j a_ij > 0]  (i >= 1),   w2_j = [argmax_i a_ij > 0]  (j >= 1)
#include "common.h"
#include "jacobi3.h"

namespace unopose {

// exp(x) for x <= 0 on the transcendental unit: v_exp_f32(x * log2 e), ~2 ulp; the softmax statistics and
// every later use of them go through the same function, so the normalisation stays self-consistent.  (The
// libm expf + IEEE divisions cost ~60 VALU instructions per matrix element and made these passes
// VALU-bound at 1.5-4 TB/s instead of HBM-bound.)
__device__ __forceinline__ float ph_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

// ---- pass 1a: per-row max and sum-exp (one wavefront per row)
__global__ __launch_bounds__(256) void row_stats_kernel(const float *__restrict__ x, int R, int C,
                                                        float *__restrict__ rmax, float *__restrict__ rsum) {
  const int b = blockIdx.y, lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= R) return;
  const float *row = x + ((size_t)b * R + i) * C;
  float m = -__builtin_inff();
  for (int j = lane; j < C; j += 64) m = fmaxf(m, row[j]);
  m = wave_max_f32(m);
  float s = 0.f;
  for (int j = lane; j < C; j += 64) s += ph_exp(row[j] - m);
  s = wave_sum_f32(s);
  if (lane == 0) {
    rmax[(size_t)b * R + i] = m;
    rsum[(size_t)b * R + i] = 1.f / s;  // the RECIPROCAL of the sum is what every consumer multiplies by
  }
}

// ---- pass 1b: per-column max and sum-exp; block = 64 columns x 4 row groups, coalesced 256-B reads
__global__ __launch_bounds__(256) void col_stats_kernel(const float *__restrict__ x, int R, int C,
                                                        float *__restrict__ cmax, float *__restrict__ csum) {
  __shared__ float sm[4][64], ss[4][64];
  const int b = blockIdx.y, tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + tx;
  const float *X = x + (size_t)b * R * C;
  float m = -__builtin_inff(), s = 0.f;
  if (j < C) {
    for (int i = ty; i < R; i += 4) {
      const float v = X[(size_t)i * C + j];
      if (v > m) {
        s = s * ph_exp(m - v) + 1.f;
        m = v;
      } else {
        s += ph_exp(v - m);
      }
    }
  }
  sm[ty][tx] = m;
  ss[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && j < C) {
    float M = fmaxf(fmaxf(sm[0][tx], sm[1][tx]), fmaxf(sm[2][tx], sm[3][tx]));
    float S = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) S += ss[g][tx] * ph_exp(sm[g][tx] - M);
    cmax[(size_t)b * C + j] = M;
    csum[(size_t)b * C + j] = 1.f / S;
  }
}

__device__ __forceinline__ float assign_val(float v, float rm, float irs, float cm, float ics, float s1, float s2) {
  // ((softmax_row * softmax_col) * s1) * s2, the reference's multiplication order
  return ((ph_exp(v - rm) * irs) * (ph_exp(v - cm) * ics)) * s1 * s2;  // irs, ics: reciprocal sums
}

// ---- pass 2a: w1_i = [max_{j>=1} a_ij > a_i0] for rows i >= 1 (first-index argmax tie rule)
__global__ __launch_bounds__(256) void row_label_kernel(const float *__restrict__ x, int R, int C,
                                                        const float *__restrict__ rmax,
                                                        const float *__restrict__ rsum,
                                                        const float *__restrict__ cmax,
                                                        const float *__restrict__ csum,
                                                        const float *__restrict__ score1,  // (B,R-1)
                                                        const float *__restrict__ score2,  // (B,C-1)
                                                        float *__restrict__ w1 /* (B,R-1) */) {
  const int b = blockIdx.y, lane = threadIdx.x & 63;
  const int i = 1 + blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= R) return;
  const float *row = x + ((size_t)b * R + i) * C;
  const float rm = rmax[(size_t)b * R + i], rs = rsum[(size_t)b * R + i];
  const float s1 = score1[(size_t)b * (R - 1) + i - 1];
  const float *CM = cmax + (size_t)b * C, *CS = csum + (size_t)b * C, *S2 = score2 + (size_t)b * (C - 1);
  float best = -1.f;
  for (int j = 1 + lane; j < C; j += 64) best = fmaxf(best, assign_val(row[j], rm, rs, CM[j], CS[j], s1, S2[j - 1]));
  best = wave_max_f32(best);
  const float a0 = assign_val(row[0], rm, rs, CM[0], CS[0], s1, 1.f);
  if (lane == 0) w1[(size_t)b * (R - 1) + i - 1] = best > a0 ? 1.f : 0.f;
}

// ---- pass 2b: w2_j = [max_{i>=1} a_ij > a_0j] for columns j >= 1
__global__ __launch_bounds__(256) void col_label_kernel(const float *__restrict__ x, int R, int C,
                                                        const float *__restrict__ rmax,
                                                        const float *__restrict__ rsum,
                                                        const float *__restrict__ cmax,
                                                        const float *__restrict__ csum,
                                                        const float *__restrict__ score1,
                                                        const float *__restrict__ score2,
                                                        float *__restrict__ w2 /* (B,C-1) */) {
  __shared__ float sb[4][64];
  const int b = blockIdx.y, tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int j = 1 + blockIdx.x * 64 + tx;
  const float *X = x + (size_t)b * R * C;
  const float *RM = rmax + (size_t)b * R, *RS = rsum + (size_t)b * R, *S1 = score1 + (size_t)b * (R - 1);
  float best = -1.f, cm = 0.f, cs = 1.f, s2 = 0.f;
  if (j < C) {
    cm = cmax[(size_t)b * C + j];
    cs = csum[(size_t)b * C + j];
    s2 = score2[(size_t)b * (C - 1) + j - 1];
    for (int i = 1 + ty; i < R; i += 4)
      best = fmaxf(best, assign_val(X[(size_t)i * C + j], RM[i], RS[i], cm, cs, S1[i - 1], s2));
  }
  sb[ty][tx] = best;
  __syncthreads();
  if (ty == 0 && j < C) {
    best = fmaxf(fmaxf(sb[0][tx], sb[1][tx]), fmaxf(sb[2][tx], sb[3][tx]));
    const float a0 = assign_val(X[j], RM[0], RS[0], cm, cs, 1.f, s2);
    w2[(size_t)b * (C - 1) + j - 1] = best > a0 ? 1.f : 0.f;
  }
}

// ---- pass 3 (fine): row weights and soft correspondences (model_utils.py:548-553)
//   A_ij = a_ij w1_i w2_j;  weight_i = sum_j A_ij;  pred_i = sum_j A_ij q_j / (weight_i + 1e-6)
// One wavefront owns FOUR rows: the per-column data (max, reciprocal sum, score, label, point) is loaded
// once per 64-column step and reused by the four rows (1 + 7/4 loads per element instead of 8: the
// one-row form was bound by the vector-memory pipe, not by HBM), and the two exponentials of a_ij are one:
//   a_ij = exp(2 x - rmax_i - cmax_j) * (s1_i w1_i / rsum_i) * (s2_j w2_j / csum_j)
__global__ __launch_bounds__(256) void fine_rows_kernel(const float *__restrict__ x, int R, int C,
                                                        const float *__restrict__ rmax,
                                                        const float *__restrict__ rsum,
                                                        const float *__restrict__ cmax,
                                                        const float *__restrict__ csum,
                                                        const float *__restrict__ score1,
                                                        const float *__restrict__ score2,
                                                        const float *__restrict__ w1, const float *__restrict__ w2,
                                                        const float *__restrict__ pts2,  // (B,C-1,3)
                                                        float *__restrict__ weight,      // (B,R-1)
                                                        float *__restrict__ pred /* (B,R-1,3) */) {
  constexpr float L2E = 1.4426950408889634f;
  const int b = blockIdx.y, lane = threadIdx.x & 63;
  const int i0 = 1 + (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;  // first of this wave's 4 rows
  if (i0 >= R) return;
  float rfac[4], rml[4];
  const float *row[4];
  bool any = false;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = min(i0 + k, R - 1);
    const size_t o = (size_t)b * (R - 1) + i - 1;
    const float wi = i0 + k < R ? w1[o] : 0.f;
    rfac[k] = rsum[(size_t)b * R + i] * score1[o] * wi;  // rsum holds 1 / sum
    rml[k] = rmax[(size_t)b * R + i] * L2E;
    row[k] = x + ((size_t)b * R + i) * C;
    any |= wi != 0.f;
  }
  float sw[4] = {0.f, 0.f, 0.f, 0.f}, px[4] = {0.f, 0.f, 0.f, 0.f}, py[4] = {0.f, 0.f, 0.f, 0.f}, pz[4] = {0.f, 0.f, 0.f, 0.f};
  if (any) {  // wave-uniform: four background rows in a row are skipped without reading them
    const float *CM = cmax + (size_t)b * C, *CS = csum + (size_t)b * C, *S2 = score2 + (size_t)b * (C - 1);
    const float *W2 = w2 + (size_t)b * (C - 1), *Q = pts2 + (size_t)b * (C - 1) * 3;
    for (int j = 1 + lane; j < C; j += 64) {
      const float g = CS[j] * S2[j - 1] * W2[j - 1];  // csum holds 1 / sum
      const float cml = CM[j] * L2E;
      const float qx = Q[(j - 1) * 3], qy = Q[(j - 1) * 3 + 1], qz = Q[(j - 1) * 3 + 2];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float e = __builtin_amdgcn_exp2f(fmaf(row[k][j], 2.f * L2E, -(rml[k] + cml)));
        const float a = e * g;
        sw[k] += a;
        px[k] = fmaf(a, qx, px[k]);
        py[k] = fmaf(a, qy, py[k]);
        pz[k] = fmaf(a, qz, pz[k]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float s = wave_sum_f32(sw[k]) * rfac[k], ax = wave_sum_f32(px[k]) * rfac[k];
    const float ay = wave_sum_f32(py[k]) * rfac[k], az = wave_sum_f32(pz[k]) * rfac[k];
    if (lane == 0 && i0 + k < R) {
      const size_t o = (size_t)b * (R - 1) + i0 + k - 1;
      weight[o] = s;
      const float inv = 1.f / (s + 1e-6f);
      pred[o * 3] = ax * inv;
      pred[o * 3 + 1] = ay * inv;
      pred[o * 3 + 2] = az * inv;
    }
  }
}

// ---- min_j |p_i - q_j| (direct differences), optional rigid transform p' = (p - t) R  (row vector)
__global__ __launch_bounds__(256) void min_dist_kernel(const float *__restrict__ p, const float *__restrict__ q, int N,
                                                       int M, const float *__restrict__ Rm /* (B,9) or null */,
                                                       const float *__restrict__ tv /* (B,3) or null */,
                                                       int cand_per_b, float *__restrict__ out /* (Bc,N) */) {
  extern __shared__ float4 smem4[];
  float *sq = reinterpret_cast<float *>(smem4);  // q staged SoA: [3][M]
  const int bc = blockIdx.y;                      // candidate index (b * cand_per_b + c)
  const int b = bc / cand_per_b;
  const float *Q = q + (size_t)b * M * 3;
  for (int e = threadIdx.x; e < M * 3; e += 256) {
    const int k = e / 3, comp = e - k * 3;
    sq[comp * M + k] = Q[e];
  }
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const float *P = p + ((size_t)b * N + i) * 3;
  float x = P[0], y = P[1], z = P[2];
  if (Rm) {
    const float *R = Rm + (size_t)bc * 9, *t = tv + (size_t)bc * 3;
    const float dx = x - t[0], dy = y - t[1], dz = z - t[2];
    x = dx * R[0] + dy * R[3] + dz * R[6];
    y = dx * R[1] + dy * R[4] + dz * R[7];
    z = dx * R[2] + dy * R[5] + dz * R[8];
  }
  float best = 3e38f;
  for (int k = 0; k < M; ++k) {
    const float dx = x - sq[k], dy = y - sq[M + k], dz = z - sq[2 * M + k];
    best = fminf(best, dx * dx + dy * dy + dz * dz);
  }
  out[(size_t)bc * N + i] = sqrtf(best);
}

// ---- coarse: ps = (a_ij w1_i w2_j)^1.5 over the (R-1)x(C-1) foreground block, then the CDF that
// torch builds with cumsum (CPU cumsum accumulates float in double) and normalises by last + 1e-8.
__global__ __launch_bounds__(512) void coarse_cdf_kernel(const float *__restrict__ x, int R, int C,
                                                         const float *__restrict__ rmax,
                                                         const float *__restrict__ rsum,
                                                         const float *__restrict__ cmax,
                                                         const float *__restrict__ csum,
                                                         const float *__restrict__ score1,
                                                         const float *__restrict__ score2,
                                                         const float *__restrict__ w1, const float *__restrict__ w2,
                                                         float *__restrict__ cdf /* (B,(R-1)*(C-1)) */) {
  __shared__ double part[512];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n1 = R - 1, n2 = C - 1, L = n1 * n2;
  const int per = (L + 511) / 512;
  const int e0 = tid * per, e1 = min(L, e0 + per);
  const float *X = x + (size_t)b * R * C;
  float *out = cdf + (size_t)b * L;
  double acc = 0.0;
  for (int e = e0; e < e1; ++e) {
    const int i = e / n2, j = e - i * n2;
    float v = 0.f;
    const float wi = w1[(size_t)b * n1 + i], wj = w2[(size_t)b * n2 + j];
    if (wi != 0.f && wj != 0.f) {
      const float a = assign_val(X[(size_t)(i + 1) * C + j + 1], rmax[(size_t)b * R + i + 1],
                                 rsum[(size_t)b * R + i + 1], cmax[(size_t)b * C + j + 1], csum[(size_t)b * C + j + 1],
                                 score1[(size_t)b * n1 + i], score2[(size_t)b * n2 + j]) * wi * wj;
      v = powf(a, 1.5f);
    }
    acc += (double)v;
    out[e] = v;  // provisional: the element itself
  }
  part[tid] = acc;
  __syncthreads();
  for (int off = 1; off < 512; off <<= 1) {  // Hillis-Steele inclusive scan of the 512 partials
    double add = tid >= off ? part[tid - off] : 0.0;
    __syncthreads();
    part[tid] += add;
    __syncthreads();
  }
  double run = tid ? part[tid - 1] : 0.0;
  const float last = (float)part[511];
  const float denom = last + 1e-8f;
  for (int e = e0; e < e1; ++e) {
    run += (double)out[e];
    out[e] = (float)run / denom;
  }
}

// ---- coarse: one thread per hypothesis: three CDF look-ups (searchsorted, left), 3-point Procrustes,
// mean residual (model_utils.py:462-474)
__global__ __launch_bounds__(256) void coarse_hypothesis_kernel(const float *__restrict__ cdf, int n1, int n2,
                                                                const float *__restrict__ rand, int nprop,
                                                                const float *__restrict__ pts1,
                                                                const float *__restrict__ pts2,
                                                                float *__restrict__ Rout, float *__restrict__ tout,
                                                                float *__restrict__ dis) {
  const int b = blockIdx.y;
  const int h = blockIdx.x * 256 + threadIdx.x;
  if (h >= nprop) return;
  const int L = n1 * n2;
  const float *cs = cdf + (size_t)b * L;
  const float *P1 = pts1 + (size_t)b * n1 * 3, *P2 = pts2 + (size_t)b * n2 * 3;
  float s[9], r[9];  // s = src (pts2 samples), r = ref (pts1 samples)
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float u = rand[((size_t)b * nprop + h) * 3 + c];
    int lo = 0, hi = L;  // first index with cs[idx] >= u
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cs[mid] < u) lo = mid + 1; else hi = mid;
    }
    const int i1 = min(lo / n2, n1 - 1), i2 = min(lo - (lo / n2) * n2, n2 - 1);
    r[c * 3] = P1[i1 * 3]; r[c * 3 + 1] = P1[i1 * 3 + 1]; r[c * 3 + 2] = P1[i1 * 3 + 2];
    s[c * 3] = P2[i2 * 3]; s[c * 3 + 1] = P2[i2 * 3 + 1]; s[c * 3 + 2] = P2[i2 * 3 + 2];
  }
  // weighted_procrustes(src = p2, ref = p1, w = 1 -> 1/(3+1e-5))
  const float wgt = 1.f / (3.f + 1e-5f);
  float sc[3], rc[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    sc[d] = s[d] * wgt + s[3 + d] * wgt + s[6 + d] * wgt;
    rc[d] = r[d] * wgt + r[3 + d] * wgt + r[6 + d] * wgt;
  }
  float H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float a0 = s[c * 3] - sc[0], a1 = s[c * 3 + 1] - sc[1], a2 = s[c * 3 + 2] - sc[2];
    const float b0 = wgt * (r[c * 3] - rc[0]), b1 = wgt * (r[c * 3 + 1] - rc[1]), b2 = wgt * (r[c * 3 + 2] - rc[2]);
    H[0] += a0 * b0; H[1] += a0 * b1; H[2] += a0 * b2;
    H[3] += a1 * b0; H[4] += a1 * b1; H[5] += a1 * b2;
    H[6] += a2 * b0; H[7] += a2 * b1; H[8] += a2 * b2;
  }
  float Rm[9];
  kabsch_from_H(H, Rm);
  float t[3];
  t[0] = rc[0] - (Rm[0] * sc[0] + Rm[1] * sc[1] + Rm[2] * sc[2]);
  t[1] = rc[1] - (Rm[3] * sc[0] + Rm[4] * sc[1] + Rm[5] * sc[2]);
  t[2] = rc[2] - (Rm[6] * sc[0] + Rm[7] * sc[1] + Rm[8] * sc[2]);
  // dis = mean_c |(p1_c - t) R - p2_c|
  float dsum = 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float dx = r[c * 3] - t[0], dy = r[c * 3 + 1] - t[1], dz = r[c * 3 + 2] - t[2];
    const float ex = dx * Rm[0] + dy * Rm[3] + dz * Rm[6] - s[c * 3];
    const float ey = dx * Rm[1] + dy * Rm[4] + dz * Rm[7] - s[c * 3 + 1];
    const float ez = dx * Rm[2] + dy * Rm[5] + dz * Rm[8] - s[c * 3 + 2];
    dsum += sqrtf(ex * ex + ey * ey + ez * ez);
  }
  const size_t o = (size_t)b * nprop + h;
  dis[o] = dsum / 3.f;
#pragma unroll
  for (int e = 0; e < 9; ++e) Rout[o * 9 + e] = Rm[e];
  tout[o * 3] = t[0]; tout[o * 3 + 1] = t[1]; tout[o * 3 + 2] = t[2];
}

// ---- coarse: candidate score = sum(w1) / (sum_i w1_i min_j |(p1_i - t) R - p2_j| + 1e-8)
// one workgroup per (pair, candidate); candidates are addressed through `top` (B,ncand) indices
__global__ __launch_bounds__(256) void coarse_score_kernel(const float *__restrict__ pts1,
                                                           const float *__restrict__ pts2, int n1, int n2,
                                                           const float *__restrict__ Rall,
                                                           const float *__restrict__ tall, int nprop,
                                                           const int64_t *__restrict__ top, int ncand,
                                                           const float *__restrict__ w1, float *__restrict__ score) {
  extern __shared__ float4 smem4[];
  float *sq = reinterpret_cast<float *>(smem4);  // [3][n2]
  __shared__ float red[4], redw[4];
  const int b = blockIdx.y, c = blockIdx.x, tid = threadIdx.x;
  const float *P1 = pts1 + (size_t)b * n1 * 3, *P2 = pts2 + (size_t)b * n2 * 3;
  for (int e = tid; e < n2 * 3; e += 256) {
    const int k = e / 3, comp = e - k * 3;
    sq[comp * n2 + k] = P2[e];
  }
  const size_t hyp = (size_t)b * nprop + (size_t)top[(size_t)b * ncand + c];
  const float *R = Rall + hyp * 9, *t = tall + hyp * 3;
  __syncthreads();
  float num = 0.f, den = 0.f;
  for (int i = tid; i < n1; i += 256) {
    const float wi = w1[(size_t)b * n1 + i];
    const float dx = P1[i * 3] - t[0], dy = P1[i * 3 + 1] - t[1], dz = P1[i * 3 + 2] - t[2];
    const float x = dx * R[0] + dy * R[3] + dz * R[6];
    const float y = dx * R[1] + dy * R[4] + dz * R[7];
    const float z = dx * R[2] + dy * R[5] + dz * R[8];
    float best = 3e38f;
    for (int k = 0; k < n2; ++k) {
      const float ex = x - sq[k], ey = y - sq[n2 + k], ez = z - sq[2 * n2 + k];
      best = fminf(best, ex * ex + ey * ey + ez * ez);
    }
    num += wi;
    den += wi * sqrtf(best);
  }
  num = wave_sum_f32(num);
  den = wave_sum_f32(den);
  if ((tid & 63) == 0) {
    redw[tid >> 6] = num;
    red[tid >> 6] = den;
  }
  __syncthreads();
  if (tid == 0) {
    const float N = (redw[0] + redw[1]) + (redw[2] + redw[3]);
    const float D = (red[0] + red[1]) + (red[2] + red[3]);
    score[(size_t)b * ncand + c] = N / (D + 1e-8f);
  }
}

// ---- training: gradient of the two-way InfoNCE ("atten") loss of compute_overlap_loss (loss_utils.py:181-187) w.r.t. the
// similarity matrix, from the streaming softmax statistics above:
//   L_b = 0.5 ( mean_{i>=1} CE(row i over ALL columns, label1[i-1]) + mean_{j>=1} CE(column j over ALL rows, label2[j-1]) )
//   dL_b / dx_ij = 0.5 ( [i>=1] (softmax_row(i,j) - [j == label1[i-1]]) / (R-1) + [j>=1] (softmax_col(i,j) - [i == label2[j-1]]) / (C-1) )
// scaled by the upstream gradient g[b].  One read and one write of the matrix; four elements per lane.
__global__ __launch_bounds__(256) void infonce_grad_kernel(const float *__restrict__ x, int R, int C, const float *__restrict__ rmax,
                                                           const float *__restrict__ rinv, const float *__restrict__ cmax,
                                                           const float *__restrict__ cinv, const long long *__restrict__ label1,
                                                           const long long *__restrict__ label2, const float *__restrict__ g,
                                                           float *__restrict__ grad) {
  const int b = blockIdx.z, i = blockIdx.y;
  const int j0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (j0 >= C) return;
  const size_t rowoff = ((size_t)b * R + i) * C;
  const float gb = 0.5f * g[b];
  const float wr = i >= 1 ? gb / (float)(R - 1) : 0.f, wc = gb / (float)(C - 1);
  const float rm = rmax[(size_t)b * R + i], ri = rinv[(size_t)b * R + i];
  const int lab1 = i >= 1 ? (int)label1[(size_t)b * (R - 1) + i - 1] : -1;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int j = j0 + u;
    if (j >= C) break;
    const float v = x[rowoff + j];
    float d = wr * (ph_exp(v - rm) * ri - (j == lab1 ? 1.f : 0.f));
    if (j >= 1) {
      const float pc = ph_exp(v - cmax[(size_t)b * C + j]) * cinv[(size_t)b * C + j];
      d += wc * (pc - ((int)label2[(size_t)b * (C - 1) + j - 1] == i ? 1.f : 0.f));
    }
    grad[rowoff + j] = d;
  }
}

// ---- the k smallest of n values per row, ascending, ties by index (torch.topk(dis, k, largest=False, sorted=True) of model_utils.py:476;
//      NaN counts as the largest value, as in torch).  Large k: rank by counting -- one thread per element compares its key with the row's n
//      keys in LDS (broadcast reads), no sort, no library launch; k <= TS_KMAX: the one-workgroup radix select below.
__device__ __forceinline__ uint32_t topk_key(float v) {
  if (v != v) return 0xFFFFFFFFu;
  const uint32_t u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // monotone float -> uint (below the NaN key: +inf maps to 0xFF800000)
}

__global__ __launch_bounds__(256) void topk_smallest_kernel(const float *__restrict__ x, int n, int k, int64_t *__restrict__ idx) {
  extern __shared__ uint32_t tk_keys[];
  const int b = blockIdx.y;
  const float *row = x + (size_t)b * n;
  for (int j = threadIdx.x; j < n; j += 256) tk_keys[j] = topk_key(row[j]);
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t ki = tk_keys[i];
  int rank = 0;
  for (int j = 0; j < n; ++j) {
    const uint32_t kj = tk_keys[j];
    rank += (kj < ki || (kj == ki && j < i)) ? 1 : 0;
  }
  if (rank < k) idx[(size_t)b * k + rank] = i;
}

// The same selection for k <= TS_KMAX with ONE workgroup per row (the counting kernel above spends n^2 comparisons per row on the whole chip --
// 6.4 k CU-microseconds per 32 pairs, which a co-running stream pays for; this one ~0.5 k): radix select of the k-th smallest key (four 8-bit
// histogram rounds over the keys in LDS), compaction of the keys below it in index order + the lowest-index ties, then the k chosen entries
// ranked among themselves.
constexpr int TS_KMAX = 2048;
__device__ __forceinline__ int ts_block_excl_scan(int v, int *wtot, int tid) {  // exclusive prefix sum over the 256 threads; wtot: 4 ints of LDS
  const int lane = tid & 63, wave = tid >> 6;
  int incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d);
    if (lane >= d) incl += t;
  }
  __syncthreads();  // (wtot may still be read from the previous scan)
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += wtot[w];
  return base + incl - v;
}

__global__ __launch_bounds__(256) void topk_select_kernel(const float *__restrict__ x, int n, int k, int64_t *__restrict__ idx) {
  extern __shared__ uint32_t tk_keys[];  // [n] keys, then [k] chosen keys, [k] chosen indices
  __shared__ int hist[256], wtot[4], sh_bin, sh_need;
  uint32_t *sel_key = tk_keys + n;
  int *sel_idx = reinterpret_cast<int *>(sel_key + k);
  const int b = blockIdx.x, tid = threadIdx.x;
  const float *row = x + (size_t)b * n;
  for (int j = tid; j < n; j += 256) tk_keys[j] = topk_key(row[j]);
  // ---- the k-th smallest key T, and how many of the keys equal to it belong to the k
  uint32_t prefix = 0u, mask = 0u;
  int need = k;
  for (int shift = 24; shift >= 0; shift -= 8) {
    hist[tid] = 0;
    __syncthreads();
    for (int j = tid; j < n; j += 256) {
      const uint32_t key = tk_keys[j];
      if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1);
    }
    __syncthreads();
    const int c = hist[tid], excl = ts_block_excl_scan(c, wtot, tid);
    if (excl < need && need <= excl + c) sh_bin = tid, sh_need = need - excl;  // exactly one thread
    __syncthreads();
    prefix |= (uint32_t)sh_bin << shift;
    mask |= 0xFFu << shift;
    need = sh_need;
  }
  const uint32_t T = prefix;
  const int c_lt = k - need;  // keys strictly below T; `need` of the keys equal to T follow, lowest indices first
  // ---- compaction in index order: thread t owns the indices [t chunk, (t + 1) chunk)
  const int chunk = (n + 255) / 256, j0 = min(tid * chunk, n), j1 = min(j0 + chunk, n);
  int a = 0, e = 0;
  for (int j = j0; j < j1; ++j) {
    const uint32_t key = tk_keys[j];
    a += key < T;
    e += key == T;
  }
  int a_off = ts_block_excl_scan(a, wtot, tid), e_off = ts_block_excl_scan(e, wtot, tid);
  for (int j = j0; j < j1; ++j) {
    const uint32_t key = tk_keys[j];
    if (key < T) {
      sel_key[a_off] = key, sel_idx[a_off] = j;
      ++a_off;
    } else if (key == T) {
      if (e_off < need) idx[(size_t)b * k + c_lt + e_off] = j;  // the ties are already in their final places
      ++e_off;
    }
  }
  __syncthreads();
  // ---- the c_lt entries below T, ranked among themselves (they sit in index order: position breaks ties)
  for (int i = tid; i < c_lt; i += 256) {
    const uint32_t ki = sel_key[i];
    int rank = 0;
    for (int m = 0; m < c_lt; ++m) {
      const uint32_t km = sel_key[m];
      rank += (km < ki || (km == ki && m < i)) ? 1 : 0;
    }
    idx[(size_t)b * k + rank] = sel_idx[i];
  }
}

// ---- the winning hypothesis of a pair (model_utils.py:486-490: pose_score.max(1), then the gathers of R and t): the first maximum of
//      score[b, :] (a NaN wins, the first one, as in torch.max), its hypothesis top[b, best], that hypothesis's R and t.
__global__ __launch_bounds__(64) void coarse_pick_kernel(const float *__restrict__ score, const int64_t *__restrict__ top, int ncand,
                                                         const float *__restrict__ Rall, const float *__restrict__ tall, int nprop,
                                                         float *__restrict__ R, float *__restrict__ t, float *__restrict__ best_score) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const float *sc = score + (size_t)b * ncand;
  uint32_t bk = 0u;
  int bi = 0x7FFFFFFF;
  for (int c = lane; c < ncand; c += 64) {
    const uint32_t kc = topk_key(sc[c]);
    if (kc > bk || (kc == bk && c < bi)) bk = kc, bi = c;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const uint32_t ok = __shfl_xor(bk, o);
    const int oi = __shfl_xor(bi, o);
    if (ok > bk || (ok == bk && oi < bi)) bk = ok, bi = oi;
  }
  const long h = (long)top[(size_t)b * ncand + bi];
  if (lane < 9) R[(size_t)b * 9 + lane] = Rall[((size_t)b * nprop + h) * 9 + lane];
  if (lane < 3) t[(size_t)b * 3 + lane] = tall[((size_t)b * nprop + h) * 3 + lane];
  if (lane == 0) best_score[b] = sc[bi];
}

}  // namespace unopose

using namespace unopose;

extern "C" {

// The streaming softmax statistics alone (training: the two-way InfoNCE loss needs row and column log-sum-exps of the similarity).
int unopose_softmax_stats(const float *x, int B, int R, int C, float *stats_ws, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(x && stats_ws, "softmax_stats: null pointer");
  UNOPOSE_REQUIRE(B >= 0 && R >= 1 && C >= 1 && B <= 65535, "softmax_stats: bad sizes");
  if (B == 0) return UNOPOSE_OK;
  hipStream_t s = (hipStream_t)stream;
  float *rmax = stats_ws, *rsum = rmax + (size_t)B * R, *cmax = rsum + (size_t)B * R, *csum = cmax + (size_t)B * C;
  hipLaunchKernelGGL(row_stats_kernel, dim3(cdiv(R, 4), B), dim3(256), 0, s, x, R, C, rmax, rsum);
  hipLaunchKernelGGL(col_stats_kernel, dim3(cdiv(C, 64), B), dim3(256), 0, s, x, R, C, cmax, csum);
  return check_launch("softmax_stats");
}

int unopose_infonce_grad(const float *x, int B, int R, int C, const float *stats_ws, const long long *label1, const long long *label2,
                         const float *g, float *grad, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(x && stats_ws && label1 && label2 && g && grad, "infonce_grad: null pointer");
  UNOPOSE_REQUIRE(B >= 0 && R >= 2 && C >= 2 && B <= 65535 && R <= 65535, "infonce_grad: bad sizes");
  if (B == 0) return UNOPOSE_OK;
  const float *rmax = stats_ws, *rsum = rmax + (size_t)B * R, *cmax = rsum + (size_t)B * R, *csum = cmax + (size_t)B * C;
  hipLaunchKernelGGL(infonce_grad_kernel, dim3(cdiv(C, 1024), R, B), dim3(256), 0, (hipStream_t)stream, x, R, C, rmax, rsum, cmax, csum,
                     label1, label2, g, grad);
  return check_launch("infonce_grad");
}

// Statistics of the soft assignment: ws = 2*(R+C) floats per batch element
// [rmax R | rsum R | cmax C | csum C]; w1 (B,R-1), w2 (B,C-1).
int unopose_assign_labels(const float *atten, int B, int R, int C, const float *score1, const float *score2,
                          float *stats_ws, float *w1, float *w2, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(atten && score1 && score2 && stats_ws && w1 && w2, "assign_labels: null pointer");
  UNOPOSE_REQUIRE(B >= 0 && R >= 2 && C >= 2 && B <= 65535, "assign_labels: bad sizes");
  if (B == 0) return UNOPOSE_OK;
  hipStream_t s = (hipStream_t)stream;
  float *rmax = stats_ws, *rsum = rmax + (size_t)B * R, *cmax = rsum + (size_t)B * R, *csum = cmax + (size_t)B * C;
  hipLaunchKernelGGL(row_stats_kernel, dim3(cdiv(R, 4), B), dim3(256), 0, s, atten, R, C, rmax, rsum);
  hipLaunchKernelGGL(col_stats_kernel, dim3(cdiv(C, 64), B), dim3(256), 0, s, atten, R, C, cmax, csum);
  hipLaunchKernelGGL(row_label_kernel, dim3(cdiv(R - 1, 4), B), dim3(256), 0, s, atten, R, C, rmax, rsum, cmax, csum,
                     score1, score2, w1);
  hipLaunchKernelGGL(col_label_kernel, dim3(cdiv(C - 1, 64), B), dim3(256), 0, s, atten, R, C, rmax, rsum, cmax, csum,
                     score1, score2, w2);
  return check_launch("assign_labels");
}

int unopose_fine_correspondences(const float *atten, int B, int R, int C, const float *score1, const float *score2,
                                 const float *stats_ws, const float *w1, const float *w2, const float *pts2,
                                 float *weight, float *pred, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(atten && score1 && score2 && stats_ws && w1 && w2 && pts2 && weight && pred,
                  "fine_correspondences: null pointer");
  UNOPOSE_REQUIRE(B >= 0 && R >= 2 && C >= 2 && B <= 65535, "fine_correspondences: bad sizes");
  if (B == 0) return UNOPOSE_OK;
  const float *rmax = stats_ws, *rsum = rmax + (size_t)B * R, *cmax = rsum + (size_t)B * R,
              *csum = cmax + (size_t)B * C;
  hipLaunchKernelGGL(fine_rows_kernel, dim3(cdiv(R - 1, 16), B), dim3(256), 0, (hipStream_t)stream, atten, R, C, rmax,
                     rsum, cmax, csum, score1, score2, w1, w2, pts2, weight, pred);
  return check_launch("fine_correspondences");
}

int unopose_min_dist(const float *p, const float *q, int B, int N, int M, const float *R, const float *t,
                     int cand_per_b, float *out, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(p && q && out && (!R == !t), "min_dist: null pointer");
  UNOPOSE_REQUIRE(B >= 0 && N >= 1 && M >= 1 && cand_per_b >= 1 && (long)B * cand_per_b <= 65535 &&
                      (size_t)M * 12 <= 64 * 1024, "min_dist: bad sizes");
  if (B == 0) return UNOPOSE_OK;
  hipLaunchKernelGGL(min_dist_kernel, dim3(cdiv(N, 256), B * cand_per_b), dim3(256), (size_t)M * 12,
                     (hipStream_t)stream, p, q, N, M, R, t, cand_per_b, out);
  return check_launch("min_dist");
}

int unopose_coarse_hypotheses(const float *atten, int B, int R, int C, const float *score1, const float *score2,
                              const float *stats_ws, const float *w1, const float *w2, const float *rand, int nprop,
                              const float *pts1, const float *pts2, float *cdf_ws, float *Rout, float *tout,
                              float *dis, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(atten && score1 && score2 && stats_ws && w1 && w2 && rand && pts1 && pts2 && cdf_ws && Rout &&
                      tout && dis, "coarse_hypotheses: null pointer");
  UNOPOSE_REQUIRE(B >= 0 && R >= 2 && C >= 2 && nprop >= 1 && B <= 65535, "coarse_hypotheses: bad sizes");
  if (B == 0) return UNOPOSE_OK;
  hipStream_t s = (hipStream_t)stream;
  const float *rmax = stats_ws, *rsum = rmax + (size_t)B * R, *cmax = rsum + (size_t)B * R,
              *csum = cmax + (size_t)B * C;
  hipLaunchKernelGGL(coarse_cdf_kernel, dim3(B), dim3(512), 0, s, atten, R, C, rmax, rsum, cmax, csum, score1, score2,
                     w1, w2, cdf_ws);
  hipLaunchKernelGGL(coarse_hypothesis_kernel, dim3(cdiv(nprop, 256), B), dim3(256), 0, s, cdf_ws, R - 1, C - 1, rand,
                     nprop, pts1, pts2, Rout, tout, dis);
  return check_launch("coarse_hypotheses");
}

int unopose_coarse_scores(const float *pts1, const float *pts2, int B, int n1, int n2, const float *Rall,
                          const float *tall, int nprop, const int64_t *top, int ncand, const float *w1, float *score,
                          unopose_stream_t stream) {
  UNOPOSE_REQUIRE(pts1 && pts2 && Rall && tall && top && w1 && score, "coarse_scores: null pointer");
  UNOPOSE_REQUIRE(B >= 0 && n1 >= 1 && n2 >= 1 && ncand >= 1 && B <= 65535 && (size_t)n2 * 12 <= 60 * 1024,
                  "coarse_scores: bad sizes");
  if (B == 0) return UNOPOSE_OK;
  hipLaunchKernelGGL(coarse_score_kernel, dim3(ncand, B), dim3(256), (size_t)n2 * 12, (hipStream_t)stream, pts1, pts2,
                     n1, n2, Rall, tall, nprop, top, ncand, w1, score);
  return check_launch("coarse_scores");
}

int unopose_topk_smallest(const float *x, int B, int n, int k, int64_t *idx, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(x && idx, "topk_smallest: null pointer");
  UNOPOSE_REQUIRE(B >= 0 && B <= 65535 && n >= 1 && k >= 1 && k <= n && (size_t)n * 4 <= 64 * 1024, "topk_smallest: bad sizes (n=%d k=%d; n <= 16384)", n, k);
  if (B == 0) return UNOPOSE_OK;
  if (k <= TS_KMAX && (size_t)(n + 2 * k) * 4 <= 60 * 1024)  // one workgroup per row: radix select + a k x k ranking
    hipLaunchKernelGGL(topk_select_kernel, dim3(B), dim3(256), (size_t)(n + 2 * k) * 4, (hipStream_t)stream, x, n, k, idx);
  else  // rank by counting over the whole row
    hipLaunchKernelGGL(topk_smallest_kernel, dim3(cdiv(n, 256), B), dim3(256), (size_t)n * 4, (hipStream_t)stream, x, n, k, idx);
  return check_launch("topk_smallest");
}

int unopose_coarse_pick(const float *score, const int64_t *top, int B, int ncand, const float *Rall, const float *tall, int nprop, float *R,
                        float *t, float *best_score, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(score && top && Rall && tall && R && t && best_score, "coarse_pick: null pointer");
  UNOPOSE_REQUIRE(B >= 0 && ncand >= 1 && nprop >= 1, "coarse_pick: bad sizes");
  if (B == 0) return UNOPOSE_OK;
  hipLaunchKernelGGL(coarse_pick_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, score, top, ncand, Rall, tall, nprop, R, t, best_score);
  return check_launch("coarse_pick");
}

}  // extern "C"
