// Saliency of the matchers' training branch (C ABI part 3: training path).
//
// Replaces, per transformer block of core/unopose/model/oneref_predator_coarse_point_matching.py:68-76 /
// oneref_predator_fine_point_matching.py:91-99 under train(),
//     inner = atten[:, 1:, 1:];  m1 = softmax(inner, dim=2) @ s2;  m2 = softmax(inner.transpose(1, 2), dim=2) @ s1
// and its autograd backward.  Through torch that is, at the fine stage (8 x 4096 x 4096 fp32 = 537 MB per tensor): two strided copies
// of the similarity, two softmax outputs, two matrix-vector products, and in backward two softmax gradients plus the adds that merge
// three dense gradients of the similarity -- a dozen passes per block.  Here: one pass over the rows and one over the columns forward
// (online softmax statistics + the weighted sum: nothing of the matrix's size is written), one pass that writes the similarity's gradient
// and one that collects the column sums backward.  Everything is a deterministic reduction (no atomics).
//   forward   m1_i = sum_j p_ij s2_j,  p_ij = exp(a_ij - R_i) / Z_i   (row statistics R, Z kept for backward)
//             m2_j = sum_i q_ij s1_i,  q_ij = exp(a_ij - C_j) / W_j   (column statistics C, W kept)
//   backward  da_ij = g1_i p_ij (s2_j - m1_i) + g2_j q_ij (s1_i - m2_j);  ds2_j = sum_i g1_i p_ij;  ds1_i = sum_j g2_j q_ij
#include "common.h"

namespace unopose {

// one wave per row i of inner (row i + 1 of atten, columns 1 ..): m1, R, Z
__global__ __launch_bounds__(256) void saliency_rows_kernel(const float *__restrict__ atten, const float *__restrict__ s2, int n1, int n2,
                                                            float *__restrict__ m1, float *__restrict__ rmax, float *__restrict__ rsum) {
  const int b = blockIdx.y, lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n1) return;
  const float *row = atten + ((size_t)b * (n1 + 1) + i + 1) * (n2 + 1) + 1;
  const float *sv = s2 + (size_t)b * n2;
  float mx = -3.4e38f;
#pragma unroll 8
  for (int j = lane; j < n2; j += 64) mx = fmaxf(mx, row[j]);
  mx = wave_max_f32(mx);
  float z = 0.f, acc = 0.f;
#pragma unroll 8
  for (int j = lane; j < n2; j += 64) {
    const float e = __expf(row[j] - mx);
    z += e;
    acc += e * sv[j];
  }
  z = wave_sum_f32(z);
  acc = wave_sum_f32(acc);
  if (lane == 0) {
    m1[(size_t)b * n1 + i] = acc / z;
    rmax[(size_t)b * n1 + i] = mx;
    rsum[(size_t)b * n1 + i] = z;
  }
}

// 64 columns per workgroup, four row phases per column: online (max, sum, weighted sum), combined through LDS in a fixed order
template <bool BWD>  // false: m2, C, W from s1;  true: ds2_j = sum_i g1_i p_ij (row statistics given)
__global__ __launch_bounds__(256) void saliency_cols_kernel(const float *__restrict__ atten, const float *__restrict__ v, const float *__restrict__ rmax,
                                                            const float *__restrict__ rsum, int n1, int n2, float *__restrict__ out,
                                                            float *__restrict__ cmax, float *__restrict__ csum) {
  __shared__ float sm[4][64], sl[4][64], sa[4][64];
  const int b = blockIdx.y, c = threadIdx.x & 63, ph = threadIdx.x >> 6, j = blockIdx.x * 64 + c;
  const bool ok = j < n2;
  const float *col = atten + ((size_t)b * (n1 + 1) + 1) * (n2 + 1) + 1 + (ok ? j : 0);
  const float *vv = v + (size_t)b * n1;
  float m = -3.4e38f, l = 0.f, acc = 0.f;
  if (BWD) {
    const float *rm = rmax + (size_t)b * n1, *rs = rsum + (size_t)b * n1;
#pragma unroll 8
    for (int i = ph; i < n1; i += 4) acc += vv[i] * (__expf(col[(size_t)i * (n2 + 1)] - rm[i]) / rs[i]);
    sa[ph][c] = acc;
    __syncthreads();
    if (ph == 0 && ok) out[(size_t)b * n2 + j] = (sa[0][c] + sa[1][c]) + (sa[2][c] + sa[3][c]);
    return;
  }
#pragma unroll 8
  for (int i = ph; i < n1; i += 4) {
    const float a = col[(size_t)i * (n2 + 1)];
    if (a > m) {
      const float r = __expf(m - a);
      l *= r, acc *= r, m = a;
    }
    const float e = __expf(a - m);
    l += e;
    acc += e * vv[i];
  }
  sm[ph][c] = m, sl[ph][c] = l, sa[ph][c] = acc;
  __syncthreads();
  if (ph == 0 && ok) {
    float M = fmaxf(fmaxf(sm[0][c], sm[1][c]), fmaxf(sm[2][c], sm[3][c])), L = 0.f, A = 0.f;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const float r = __expf(sm[p][c] - M);
      L += sl[p][c] * r;
      A += sa[p][c] * r;
    }
    out[(size_t)b * n2 + j] = A / L;
    cmax[(size_t)b * n2 + j] = M;
    csum[(size_t)b * n2 + j] = L;
  }
}

// one wave per row of atten (row 0 and column 0 receive zeros): da and ds1
__global__ __launch_bounds__(256) void saliency_bwd_rows_kernel(const float *__restrict__ atten, const float *__restrict__ s1, const float *__restrict__ s2,
                                                                const float *__restrict__ m1, const float *__restrict__ m2,
                                                                const float *__restrict__ rmax, const float *__restrict__ rsum,
                                                                const float *__restrict__ cmax, const float *__restrict__ csum,
                                                                const float *__restrict__ g1, const float *__restrict__ g2, int n1, int n2,
                                                                float *__restrict__ da, float *__restrict__ ds1) {
  const int b = blockIdx.y, lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);  // row of atten, 0 .. n1
  if (r > n1) return;
  float *drow = da + ((size_t)b * (n1 + 1) + r) * (n2 + 1);
  if (r == 0) {
    for (int j = lane; j <= n2; j += 64) drow[j] = 0.f;
    return;
  }
  const int i = r - 1;
  const float *row = atten + ((size_t)b * (n1 + 1) + r) * (n2 + 1) + 1;
  const size_t bi = (size_t)b * n1 + i, b2 = (size_t)b * n2;
  const float R = rmax[bi], iz = 1.f / rsum[bi], mm1 = m1[bi], s1i = s1[bi], g1i = g1[bi];
  float acc = 0.f;
  if (lane == 0) drow[0] = 0.f;
#pragma unroll 4
  for (int j = lane; j < n2; j += 64) {
    const float a = row[j];
    const float p = __expf(a - R) * iz, q = __expf(a - cmax[b2 + j]) / csum[b2 + j];
    const float gq = g2[b2 + j] * q;
    drow[1 + j] = g1i * p * (s2[b2 + j] - mm1) + gq * (s1i - m2[b2 + j]);
    acc += gq;
  }
  acc = wave_sum_f32(acc);
  if (lane == 0) ds1[bi] = acc;
}

}  // namespace unopose

using namespace unopose;

extern "C" {

int unopose_saliency_train_forward(const float *atten, const float *s1, const float *s2, int B, int n1, int n2, float *m1, float *m2, float *rmax,
                                   float *rsum, float *cmax, float *csum, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(atten && s1 && s2 && m1 && m2 && rmax && rsum && cmax && csum, "saliency_train_forward: null pointer");
  UNOPOSE_REQUIRE(B >= 1 && B <= 65535 && n1 >= 1 && n2 >= 1, "saliency_train_forward: bad sizes (B=%d n1=%d n2=%d)", B, n1, n2);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(saliency_rows_kernel, dim3(cdiv(n1, 4), B), dim3(256), 0, s, atten, s2, n1, n2, m1, rmax, rsum);
  hipLaunchKernelGGL(saliency_cols_kernel<false>, dim3(cdiv(n2, 64), B), dim3(256), 0, s, atten, s1, (const float *)nullptr, (const float *)nullptr, n1, n2,
                     m2, cmax, csum);
  return check_launch("saliency_train_forward");
}

int unopose_saliency_train_backward(const float *atten, const float *s1, const float *s2, const float *m1, const float *m2, const float *rmax,
                                    const float *rsum, const float *cmax, const float *csum, const float *g1, const float *g2, int B, int n1, int n2,
                                    float *d_atten, float *ds1, float *ds2, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(atten && s1 && s2 && m1 && m2 && rmax && rsum && cmax && csum && g1 && g2 && d_atten && ds1 && ds2,
                  "saliency_train_backward: null pointer");
  UNOPOSE_REQUIRE(B >= 1 && B <= 65535 && n1 >= 1 && n2 >= 1, "saliency_train_backward: bad sizes (B=%d n1=%d n2=%d)", B, n1, n2);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(saliency_bwd_rows_kernel, dim3(cdiv(n1 + 1, 4), B), dim3(256), 0, s, atten, s1, s2, m1, m2, rmax, rsum, cmax, csum, g1, g2, n1, n2,
                     d_atten, ds1);
  hipLaunchKernelGGL(saliency_cols_kernel<true>, dim3(cdiv(n2, 64), B), dim3(256), 0, s, atten, g1, rmax, rsum, n1, n2, ds2, (float *)nullptr,
                     (float *)nullptr);
  return check_launch("saliency_train_backward");
}

}  // extern "C"
