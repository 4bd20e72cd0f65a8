// Small batched fp32 contraction on the exact-fp32 matrix instruction (C ABI part 2d):
//     C[b][i][j] = alpha * sum_k A[b][i][k] * Bm[b][j][k]       (arbitrary element strides for both operands; the batch index
//     b = bo * bi + bi_ addresses the operands as bo * s?b + bi_ * s?h: (pair, head) batches of interleaved heads)
// v_mfma_f32_32x32x2_f32 is a k-ordered fp32 fma chain (one rounding per product: the arithmetic of an fp32 library GEMM) at the
// fp32 vector rate -- plenty for the three contractions that stayed on library kernels: the coarse feature similarity
// (model_utils.py:260-282: 197 x 197 x 256 per pair), the focused linear attention's k^T v (transformer.py:560-566: 64 x 64 x 196
// per head) and, in fp32 mode, rotations of point sets.  One wavefront per 32 x 32 output block.
#include "common.h"

namespace unopose {

template <bool KVEC>  // KVEC: both operands have unit k stride and 16-byte aligned rows -> 8 consecutive k per lane and load
__global__ __launch_bounds__(64) void bmm_f32_kernel(const float *__restrict__ A, long sab, long sah, long sai, long sak,
                                                     const float *__restrict__ Bm, long sbb, long sbh, long sbj, long sbk,
                                                     float *__restrict__ C, int bi, int n, int m, int K, float alpha) {
  const int lane = threadIdx.x, l31 = lane & 31, hi = lane >> 5;
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32, b = blockIdx.z;
  const int bo = b / bi, bh = b - bo * bi;
  const float *ap = A + bo * sab + bh * sah + (long)min(i0 + l31, n - 1) * sai;
  const float *bp = Bm + bo * sbb + bh * sbh + (long)min(j0 + l31, m - 1) * sbj;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  int k = 0;
  if (KVEC) {
    for (; k + 8 <= K; k += 8) {
      const float4 a0 = *reinterpret_cast<const float4 *>(ap + k), a1 = *reinterpret_cast<const float4 *>(ap + k + 4);
      const float4 b0 = *reinterpret_cast<const float4 *>(bp + k), b1 = *reinterpret_cast<const float4 *>(bp + k + 4);
      // lane (row, hi) feeds k + 2 t + hi of k-step t
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(hi ? a0.y : a0.x, hi ? b0.y : b0.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(hi ? a0.w : a0.z, hi ? b0.w : b0.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(hi ? a1.y : a1.x, hi ? b1.y : b1.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(hi ? a1.w : a1.z, hi ? b1.w : b1.z, acc, 0, 0, 0);
    }
  }
  for (; k < K; k += 2) {
    const int kk = k + hi;
    const float a = kk < K ? ap[(long)kk * sak] : 0.f, bv = kk < K ? bp[(long)kk * sbk] : 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc, 0, 0, 0);
  }
  // D[i][j]: lane holds column j = l31, rows (r & 3) + 8 (r >> 2) + 4 hi
  const int j = j0 + l31;
  if (j < m) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = i0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (i < n) C[((long)b * n + i) * m + j] = acc[r] * alpha;
    }
  }
}

// The same contraction for large outputs (the fp32 path's 2049 x 2049 x 256 fine similarity per pair): one wavefront per 64 x 64 block,
// four accumulators over the same operand registers -- half the operand bytes per MFMA of the 32 x 32 form, which at 4 flop per
// L2 byte was bound by the operand reads.  Every output element sees the same k-ordered fma chain, so the result is bit-identical.
__global__ __launch_bounds__(64) void bmm_f32_kernel64(const float *__restrict__ A, long sab, long sah, long sai, const float *__restrict__ Bm, long sbb,
                                                       long sbh, long sbj, float *__restrict__ C, int bi, int n, int m, int K, float alpha) {
  const int lane = threadIdx.x, l31 = lane & 31, hi = lane >> 5;
  const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64, b = blockIdx.z;
  const int bo = b / bi, bh = b - bo * bi;
  const float *ap[2], *bp[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    ap[u] = A + bo * sab + bh * sah + (long)min(i0 + 32 * u + l31, n - 1) * sai;
    bp[u] = Bm + bo * sbb + bh * sbh + (long)min(j0 + 32 * u + l31, m - 1) * sbj;
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[u][v][r] = 0.f;
  int k = 0;
  for (; k + 8 <= K; k += 8) {
    float a[2][4], bq[2][4];  // lane (row, hi) feeds k + 2 t + hi of k-step t
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float4 a0 = *reinterpret_cast<const float4 *>(ap[u] + k), a1 = *reinterpret_cast<const float4 *>(ap[u] + k + 4);
      const float4 b0 = *reinterpret_cast<const float4 *>(bp[u] + k), b1 = *reinterpret_cast<const float4 *>(bp[u] + k + 4);
      a[u][0] = hi ? a0.y : a0.x; a[u][1] = hi ? a0.w : a0.z; a[u][2] = hi ? a1.y : a1.x; a[u][3] = hi ? a1.w : a1.z;
      bq[u][0] = hi ? b0.y : b0.x; bq[u][1] = hi ? b0.w : b0.z; bq[u][2] = hi ? b1.y : b1.x; bq[u][3] = hi ? b1.w : b1.z;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) acc[u][v] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][t], bq[v][t], acc[u][v], 0, 0, 0);
  }
  for (; k < K; k += 2) {
    const int kk = k + hi;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int v = 0; v < 2; ++v)
        acc[u][v] = __builtin_amdgcn_mfma_f32_32x32x2f32(kk < K ? ap[u][kk] : 0.f, kk < K ? bp[v][kk] : 0.f, acc[u][v], 0, 0, 0);
  }
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const int j = j0 + 32 * v + l31;
      if (j < m) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = i0 + 32 * u + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (i < n) C[((long)b * n + i) * m + j] = acc[u][v][r] * alpha;
        }
      }
    }
}

}  // namespace unopose

using namespace unopose;

extern "C" int unopose_bmm_f32(const float *A, long sab, long sah, long sai, long sak, const float *Bm, long sbb, long sbh, long sbj, long sbk,
                               float *C, int bo, int bi, int n, int m, int K, float alpha, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(A && Bm && C, "bmm_f32: null pointer");
  UNOPOSE_REQUIRE(bo >= 1 && bi >= 1 && (long)bo * bi <= 65535 && n >= 1 && m >= 1 && K >= 1, "bmm_f32: bad sizes (batch=%d x %d n=%d m=%d K=%d)",
                  bo, bi, n, m, K);
  const dim3 grid(cdiv(m, 32), cdiv(n, 32), bo * bi);
  const bool kvec = sak == 1 && sbk == 1 && (sai | sbj | sab | sbb | sah | sbh) % 4 == 0 &&
                    (reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(Bm)) % 16 == 0;
  if (kvec && n >= 256 && m >= 256)
    hipLaunchKernelGGL(bmm_f32_kernel64, dim3(cdiv(m, 64), cdiv(n, 64), bo * bi), dim3(64), 0, (hipStream_t)stream, A, sab, sah, sai, Bm, sbb, sbh, sbj, C, bi,
                       n, m, K, alpha);
  else if (kvec)
    hipLaunchKernelGGL(bmm_f32_kernel<true>, grid, dim3(64), 0, (hipStream_t)stream, A, sab, sah, sai, sak, Bm, sbb, sbh, sbj, sbk, C, bi, n, m, K, alpha);
  else
    hipLaunchKernelGGL(bmm_f32_kernel<false>, grid, dim3(64), 0, (hipStream_t)stream, A, sab, sah, sai, sak, Bm, sbb, sbh, sbj, sbk, C, bi, n, m, K, alpha);
  return check_launch("bmm_f32");
}
