// fp32-class "linear" GEMM for gfx950 (C ABI part 2b):  C[M][N] = act( A[M][K] . W[N][K]^T + bias[N] ) with fp32 data.
// The reference's default precision (configs/main_cfg.py:87-89: no autocast) runs every nn.Linear in fp32; gfx950 has no
// TF32-class matrix instruction and its exact-fp32 MFMA runs at 1/16 of the bf16 rate, so each operand is SPLIT into two
// bf16 numbers (hi = bf16(x), lo = bf16(x - hi): x = hi + lo to 2^-17 relative) and a product is three bf16 MFMAs with
// fp32 accumulation:  a w ~ ah wh + ah wl + al wh  (the dropped al wl term is 2^-18 relative) -- the form the attention,
// embedding and PE kernels of this library already use for their fp32 mode.
//
// Operand layout ("split" layout, the same bytes per element as fp32): row r, k-block j (32 consecutive k) is one 128-byte
// line [hi(k 32j .. 32j+31) | lo(k 32j .. 32j+31)], rows K * 4 bytes apart.  unopose_split_bf16x2 produces it from fp32
// rows; the GEMM can also WRITE its result in this layout (the next linear's input: fc1 -> fc2).
// The kernel IS the bf16 kernel (gemm_kernel.h, template parameter F32): a K-tile is 32 k -- the same 128-byte rows, hence the same
// half-tile LDS-DMA stream with counted waits, ping-pong wave groups, stream across tiles and dynamic tile tickets -- with 24 MFMAs per
// phase instead of 16 (hi.hi + hi.lo + lo.hi per fragment pair) and its own epilogues (fp32 / split-layout / bf16 + residual output).
// (Round 3's form of this kernel drained vmcnt(0) + barrier every stage; same-box numbers in DESIGN.md section 7.)
#include "gemm_kernel.h"

namespace unopose {

#define GEMMF_BK 32

// fp32 rows -> split layout.  One thread per 8 consecutive k (32 B in, 16 B hi + 16 B lo out).
__global__ __launch_bounds__(256) void split_bf16x2_kernel(const float *__restrict__ X, u16 *__restrict__ Xs, long total8, int K) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total8) return;
  const int k8 = K >> 3;
  const long r = i / k8;
  const int c = (int)(i - r * k8);  // 8-element chunk of the row
  const float4 a = *reinterpret_cast<const float4 *>(X + r * K + c * 8), b = *reinterpret_cast<const float4 *>(X + r * K + c * 8 + 4);
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  uint32_t h[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    h[j] = cvt_pk_bf16_f32(v[2 * j], v[2 * j + 1]);
    const float r0 = v[2 * j] - __uint_as_float(h[j] << 16), r1 = v[2 * j + 1] - __uint_as_float(h[j] & 0xffff0000u);
    l[j] = cvt_pk_bf16_f32(r0, r1);
  }
  // k-block j = c / 4 (32 k), chunk c % 4 inside the block; hi at +0, lo at +64 bytes of the 128-byte line
  char *line = reinterpret_cast<char *>(Xs) + (size_t)r * K * 4 + (size_t)(c >> 2) * 128 + (c & 3) * 16;
  *reinterpret_cast<uint4 *>(line) = make_uint4(h[0], h[1], h[2], h[3]);
  *reinterpret_cast<uint4 *>(line + 64) = make_uint4(l[0], l[1], l[2], l[3]);
}

}  // namespace unopose

using namespace unopose;

extern "C" {

int unopose_split_bf16x2(const float *X, long M, int K, void *Xs, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(X && Xs, "split_bf16x2: null pointer");
  UNOPOSE_REQUIRE(M >= 1 && K >= 32 && K % 32 == 0, "split_bf16x2: needs K %% 32 == 0 (got M=%ld K=%d)", M, K);
  const long total8 = M * (K / 8);
  hipLaunchKernelGGL(split_bf16x2_kernel, dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, X, (u16 *)Xs, total8, K);
  return check_launch("split_bf16x2");
}

int unopose_linear_f32x3(const void *As, const void *Ws, const float *bias, float *C, void *Cs, long M, int N, int K, int epilogue,
                         unopose_stream_t stream) {
  UNOPOSE_REQUIRE(As && Ws && bias && (C || Cs), "linear_f32x3: null pointer");
  UNOPOSE_REQUIRE(M >= 1 && M < (1L << 31) && N >= GEMM_BN && N % GEMM_BN == 0 && K >= GEMMF_BK && K % GEMMF_BK == 0,
                  "linear_f32x3: needs N %% 256 == 0 and K %% 32 == 0 (got M=%ld N=%d K=%d)", M, N, K);
  UNOPOSE_REQUIRE((size_t)M * K * 4 < (1UL << 32) && (size_t)N * K * 4 < (1UL << 32) && (size_t)M * N * 4 < (1UL << 32),
                  "linear_f32x3: operand or output larger than 4 GiB (32-bit buffer offsets)");
  UNOPOSE_REQUIRE(epilogue >= 0 && epilogue <= 2, "linear_f32x3: epilogue must be 0 (bias), 1 (bias + GELU) or 2 (bias + ReLU)");
  const int tiles_m = cdiv(M, GEMM_BM), tiles_n = N / GEMM_BN, tiles = tiles_m * tiles_n;
  // too few 256 x 256 tiles for one per CU (the matcher's 197-token layers): 128 x 128 tiles, four times the workgroups (gemm_small.hip)
  if (tiles < gemm_small_tiles_limit()) return gemm_small_linear_f32(As, Ws, bias, C, Cs, M, N, K, epilogue, (hipStream_t)stream);
  const int n_cu = gemm_cu_count();
  const int grid = tiles >= n_cu ? n_cu : ((tiles + 7) & ~7);
  const int nt = (size_t)M * N * 4 > (32u << 20) ? 1 : 0;
  hipStream_t s = (hipStream_t)stream;
  int *const sched = tiles > grid ? gemm_sched_slot((hipStream_t)stream) : nullptr;
#define UNOPOSE_LAUNCH_F32X3(E)                                                                                                            \
  hipLaunchKernelGGL((gemm256_kernel<E, false, true>), dim3(grid), dim3(512), 0, s, As, Ws, bias, (void *)C, (int)M, N, K, tiles_n, tiles, nt,  \
                     (const int *)nullptr, (const int *)nullptr, (const u16 *)nullptr, (const float *)nullptr, (const float *)nullptr, 0.f, 0, \
                     0, 0, sched, Cs)
  if (epilogue == 1) UNOPOSE_LAUNCH_F32X3(1);
  else if (epilogue == 2) UNOPOSE_LAUNCH_F32X3(2);
  else UNOPOSE_LAUNCH_F32X3(0);
#undef UNOPOSE_LAUNCH_F32X3
  return check_launch("linear_f32x3");
}

int unopose_linear_f32x3_bf16(const void *As, const void *Ws, const float *bias, const void *resid, void *Cb, long M, int N, int K,
                              unopose_stream_t stream) {
  UNOPOSE_REQUIRE(As && Ws && bias && Cb, "linear_f32x3_bf16: null pointer");
  UNOPOSE_REQUIRE(M >= 1 && M < (1L << 31) && N >= GEMM_BN && N % GEMM_BN == 0 && K >= GEMMF_BK && K % GEMMF_BK == 0,
                  "linear_f32x3_bf16: needs N %% 256 == 0 and K %% 32 == 0 (got M=%ld N=%d K=%d)", M, N, K);
  UNOPOSE_REQUIRE((size_t)M * K * 4 < (1UL << 32) && (size_t)N * K * 4 < (1UL << 32) && (size_t)M * N * 2 < (1UL << 32),
                  "linear_f32x3_bf16: operand or output larger than 4 GiB (32-bit buffer offsets)");
  const int tiles_m = cdiv(M, GEMM_BM), tiles_n = N / GEMM_BN, tiles = tiles_m * tiles_n;
  const int n_cu = gemm_cu_count();
  const int grid = tiles >= n_cu ? n_cu : ((tiles + 7) & ~7);
  hipLaunchKernelGGL((gemm256_kernel<4, false, true>), dim3(grid), dim3(512), 0, (hipStream_t)stream, As, Ws, bias, Cb, (int)M, N, K, tiles_n,
                     tiles, 0, (const int *)nullptr, (const int *)nullptr, (const u16 *)resid, (const float *)nullptr, (const float *)nullptr,
                     0.f, 0, 0, 0, tiles > grid ? gemm_sched_slot((hipStream_t)stream) : nullptr, (void *)nullptr);
  return check_launch("linear_f32x3_bf16");
}

}  // extern "C"
