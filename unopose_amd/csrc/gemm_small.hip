// Small-tile form of the bf16 linear GEMM of gemm.hip (same C ABI entry points, chosen inside them by shape):
//     C[M][N] = act( A[M][K] . W[N][K]^T + bias[N] ),   A / W / C bf16 row-major, bias fp32, fp32 accumulation.
// User: the matcher's linears on the 197-token side (transformer.py:151-193: 6 304 or 12 608 rows, N in {256, 512}, K in {256,
// 512}) -- 25 to 200 tiles of 256 x 256 on 256 CUs; here they are 100 to 800 workgroups.
// Structure: 256 threads = 4 waves, one BM x BN output tile per workgroup, 64 x 64 outputs per wave
// (2 x 2 v_mfma_f32_32x32x16_bf16 blocks, operands swapped as in gemm.hip):
//     <128, 128>  waves 2 (M) x 2 (N)      <64, 256>  waves 1 x 4: a whole 256-wide row per tile, which the residual + LayerNorm
//                                                     epilogue (EPI 3) needs
// Operand stages are the LDS image of gemm.hip ([rows][128 B], 16-byte chunks XOR-swizzled on the SOURCE address of the
// LDS-DMA), NST stages of K = 64 (template parameter; the launches use 2: 64 KiB, two workgroups per CU at 128 x 128 -- or one
// beside a workgroup of another stream's kernel).
// K loop per K-tile:  issue the DMA pieces of K-tile t + NST - 1  |  counted vmcnt -> K-tile t has landed  |  barrier  |
// 16 fragment reads + 16 MFMAs  |  barrier.  Epilogue: bias / GELU / ReLU / residual + LayerNorm on the fp32 accumulators, bf16 staged
// through the (now free) stage buffers, whole 128-byte row segments out through a buffer descriptor (rows past M dropped).
#include "gemm_common.h"

namespace unopose {

#define GS_BK 64
#ifndef GS_PREFETCH
#define GS_PREFETCH 1
#endif

template <int BM, int BN, int EPI, int NST>
__global__ __launch_bounds__(256, (BM == 128 && NST == 2) ? 2 : 1) void gemm_small_kernel(const u16 *__restrict__ A, const u16 *__restrict__ W,
                                                                           const float *__restrict__ bias, u16 *__restrict__ C, int M, int N,
                                                                           int K, int tiles_m, int tiles_n, int lda, int ldw, int ldc,
                                                                           const u16 *__restrict__ resid, const float *__restrict__ ln_w,
                                                                           const float *__restrict__ ln_b, float ln_eps, u16 *__restrict__ vt = nullptr,
                                                                           int vt_m = 0, int vt_pad = 0) {
  // EPI 4 (round 6): bias only, and the LAST 256 columns -- the V half of the token attention's k | v projection (transformer.py:130-148,
  // 386-405) -- leave TRANSPOSED and key-padded, vt[cloud][channel][key] with `vt_pad` keys per row of which the first `vt_m` are tokens and the rest
  // zeros: the operand layout of unopose_token_attention, written by the projection itself instead of by a transpose launch per attention call.
  constexpr int WN = BN / 64, WM = 4 / WN;  // wave grid
  static_assert(WM * 64 == BM, "4 waves of 64 x 64");
  constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, STAGE = A_BYTES + W_BYTES;
  constexpr int PIECES = (BM + BN) / 8, PPW = PIECES / 4;  // 1-KiB DMA pieces per stage, per wave
  constexpr int A_PIECES = BM / 8;
  __shared__ __attribute__((aligned(1024))) char smem[NST * STAGE + (EPI == 3 ? 64 * 4 * 8 : 0)];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, hi = lane >> 5;
  // workgroup b sits on XCD b % 8 (speed assumption only): each XCD takes a contiguous range of the tile sequence, column tiles
  // fastest, so the tiles an XCD runs together share A row panels and the (small) weight matrix in its L2
  const int tiles = tiles_m * tiles_n;
  const int per_xcd = (tiles + 7) >> 3;
  const int t_id = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (t_id >= tiles) return;
  const int tm = t_id / tiles_n, tn = t_id - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc((void *)A, 0, (int)((size_t)M * lda * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc((void *)W, 0, (int)((size_t)N * ldw * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t c_rs = __builtin_amdgcn_make_buffer_rsrc((void *)C, 0, (int)((size_t)M * ldc * 2), 0x00020000);
  const int nk = K / GS_BK;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;

  // DMA piece j = wave + 4 i (i < PPW) of a stage: rows 8 j .. 8 j + 7 of the [A rows | W rows] sequence.  (row >> 1) & 7 of a
  // lane's row depends on j only through its parity = the wave's parity, so one per-lane base per operand serves all pieces;
  // the piece's row offset is added in the VGPR (A rows past M must fall outside the descriptor).
  const int r8 = lane >> 3;
  const int c_sw = (lane & 7) ^ (((wave & 1) << 2) | (r8 >> 1));
  const uint32_t a_lane = (uint32_t)(((size_t)(m0 + r8) * lda + c_sw * 8) * 2);
  const uint32_t w_lane = (uint32_t)(((size_t)(n0 + r8) * ldw + c_sw * 8) * 2);
  auto stage = [&](int kt, int buf) {
    const int so = kt * (GS_BK * 2);
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int j = wave + 4 * i;  // (wave-uniform; A_PIECES is a multiple of 4, so i decides the operand)
      if (4 * i < A_PIECES)
        gemm_dma16(lds0 + buf * STAGE + j * 1024, a_lane + (uint32_t)(j * 8) * (uint32_t)(lda * 2), a_rs, so);
      else
        gemm_dma16(lds0 + buf * STAGE + j * 1024, w_lane + (uint32_t)((j - A_PIECES) * 8) * (uint32_t)(ldw * 2), w_rs, so);
    }
  };
  // fragment reads: tile row r = base + l31 (base a multiple of 32), chunk c = 2 ks + hi
  const int fx = (l31 >> 1) & 7;
  uint32_t fr_off[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) fr_off[ks] = (uint32_t)((l31 >> 3) * 1024 + (l31 & 7) * 128 + ((((ks << 1) | hi) ^ fx) << 4));
  const uint32_t a_base = (uint32_t)(wm * 64 * 128), w_base = (uint32_t)(A_BYTES + wn * 64 * 128);

  // bias of the lane's columns (needed only in the epilogue; issued first so that it is the oldest load)
  float4 bv[2][4];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int g = 0; g < 4; ++g) bv[nb][g] = *reinterpret_cast<const float4 *>(bias + n0 + wn * 64 + nb * 32 + 8 * g + 4 * hi);
  // NST stages: K-tiles t + 1 .. t + NST - 1 are in flight while K-tile t is computed (NST = 2: two workgroups per CU cover each
  // other's waits; NST = 3 / 4: one workgroup per CU -- few tiles or a long K -- hides the L2 / HBM latency itself)
#pragma unroll
  for (int i = 0; i < NST - 1; ++i)
    if (i < nk) stage(i, i);
  f32x16 acc[2][2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nb][mb][r] = 0.f;
  int buf = 0, nbuf = NST - 1;  // stage of K-tile t / of K-tile t + NST - 1
  for (int t = 0; t < nk; ++t) {
    if (t + NST - 1 < nk) stage(t + NST - 1, nbuf);
    // loads still allowed in flight: the stages of K-tiles t + 1 .. min(t + NST - 1, nk - 1)
    const int ahead = min(NST - 1, nk - 1 - t);
    if (ahead >= 3)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PPW) : "memory");
    else if (ahead == 2)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
    else if (ahead == 1)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const char *lb = smem + buf * STAGE;
#if GS_PREFETCH
    // the fragments of k-step ks + 1 are read (into the other of two register sets) before the MFMAs of k-step ks are issued: left to
    // the compiler, one set was reused and every k-step began with its four reads
    bf16x8 wf[2][2], af[2][2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) wf[0][nb] = *reinterpret_cast<const bf16x8 *>(lb + w_base + nb * 4096 + fr_off[0]);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) af[0][mb] = *reinterpret_cast<const bf16x8 *>(lb + a_base + mb * 4096 + fr_off[0]);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks + 1 < 4) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) wf[(ks + 1) & 1][nb] = *reinterpret_cast<const bf16x8 *>(lb + w_base + nb * 4096 + fr_off[ks + 1]);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) af[(ks + 1) & 1][mb] = *reinterpret_cast<const bf16x8 *>(lb + a_base + mb * 4096 + fr_off[ks + 1]);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) acc[nb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks & 1][nb], af[ks & 1][mb], acc[nb][mb], 0, 0, 0);
    }
#else
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 wf[2], af[2];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) wf[nb] = *reinterpret_cast<const bf16x8 *>(lb + w_base + nb * 4096 + fr_off[ks]);
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) af[mb] = *reinterpret_cast<const bf16x8 *>(lb + a_base + mb * 4096 + fr_off[ks]);
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) acc[nb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nb], af[mb], acc[nb][mb], 0, 0, 0);
    }
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this K-tile's reads are done before anybody refills the buffer
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    buf = buf + 1 == NST ? 0 : buf + 1;
    nbuf = nbuf + 1 == NST ? 0 : nbuf + 1;
  }
  // ---- epilogue: acc[nb][mb][4g + e] = C[m = m0 + wm*64 + mb*32 + l31][n = n0 + wn*64 + nb*32 + 8g + 4hi + e]
  if (EPI == 3) {
    float2 *ln_part = reinterpret_cast<float2 *>(smem + NST * STAGE);  // [row 0..63][wn]
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      const int m = min(m0 + mb * 32 + l31, M - 1);
      const u16 *rp = resid + (size_t)m * BN + wn * 64 + 4 * hi;
      float a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const uint2 r = *reinterpret_cast<const uint2 *>(rp + nb * 32 + 8 * g);
          const float v0 = acc[nb][mb][4 * g + 0] + bv[nb][g].x + __uint_as_float(r.x << 16);
          const float v1 = acc[nb][mb][4 * g + 1] + bv[nb][g].y + __uint_as_float(r.x & 0xffff0000u);
          const float v2 = acc[nb][mb][4 * g + 2] + bv[nb][g].z + __uint_as_float(r.y << 16);
          const float v3 = acc[nb][mb][4 * g + 3] + bv[nb][g].w + __uint_as_float(r.y & 0xffff0000u);
          acc[nb][mb][4 * g + 0] = v0;
          acc[nb][mb][4 * g + 1] = v1;
          acc[nb][mb][4 * g + 2] = v2;
          acc[nb][mb][4 * g + 3] = v3;
          a1 += (v0 + v1) + (v2 + v3);
          a2 += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
        }
      a1 += __shfl_xor(a1, 32);  // lanes l31 and l31 + 32 hold complementary columns of the same row
      a2 += __shfl_xor(a2, 32);
      if (hi == 0) ln_part[(mb * 32 + l31) * 4 + wn] = make_float2(a1, a2);
    }
    __syncthreads();
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      const float2 *pp = ln_part + (mb * 32 + l31) * 4;
      const float t1 = (pp[0].x + pp[1].x) + (pp[2].x + pp[3].x), t2 = (pp[0].y + pp[1].y) + (pp[2].y + pp[3].y);
      const float mean = t1 * (1.f / BN);
      const float rstd = rsqrtf(fmaxf(t2 * (1.f / BN) - mean * mean, 0.f) + ln_eps);
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nl = wn * 64 + nb * 32 + 8 * g + 4 * hi;
          const float4 gw = *reinterpret_cast<const float4 *>(ln_w + nl), gb = *reinterpret_cast<const float4 *>(ln_b + nl);
          acc[nb][mb][4 * g + 0] = (acc[nb][mb][4 * g + 0] - mean) * rstd * gw.x + gb.x;
          acc[nb][mb][4 * g + 1] = (acc[nb][mb][4 * g + 1] - mean) * rstd * gw.y + gb.y;
          acc[nb][mb][4 * g + 2] = (acc[nb][mb][4 * g + 2] - mean) * rstd * gw.z + gb.z;
          acc[nb][mb][4 * g + 3] = (acc[nb][mb][4 * g + 3] - mean) * rstd * gw.w + gb.w;
        }
    }
  }
  // one pass of 64 rows per wave through 8 KiB of the stage buffers (16-byte slots XOR-swizzled by row)
  char *cw = smem + wave * 8192;
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int nl = nb * 32 + 8 * g + 4 * hi;
        float v0 = acc[nb][mb][4 * g + 0], v1 = acc[nb][mb][4 * g + 1], v2 = acc[nb][mb][4 * g + 2], v3 = acc[nb][mb][4 * g + 3];
        if (EPI != 3) {
          v0 += bv[nb][g].x;
          v1 += bv[nb][g].y;
          v2 += bv[nb][g].z;
          v3 += bv[nb][g].w;
        }
        if (EPI == 1) {
          v0 = gelu_bf16_class(v0);
          v1 = gelu_bf16_class(v1);
          v2 = gelu_bf16_class(v2);
          v3 = gelu_bf16_class(v3);
        }
        if (EPI == 2) {
          v0 = fmaxf(v0, 0.f);
          v1 = fmaxf(v1, 0.f);
          v2 = fmaxf(v2, 0.f);
          v3 = fmaxf(v3, 0.f);
        }
        const int row = mb * 32 + l31;
        const int slot16 = (nl >> 3) ^ (row & 7);
        *reinterpret_cast<uint2 *>(cw + row * 128 + slot16 * 16 + (nl & 4) * 2) = make_uint2(cvt_pk_bf16_f32(v0, v1), cvt_pk_bf16_f32(v2, v3));
      }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (EPI == 4 && n0 + wn * 64 >= N - 256) {
    // this wave's 64 x 64 block belongs to V: lane = one token row (its 128 staged bytes come back as 8 x 16 B), then one 2-byte store per
    // channel -- consecutive lanes are consecutive keys of a channel row of vt: whole contiguous runs, split only where a cloud ends
    const int grow = m0 + wm * 64 + lane;
    const int cb = grow / vt_m, key = grow - cb * vt_m;
    const int ch0 = n0 + wn * 64 - (N - 256);
    u16 *dst = vt + ((size_t)cb * 256 + ch0) * vt_pad + key;
    u32x4 rowv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) rowv[q] = *reinterpret_cast<const u32x4 *>(cw + lane * 128 + ((q ^ (lane & 7)) << 4));
    if (grow < M) {
#pragma unroll
      for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          dst[(size_t)(q * 8 + 2 * e) * vt_pad] = (u16)(rowv[q][e] & 0xFFFFu);
          dst[(size_t)(q * 8 + 2 * e + 1) * vt_pad] = (u16)(rowv[q][e] >> 16);
        }
    }
    // the padding keys vt_m .. vt_pad - 1 of every cloud that ENDS inside this wave's rows: lanes 0 .. vt_pad - vt_m - 1 write the zeros
    const int g0 = m0 + wm * 64, g1 = min(g0 + 64, M);  // rows [g0, g1)
    for (int cbz = g0 / vt_m; cbz * vt_m + vt_m <= g1; ++cbz) {  // (wave-uniform: clouds whose last token lies in [g0, g1))
      if (cbz * vt_m + vt_m - 1 < g0) continue;
      if (lane < vt_pad - vt_m) {
        u16 *z = vt + ((size_t)cbz * 256 + ch0) * vt_pad + vt_m + lane;
#pragma unroll 8
        for (int c = 0; c < 64; ++c) z[(size_t)c * vt_pad] = 0;
      }
    }
    return;
  }
  const uint32_t c_v0 = (uint32_t)((((size_t)m0 + wm * 64 + (lane >> 3)) * ldc + n0 + wn * 64 + (lane & 7) * 8) * 2);
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int row = it * 8 + (lane >> 3), q = lane & 7;
    const u32x4 v = *reinterpret_cast<const u32x4 *>(cw + row * 128 + ((q ^ (row & 7)) << 4));
    __builtin_amdgcn_raw_buffer_store_b128(v, c_rs, c_v0 + (uint32_t)(it * 8) * (uint32_t)(ldc * 2), 0, 0);
  }
}

template <int BM, int BN, int EPI, int NST>
static int launch_small(const void *A, const void *W, const float *bias, void *C, long M, int N, int K, int lda, int ldw, int ldc,
                        const void *resid, const float *ln_w, const float *ln_b, float eps, hipStream_t s) {
  const int tiles_m = cdiv(M, BM), tiles_n = N / BN;
  const int tiles = tiles_m * tiles_n;
  const int grid = ((tiles + 7) >> 3) << 3;
  hipLaunchKernelGGL((gemm_small_kernel<BM, BN, EPI, NST>), dim3(grid), dim3(256), 0, s, (const u16 *)A, (const u16 *)W, bias, (u16 *)C, (int)M, N,
                     K, tiles_m, tiles_n, lda, ldw, ldc, (const u16 *)resid, ln_w, ln_b, eps);
  return check_launch("linear_bf16 (small tiles)");
}

// Entry points for gemm.hip's C ABI functions (N % 128 == 0, K % 64 == 0; epilogue 0 / 1 / 2).  Two stages (64 / 82 KiB of LDS): a
// workgroup then fits on a CU BESIDE a workgroup of the positional-encoding kernel (76 KiB), which runs on a side stream under the
// coarse stage -- with 3 - 4 stages (measured no faster per launch: 5.8 vs 5.5 us at 6304 x 256 x 256) every coarse-stage linear
// waited for a whole CU, up to 1.3 ms behind the PE launch (profiles/r04_kernel_stats_isolated_b32_s518.csv, first collection).
int gemm_small_linear(const void *A, const void *W, const float *bias, void *C, long M, int N, int K, int lda, int ldw, int ldc, int epilogue,
                      hipStream_t s) {
  if (epilogue == 1) return launch_small<128, 128, 1, 2>(A, W, bias, C, M, N, K, lda, ldw, ldc, nullptr, nullptr, nullptr, 0.f, s);
  if (epilogue == 2) return launch_small<128, 128, 2, 2>(A, W, bias, C, M, N, K, lda, ldw, ldc, nullptr, nullptr, nullptr, 0.f, s);
  return launch_small<128, 128, 0, 2>(A, W, bias, C, M, N, K, lda, ldw, ldc, nullptr, nullptr, nullptr, 0.f, s);
}
// N == 256: LayerNorm(A W^T + bias + resid) * ln_w + ln_b, 64-row tiles
// bias only; the last 256 columns written transposed + key-padded into vt (EPI 4 above); C keeps the other columns (its last 256 are not written)
int gemm_small_linear_vt(const void *A, const void *W, const float *bias, void *C, void *vt, long M, int N, int K, int tokens, int key_pad, hipStream_t s) {
  const int tiles_m = cdiv(M, 128), tiles_n = N / 128;
  const int tiles = tiles_m * tiles_n;
  const int grid = ((tiles + 7) >> 3) << 3;
  hipLaunchKernelGGL((gemm_small_kernel<128, 128, 4, 2>), dim3(grid), dim3(256), 0, s, (const u16 *)A, (const u16 *)W, bias, (u16 *)C, (int)M, N, K, tiles_m,
                     tiles_n, K, K, N, (const u16 *)nullptr, (const float *)nullptr, (const float *)nullptr, 0.f, (u16 *)vt, tokens, key_pad);
  return check_launch("linear_bf16_kv_vt");
}
int gemm_small_linear_ln(const void *A, const void *W, const float *bias, const void *resid, const float *ln_w, const float *ln_b, float eps,
                         void *C, long M, int K, hipStream_t s) {
  return launch_small<64, 256, 3, 2>(A, W, bias, C, M, 256, K, K, K, 256, resid, ln_w, ln_b, eps, s);
}

// ---- fp32-class form (gemm_f32.hip's split operands: 128-byte rows hold 32 k as [hi (32 bf16) | lo (32 bf16)]): 128 x 128 tiles, 4 waves of
// 64 x 64, two stages of K = 32, 24 MFMAs per K-tile and wave (hi.hi + hi.lo + lo.hi per fragment pair).  Users: the fp32 path's matcher
// linears (6 304 - 12 608 rows x 256 / 512: 25 - 100 tiles of 256 x 256 -- 96 launches, 5.4 ms per forward on the 256-tile kernel).
// Epilogue: bias / exact-erf GELU / ReLU, the 64 x 64 fp32 block of a wave staged through 16 KiB of the (free) stage buffers (16-byte
// slots XOR-swizzled by row), then whole 256-byte row segments as fp32 (C) and / or in the split layout (Cs).
template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_small_f32_kernel(const char *__restrict__ A, const char *__restrict__ W, const float *__restrict__ bias,
                                                                float *__restrict__ C, char *__restrict__ Cs, int M, int N, int K, int tiles_m, int tiles_n) {
  constexpr int BM = 128, BN = 128, NST = 2;
  constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;
  constexpr int PPW = (BM + BN) / 8 / 4, A_PIECES = BM / 8;
  __shared__ __attribute__((aligned(1024))) char smem[NST * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hi = lane >> 5;
  const int tiles = tiles_m * tiles_n;
  const int per_xcd = (tiles + 7) >> 3;
  const int t_id = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (t_id >= tiles) return;
  const int tm = t_id / tiles_n, tn = t_id - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const size_t rowb = (size_t)K * 4;  // bytes per operand row
  const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc((void *)A, 0, (int)((size_t)M * rowb), 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc((void *)W, 0, (int)((size_t)N * rowb), 0x00020000);
  const int nk = K / 32;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
  const int r8 = lane >> 3;
  const int c_sw = (lane & 7) ^ (((wave & 1) << 2) | (r8 >> 1));
  const uint32_t a_lane = (uint32_t)((size_t)(m0 + r8) * rowb + c_sw * 16);
  const uint32_t w_lane = (uint32_t)((size_t)(n0 + r8) * rowb + c_sw * 16);
  auto stage = [&](int kt, int buf) {
    const int so = kt * 128;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int j = wave + 4 * i;
      if (4 * i < A_PIECES)
        gemm_dma16(lds0 + buf * STAGE + j * 1024, a_lane + (uint32_t)(j * 8) * (uint32_t)rowb, a_rs, so);
      else
        gemm_dma16(lds0 + buf * STAGE + j * 1024, w_lane + (uint32_t)((j - A_PIECES) * 8) * (uint32_t)rowb, w_rs, so);
    }
  };
  // fragment reads: tile row r = base + l31, 16-byte chunk c: hi part of k-step ks = 2 ks + hi, lo part = 4 + 2 ks + hi
  const int fx = (l31 >> 1) & 7;
  uint32_t fr_off[4];
#pragma unroll
  for (int c2 = 0; c2 < 4; ++c2) fr_off[c2] = (uint32_t)((l31 >> 3) * 1024 + (l31 & 7) * 128 + ((((c2 << 1) | hi) ^ fx) << 4));
  const uint32_t a_base = (uint32_t)(wm * 64 * 128), w_base = (uint32_t)(A_BYTES + wn * 64 * 128);
  float4 bv[2][4];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int g = 0; g < 4; ++g) bv[nb][g] = *reinterpret_cast<const float4 *>(bias + n0 + wn * 64 + nb * 32 + 8 * g + 4 * hi);
  stage(0, 0);
  f32x16 acc[2][2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nb][mb][r] = 0.f;
  for (int t = 0; t < nk; ++t) {
    const int buf = t & 1;
    if (t + 1 < nk) {
      stage(t + 1, buf ^ 1);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const char *lb = smem + buf * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 wh[2], wl[2], ah[2], al[2];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        wh[nb] = *reinterpret_cast<const bf16x8 *>(lb + w_base + nb * 4096 + fr_off[ks]);
        wl[nb] = *reinterpret_cast<const bf16x8 *>(lb + w_base + nb * 4096 + fr_off[2 + ks]);
      }
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        ah[mb] = *reinterpret_cast<const bf16x8 *>(lb + a_base + mb * 4096 + fr_off[ks]);
        al[mb] = *reinterpret_cast<const bf16x8 *>(lb + a_base + mb * 4096 + fr_off[2 + ks]);
      }
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {  // (the order of gemm_kernel.h's F32 form: small terms first)
          acc[nb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[nb], ah[mb], acc[nb][mb], 0, 0, 0);
          acc[nb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[nb], al[mb], acc[nb][mb], 0, 0, 0);
          acc[nb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[nb], ah[mb], acc[nb][mb], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  // ---- epilogue: acc[nb][mb][4g + e] = C[m = m0 + wm*64 + mb*32 + l31][n = n0 + wn*64 + nb*32 + 8g + 4hi + e]
  char *cw = smem + wave * 16384;  // [64 rows][256 B], 16-byte slot c of row r at c ^ (r & 15)
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v0 = acc[nb][mb][4 * g + 0] + bv[nb][g].x, v1 = acc[nb][mb][4 * g + 1] + bv[nb][g].y;
        float v2 = acc[nb][mb][4 * g + 2] + bv[nb][g].z, v3 = acc[nb][mb][4 * g + 3] + bv[nb][g].w;
        if (EPI == 1) {
          v0 = gelu_erf(v0);
          v1 = gelu_erf(v1);
          v2 = gelu_erf(v2);
          v3 = gelu_erf(v3);
        }
        if (EPI == 2) {
          v0 = fmaxf(v0, 0.f);
          v1 = fmaxf(v1, 0.f);
          v2 = fmaxf(v2, 0.f);
          v3 = fmaxf(v3, 0.f);
        }
        const int row = mb * 32 + l31, slot = nb * 8 + 2 * g + hi;
        *reinterpret_cast<float4 *>(cw + row * 256 + ((slot ^ (row & 15)) << 4)) = make_float4(v0, v1, v2, v3);
      }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int row = it * 4 + (lane >> 4), q = lane & 15;
    const int m = m0 + wm * 64 + row;
    const float4 v = *reinterpret_cast<const float4 *>(cw + row * 256 + ((q ^ (row & 15)) << 4));
    if (m < M) {
      const int n = n0 + wn * 64 + q * 4;
      if (C) *reinterpret_cast<float4 *>(C + (size_t)m * N + n) = v;
      if (Cs) {
        uint2 h, l;
        h.x = cvt_pk_bf16_f32(v.x, v.y);
        h.y = cvt_pk_bf16_f32(v.z, v.w);
        l.x = cvt_pk_bf16_f32(v.x - __uint_as_float(h.x << 16), v.y - __uint_as_float(h.x & 0xffff0000u));
        l.y = cvt_pk_bf16_f32(v.z - __uint_as_float(h.y << 16), v.w - __uint_as_float(h.y & 0xffff0000u));
        char *line = Cs + (size_t)m * N * 4 + (size_t)(n >> 5) * 128 + (n & 31) * 2;
        *reinterpret_cast<uint2 *>(line) = h;
        *reinterpret_cast<uint2 *>(line + 64) = l;
      }
    }
  }
}

// Entry point for gemm_f32.hip's unopose_linear_f32x3 (N % 128 == 0, K % 32 == 0; epilogue 0 / 1 / 2; C and / or Cs).
int gemm_small_linear_f32(const void *As, const void *Ws, const float *bias, float *C, void *Cs, long M, int N, int K, int epilogue, hipStream_t s) {
  const int tiles_m = cdiv(M, 128), tiles_n = N / 128, tiles = tiles_m * tiles_n;
  const int grid = ((tiles + 7) >> 3) << 3;
#define UNOPOSE_LAUNCH_SMALL_F32(E)                                                                                                                   \
  hipLaunchKernelGGL((gemm_small_f32_kernel<E>), dim3(grid), dim3(256), 0, s, (const char *)As, (const char *)Ws, bias, C, (char *)Cs, (int)M, N, K, \
                     tiles_m, tiles_n)
  if (epilogue == 1) UNOPOSE_LAUNCH_SMALL_F32(1);
  else if (epilogue == 2) UNOPOSE_LAUNCH_SMALL_F32(2);
  else UNOPOSE_LAUNCH_SMALL_F32(0);
#undef UNOPOSE_LAUNCH_SMALL_F32
  return check_launch("linear_f32x3 (small tiles)");
}

}  // namespace unopose
