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
// The kernel is the bf16 kernel's structure (gemm.hip: 256 x 256 tile, 8 waves as 2 x 4, LDS-DMA of 1-KiB pieces into a
// source-swizzled [256][128 B] image, two LDS buffers, burst DMA issue, persistent lock-stepped XCD-aware tile walk) with a
// stage of 32 k instead of 64: per stage and wave 2 k-substeps x 8 output blocks x 3 MFMAs = 48 v_mfma_f32_32x32x16_bf16
// against the same 64 KiB of LDS-DMA and the same 24 fragment reads -- the loop is matrix-pipe-bound where the bf16 loop
// is issue-bound.  One fragment set (48 VGPRs): the partner wave of the SIMD covers the read latency with its own 24 MFMAs.
#include "gemm_common.h"

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

// EPI 0: bias; 1: bias + exact-erf GELU; 2: bias + ReLU.  C (fp32, row-major) and / or Cs (split layout) are written.
// EPI 3: bias, result rounded to bf16 and added to the bf16 residual `Cs` (may be NULL), written to `C` as bf16 (M,N).
template <int EPI>
__global__ __launch_bounds__(512, 1) void gemm_f32x3_kernel(const char *__restrict__ A, const char *__restrict__ W,
                                                            const float *__restrict__ bias, float *__restrict__ C, char *__restrict__ Cs,
                                                            int M, int N, int K, int tiles_n, int tiles, int nt_store) {
  __shared__ __attribute__((aligned(1024))) char smem[2 * GEMM_BUFBYTES];
  __shared__ __attribute__((aligned(16))) float bias_lds[GEMM_BN];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int l31 = lane & 31, hi = lane >> 5;
  // persistent, lock-stepped, XCD-aware tile walk: see gemm.hip
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int cq = tiles >> 3, cr = tiles & 7;
  const int chunk_base = xcd < cr ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq, chunk_len = cq + (xcd < cr ? 1 : 0);
  const int tiles_m = tiles / tiles_n, per_group = GEMM_GM * tiles_n;
  const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc((void *)A, 0, (int)((size_t)M * K * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc((void *)W, 0, (int)((size_t)N * K * 4), 0x00020000);
  const int nk = K / GEMMF_BK;

  // fragment read addresses: tile row r = base + l31, hi part chunk c = 2 ks + hi, lo part chunk 4 + 2 ks + hi
  const int fx = (l31 >> 1) & 7;
  uint32_t fr_h[2], fr_l[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    fr_h[ks] = (uint32_t)((l31 >> 3) * 1024 + (l31 & 7) * 128 + ((((ks << 1) | hi) ^ fx) << 4));
    fr_l[ks] = (uint32_t)((l31 >> 3) * 1024 + (l31 & 7) * 128 + (((4 | (ks << 1) | hi) ^ fx) << 4));
  }
  const uint32_t a_base = (uint32_t)(wm * 128 * 128);
  const uint32_t w_base = (uint32_t)(GEMM_OPBYTES + wn * 64 * 128);

  struct TileP {
    int m0, n0, rot;
    uint32_t a_off[4], w_off[4];
  };
  auto tile_params = [&](int ti, int step, TileP &p) {
    const int t = chunk_base + ti;
    const int mg = t / per_group, rr = t - mg * per_group;
    const int gm = min(GEMM_GM, tiles_m - mg * GEMM_GM);
    const int tn = rr / gm, tm = mg * GEMM_GM + (rr - tn * gm);
    p.m0 = __builtin_amdgcn_readfirstlane(tm * GEMM_BM);
    p.n0 = __builtin_amdgcn_readfirstlane(tn * GEMM_BN);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wave * 32 + i * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((row >> 1) & 7);
      p.a_off[i] = (uint32_t)((size_t)(p.m0 + row) * K * 4 + c * 16);
      p.w_off[i] = (uint32_t)((size_t)(p.n0 + row) * K * 4 + c * 16);
    }
    const int skew = ((tm & 3) + tn) % (GEMM_SKEW > 1 ? GEMM_SKEW : 1);
    p.rot = __builtin_amdgcn_readfirstlane((xcd * 5 + step * 3 + skew) % nk);
  };
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
  auto stage_tile = [&](const TileP &p, int buf, int kt) {
    kt += p.rot;
    if (kt >= nk) kt -= nk;
    const uint32_t la = lds0 + (uint32_t)(buf * GEMM_BUFBYTES + wave * 4096);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      gemm_dma16(la + i * 1024, p.a_off[i], a_rs, kt * GEMM_ROWB);
      gemm_dma16(la + i * 1024 + GEMM_OPBYTES, p.w_off[i], w_rs, kt * GEMM_ROWB);
    }
  };
  const bool can_prefetch = (nk & 1) == 0;
  TileP cur;
  float4 cur_bv;
  bool have = false;
  for (int ti = slot, step = 0; ti < chunk_len; ti += nslots, ++step) {
    if (!have) {
      tile_params(ti, step, cur);
      cur_bv = *reinterpret_cast<const float4 *>(bias + cur.n0 + lane * 4);
      stage_tile(cur, 0, 0);
    }
    const int m0 = __builtin_amdgcn_readfirstlane(cur.m0), n0 = __builtin_amdgcn_readfirstlane(cur.n0);
    cur.rot = __builtin_amdgcn_readfirstlane(cur.rot);

    f32x16 acc[2][4];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][mb][r] = 0.f;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wave == 0) *reinterpret_cast<float4 *>(bias_lds + lane * 4) = cur_bv;
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int buf = kt & 1;
      const char *lb = smem + buf * GEMM_BUFBYTES;
      if (kt + 1 < nk) stage_tile(cur, buf ^ 1, kt + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 wh[2], wl[2], ah[4], al[4];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          wh[nb] = *reinterpret_cast<const bf16x8 *>(lb + w_base + nb * 4096 + fr_h[ks]);
          wl[nb] = *reinterpret_cast<const bf16x8 *>(lb + w_base + nb * 4096 + fr_l[ks]);
        }
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
          ah[mb] = *reinterpret_cast<const bf16x8 *>(lb + a_base + mb * 4096 + fr_h[ks]);
          al[mb] = *reinterpret_cast<const bf16x8 *>(lb + a_base + mb * 4096 + fr_l[ks]);
        }
        // small terms first: the accumulator then sees (ah wl + al wh) + ah wh of this k-substep in that order
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int mb = 0; mb < 4; ++mb) {
            acc[nb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[nb], ah[mb], acc[nb][mb], 0, 0, 0);
            acc[nb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[nb], al[mb], acc[nb][mb], 0, 0, 0);
            acc[nb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[nb], ah[mb], acc[nb][mb], 0, 0, 0);
          }
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }

    const bool more = can_prefetch && ti + nslots < chunk_len;
    TileP nxt;
    float4 nxt_bv;
    if (more) {
      tile_params(ti + nslots, step + 1, nxt);
      nxt_bv = *reinterpret_cast<const float4 *>(bias + nxt.n0 + lane * 4);
      stage_tile(nxt, 0, 0);
    }
    // ---- epilogue: acc[nb][mb][4g + e] = C[m = wm*128 + mb*32 + l31][n = wn*64 + nb*32 + 8g + 4hi + e]
    //      four passes (one per mb) of 32 rows x 64 columns per wave through buffer 1 (8 KiB per wave)
    char *cw = smem + GEMM_BUFBYTES + wave * 8192;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      float v[2][4][4];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 bv = *reinterpret_cast<const float4 *>(bias_lds + wn * 64 + nb * 32 + 8 * g + 4 * hi);
          const float b4[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float x = acc[nb][mb][4 * g + e] + b4[e];
            if (EPI == 1) x = gelu_erf(x);
            if (EPI == 2) x = fmaxf(x, 0.f);
            v[nb][g][e] = x;
          }
        }
      const int m_base = m0 + wm * 128 + mb * 32;
      if (EPI == 3) {
        // bf16 image: row = 128 B = 8 slots of 16 B (8 columns), slot nb * 4 + g, XOR-swizzled by the row
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<uint2 *>(cw + l31 * 128 + (((nb * 4 + g) ^ (l31 & 7)) << 4) + hi * 8) =
                make_uint2(cvt_pk_bf16_f32(v[nb][g][0], v[nb][g][1]), cvt_pk_bf16_f32(v[nb][g][2], v[nb][g][3]));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int row = it * 8 + (lane >> 3), q = lane & 7;
          uint4 o = *reinterpret_cast<const uint4 *>(cw + row * 128 + ((q ^ (row & 7)) << 4));
          const int m = m_base + row;
          if (m < M) {
            const size_t off = ((size_t)m * N + n0 + wn * 64 + q * 8) * 2;
            if (Cs) {
              const uint4 r = *reinterpret_cast<const uint4 *>(Cs + off);
              const uint32_t ov[4] = {o.x, o.y, o.z, o.w}, rv[4] = {r.x, r.y, r.z, r.w};
              uint32_t sv[4];
#pragma unroll
              for (int e = 0; e < 4; ++e)
                sv[e] = cvt_pk_bf16_f32(__uint_as_float(ov[e] << 16) + __uint_as_float(rv[e] << 16),
                                        __uint_as_float(ov[e] & 0xffff0000u) + __uint_as_float(rv[e] & 0xffff0000u));
              o = make_uint4(sv[0], sv[1], sv[2], sv[3]);
            }
            *reinterpret_cast<uint4 *>(reinterpret_cast<char *>(C) + off) = o;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        continue;
      }
      if (C) {
        // fp32 image: row = 256 B = 16 slots of 16 B, slot s = nb * 8 + 2 g + hi, XOR-swizzled by the row
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int s = (nb * 8 + 2 * g + hi) ^ (l31 & 15);
            *reinterpret_cast<float4 *>(cw + l31 * 256 + s * 16) = make_float4(v[nb][g][0], v[nb][g][1], v[nb][g][2], v[nb][g][3]);
          }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int row = it * 4 + (lane >> 4), q = lane & 15;
          const uint4 o = *reinterpret_cast<const uint4 *>(cw + row * 256 + ((q ^ (row & 15)) << 4));
          const int m = m_base + row;
          if (m < M) {
            char *dst = reinterpret_cast<char *>(C) + ((size_t)m * N + n0 + wn * 64 + q * 4) * 4;
            if (nt_store) {
              const u32x4 vv = {o.x, o.y, o.z, o.w};
              __builtin_nontemporal_store(vv, reinterpret_cast<u32x4 *>(dst));
            } else {
              *reinterpret_cast<uint4 *>(dst) = o;
            }
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      if (Cs) {
        // split image of the wave's 64 columns = 2 k-blocks of 128 B: block nb, element j = 8 g + 4 hi + e: hi part at byte
        // nb * 128 + 2 j, lo part 64 bytes further; 16-byte slots XOR-swizzled by the row
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const uint32_t h0 = cvt_pk_bf16_f32(v[nb][g][0], v[nb][g][1]), h1 = cvt_pk_bf16_f32(v[nb][g][2], v[nb][g][3]);
            const float r0 = v[nb][g][0] - __uint_as_float(h0 << 16), r1 = v[nb][g][1] - __uint_as_float(h0 & 0xffff0000u);
            const float r2 = v[nb][g][2] - __uint_as_float(h1 << 16), r3 = v[nb][g][3] - __uint_as_float(h1 & 0xffff0000u);
            const uint32_t l0 = cvt_pk_bf16_f32(r0, r1), l1 = cvt_pk_bf16_f32(r2, r3);
            const int sh = (nb * 8 + g) ^ (l31 & 15), sl = (nb * 8 + 4 + g) ^ (l31 & 15);  // slot of the hi / lo 16-byte chunk (8 elements)
            *reinterpret_cast<uint2 *>(cw + l31 * 256 + sh * 16 + hi * 8) = make_uint2(h0, h1);
            *reinterpret_cast<uint2 *>(cw + l31 * 256 + sl * 16 + hi * 8) = make_uint2(l0, l1);
          }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int row = it * 4 + (lane >> 4), q = lane & 15;
          const uint4 o = *reinterpret_cast<const uint4 *>(cw + row * 256 + ((q ^ (row & 15)) << 4));
          const int m = m_base + row;
          // the row's split bytes of columns n0 + wn * 64 .. + 63 are contiguous: (n / 32) * 128 = (n0 + wn * 64) * 4
          if (m < M) *reinterpret_cast<uint4 *>(Cs + (size_t)m * N * 4 + (size_t)(n0 + wn * 64) * 4 + q * 16) = o;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
    __syncthreads();
    have = more;
    if (more) {
      cur = nxt;
      cur_bv = nxt_bv;
    }
  }
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
  UNOPOSE_REQUIRE((size_t)M * K * 4 < (1UL << 32) && (size_t)N * K * 4 < (1UL << 32), "linear_f32x3: operand larger than 4 GiB");
  UNOPOSE_REQUIRE(epilogue >= 0 && epilogue <= 2, "linear_f32x3: epilogue must be 0 (bias), 1 (bias + GELU) or 2 (bias + ReLU)");
  const int tiles_m = cdiv(M, GEMM_BM), tiles_n = N / GEMM_BN, tiles = tiles_m * tiles_n;
  const int n_cu = gemm_cu_count();
  const int grid = tiles >= n_cu ? n_cu : ((tiles + 7) & ~7);
  const int nt = (size_t)M * N * 4 > (32u << 20) ? 1 : 0;
  hipStream_t s = (hipStream_t)stream;
#define UNOPOSE_LAUNCH_F32X3(E)                                                                                               \
  hipLaunchKernelGGL(gemm_f32x3_kernel<E>, dim3(grid), dim3(512), 0, s, (const char *)As, (const char *)Ws, bias, C, (char *)Cs, (int)M, N, \
                     K, tiles_n, tiles, nt)
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
  UNOPOSE_REQUIRE((size_t)M * K * 4 < (1UL << 32) && (size_t)N * K * 4 < (1UL << 32), "linear_f32x3_bf16: operand larger than 4 GiB");
  const int tiles_m = cdiv(M, GEMM_BM), tiles_n = N / GEMM_BN, tiles = tiles_m * tiles_n;
  const int n_cu = gemm_cu_count();
  const int grid = tiles >= n_cu ? n_cu : ((tiles + 7) & ~7);
  hipLaunchKernelGGL(gemm_f32x3_kernel<3>, dim3(grid), dim3(512), 0, (hipStream_t)stream, (const char *)As, (const char *)Ws, bias,
                     (float *)Cb, (char *)const_cast<void *>(resid), (int)M, N, K, tiles_n, tiles, 0);
  return check_launch("linear_f32x3_bf16");
}

}  // extern "C"
