// Geometric structure embedding for gfx950 (C ABI part 2).
//
// Replaces GeometricStructureEmbedding.forward (core/unopose/model/transformer.py:303-350):
//   E[b,i,j,:] = proj_d(sinus(|p_i-p_j|/sigma_d)) + max_k proj_a(sinus(angle(p_knn(i,k)-p_i, p_j-p_i)*factor_a))
// The reference materialises the (B,n,n,k,256) sinusoidal tensor and runs two Linear layers over it
// (20.35 GFLOP and ~0.5 GB of temporaries per cloud).  Here one kernel generates the sinusoids of a tile
// of 32 (i,j) pairs straight into LDS in MFMA A-operand layout (never touching HBM), contracts them with
// the two 256x256 weight matrices on the matrix cores (v_mfma_f32_32x32x16_bf16) and applies the
// max-over-k / bias epilogue on the accumulators, so HBM sees only the (B,n,n,256) result once.
//
// Tiling (wave64, 8 waves / workgroup): M-tile = 32 pairs x 4 "sets" (distance, angle k=0..2), each set
// is one 32x256 A matrix; wave w owns output channels [32w, 32w+32) for all four sets, so max over k is
// an element-wise max of three accumulator registers (C/D layout is identical across the sets).
// SPLIT=true keeps fp32-class accuracy on the bf16 matrix cores by splitting both operands into
// hi + lo bf16 parts (3 MFMAs per product, ~2^-16 relative error); SPLIT=false is the autocast(bf16) path.
#include "common.h"
#include <type_traits>

namespace unopose {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

__device__ __forceinline__ u16 f2bf(float f) {  // round-to-nearest-even
  uint32_t u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (u16)(u >> 16);
}
__device__ __forceinline__ float bf2f(u16 h) { return __uint_as_float(((uint32_t)h) << 16); }
// gfx950 packed fp32 -> bf16 conversion (RNE): low half = cvt(a), high half = cvt(b)
__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
  return cvt_pk_bf16_f32(a, b);
}

// sin and cos of |x| <~ 1e3 with ~1e-7 absolute error: 3-term Cody-Waite reduction by pi/2 + cephes
// single-precision minimax polynomials on [-pi/4, pi/4].
__device__ __forceinline__ void sincos_cw(float x, float &s, float &c) {
  const float k = rintf(x * 0.636619772367581343f);
  float r = fmaf(-k, 1.5707963705062866f, x);  // pi/2 = hi + mid + lo (each a float)
  r = fmaf(-k, -4.371138828673793e-08f, r);
  r = fmaf(-k, -1.7151245100058819e-15f, r);
  const float z = r * r;
  const float sp = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
  const float cp =
      fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f) * z, z,
           fmaf(-0.5f, z, 1.0f));
  const int q = ((int)k) & 3;
  const float s0 = (q & 1) ? cp : sp;
  const float c0 = (q & 1) ? sp : cp;
  s = (q & 2) ? -s0 : s0;
  c = ((q + 1) & 2) ? -c0 : c0;
}

// |x| <= ~0.35: Taylor polynomials, absolute error < 1e-8
__device__ __forceinline__ void sincos_small(float x, float &s, float &c) {
  const float z = x * x;
  s = fmaf(fmaf(fmaf(-1.9841270e-4f, z, 8.3333333e-3f), z, -1.6666667e-1f) * z, x, x);
  c = fmaf(fmaf(fmaf(-1.3888889e-3f, z, 4.1666667e-2f), z, -0.5f), z, 1.0f);
}

// k nearest neighbours (self excluded) of every point of a small cloud: knn (B,n,3).
__global__ __launch_bounds__(256) void geo_knn_kernel(const float *__restrict__ pts, int n,
                                                      int32_t *__restrict__ knn) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float *P = pts + (size_t)b * n * 3;
  const float x = P[i * 3], y = P[i * 3 + 1], z = P[i * 3 + 2];
  float d0 = 3e38f, d1 = 3e38f, d2 = 3e38f;
  int i0 = 0, i1 = 0, i2 = 0;
  for (int j = 0; j < n; ++j) {
    if (j == i) continue;
    const float dx = P[j * 3] - x, dy = P[j * 3 + 1] - y, dz = P[j * 3 + 2] - z;
    const float d = dx * dx + dy * dy + dz * dz;
    if (d < d0) { d2 = d1; i2 = i1; d1 = d0; i1 = i0; d0 = d; i0 = j; }
    else if (d < d1) { d2 = d1; i2 = i1; d1 = d; i1 = j; }
    else if (d < d2) { d2 = d; i2 = j; }
  }
  int32_t *K = knn + ((size_t)b * n + i) * 3;
  K[0] = i0; K[1] = i1; K[2] = i2;
}

constexpr int GE_PAIRS = 32;   // pairs per workgroup tile
constexpr int GE_DIM = 256;    // hidden_dim
constexpr int GE_THREADS = 512;

__device__ __forceinline__ int ge_swz(int row, int kbyte) {
  // 512-byte rows; XOR the 16-byte slot index with (row & 15): conflict-free ds_read_b128 for the
  // 32x32x16 A-operand access pattern (16 distinct rows per LDS lane group)
  return row * 512 + (kbyte ^ ((row & 15) << 4));
}

#ifndef GE_PIN
#define GE_PIN 0  // 1 (measured, not kept): fragment reads batched and pinned ahead of the MFMAs -- 95 -> 138 VGPRs (5 -> 3 waves per SIMD): bf16 step -2 %, fp32 step +0.3 %
#endif
template <bool SPLIT, bool OUT_BF16>
__global__ __launch_bounds__(GE_THREADS) void geo_embed_kernel(
    const float *__restrict__ pts, const int32_t *__restrict__ knn, const u16 *__restrict__ wd_hi,
    const u16 *__restrict__ wd_lo, const u16 *__restrict__ wa_hi, const u16 *__restrict__ wa_lo,
    const float *__restrict__ bias, const float *__restrict__ div_term, int n, float sigma_d, float factor_a,
    int reduce_mean, void *__restrict__ out_) {
  extern __shared__ float4 smem4[];
  char *Ahi = reinterpret_cast<char *>(smem4);           // [4 sets][32 rows][512 B]
  char *Alo = Ahi + 4 * GE_PAIRS * 512;                  // only when SPLIT
  const int b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nn = n * n;
  const int pair0 = blockIdx.x * GE_PAIRS;
  const float *P = pts + (size_t)b * n * 3;
  const int32_t *KNN = knn + (size_t)b * n * 3;

  // ---------------- phase 1: sinusoids of 4 x 32 index values -> LDS (bf16, A-operand layout)
  // (a) the 128 index values of the tile, ONE lane each (waves 0-1), parked in LDS
  float *sidx = reinterpret_cast<float *>(Ahi + (SPLIT ? 2 : 1) * 4 * GE_PAIRS * 512);  // [128]
  if (tid < 4 * GE_PAIRS) {
    const int set = tid >> 5, row = tid & 31;
    const int pair = min(pair0 + row, nn - 1);
    const int i = pair / n, j = pair - i * n;
    const float xi = P[i * 3], yi = P[i * 3 + 1], zi = P[i * 3 + 2];
    const float ax = P[j * 3] - xi, ay = P[j * 3 + 1] - yi, az = P[j * 3 + 2] - zi;  // anchor = p_j - p_i
    float idx;
    if (set == 0) {
      idx = sqrtf(ax * ax + ay * ay + az * az) / sigma_d;
    } else {
      const int q = KNN[i * 3 + set - 1];
      const float rx = P[q * 3] - xi, ry = P[q * 3 + 1] - yi, rz = P[q * 3 + 2] - zi;  // ref = p_knn - p_i
      const float cx = ry * az - rz * ay, cy = rz * ax - rx * az, cz = rx * ay - ry * ax;
      const float sinv = sqrtf(cx * cx + cy * cy + cz * cz);
      // torch.sum starts from +0, so an all-(-0) dot product (i == j, anchor = 0) is +0 there: keep
      // atan2(0, +0) = 0 rather than atan2(0, -0) = pi
      const float cosv = 0.0f + (rx * ax + ry * ay + rz * az);
      idx = atan2f(sinv, cosv) * factor_a;
    }
    sidx[tid] = idx;
  }
  __syncthreads();
  // (b) lane t of every wave owns frequencies t and t+64; the upper one has div_term <= 1e-2, so its
  // argument stays below ~0.3 for any index < 32 and needs no range reduction
  const float div0 = div_term[lane], div1 = div_term[lane + 64];
  for (int c = wave; c < 4 * GE_PAIRS; c += GE_THREADS / 64) {
    const float idx = sidx[c];
    float s0, c0, s1, c1;
    const float w1 = idx * div1;
    if (!SPLIT) {
      // bf16 operands (8-bit mantissa): the hardware v_sin_f32 / v_cos_f32 (argument in revolutions,
      // ~1e-6 absolute error) are far more accurate than the rounding that follows
      // (both frequency halves: the polynomial for the small-argument half cost ~20 VALU instructions per
      // pair against two transcendental issues, and VALU issue is what bounds this kernel)
      const float r0 = idx * div0 * 0.15915494309189535f, r1 = w1 * 0.15915494309189535f;
      s0 = __builtin_amdgcn_sinf(r0);
      c0 = __builtin_amdgcn_cosf(r0);
      s1 = __builtin_amdgcn_sinf(r1);
      c1 = __builtin_amdgcn_cosf(r1);
    } else {
      sincos_cw(idx * div0, s0, c0);
      if (fabsf(idx) < 32.f) {  // wave-uniform
        sincos_small(w1, s1, c1);
      } else {
        sincos_cw(w1, s1, c1);
      }
    }
    // channel 2t = sin(w_t), 2t+1 = cos(w_t)  (transformer.py:278-282)
    const int rbase = c;  // = set * GE_PAIRS + row
    const uint32_t p0 = cvt_pk_bf16(s0, c0), p1 = cvt_pk_bf16(s1, c1);
    *reinterpret_cast<uint32_t *>(Ahi + ge_swz(rbase, lane * 4)) = p0;
    *reinterpret_cast<uint32_t *>(Ahi + ge_swz(rbase, (lane + 64) * 4)) = p1;
    if (SPLIT) {
      const uint32_t q0 = cvt_pk_bf16(s0 - __uint_as_float(p0 << 16), c0 - __uint_as_float(p0 & 0xFFFF0000u));
      const uint32_t q1 = cvt_pk_bf16(s1 - __uint_as_float(p1 << 16), c1 - __uint_as_float(p1 & 0xFFFF0000u));
      *reinterpret_cast<uint32_t *>(Alo + ge_swz(rbase, lane * 4)) = q0;
      *reinterpret_cast<uint32_t *>(Alo + ge_swz(rbase, (lane + 64) * 4)) = q1;
    }
  }
  __syncthreads();

  // ---------------- phase 2: [4 x (32 x 256)] x (256 x 32-channel slice) on the matrix cores
  f32x16 acc[4];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;
  const int arow = lane & 31, khalf = lane >> 5;
  const int ch = wave * 32 + (lane & 31);  // B operand column = output channel
  // weights arrive in MFMA-fragment order [k-step 16][channel group 8][lane 64][8 bf16]: one B-operand
  // load of a wave is 1 KiB contiguous (8 cache lines instead of 32 strided ones)
  const bf16x8 *Wd_hi = reinterpret_cast<const bf16x8 *>(wd_hi) + wave * 64 + lane;
  const bf16x8 *Wa_hi = reinterpret_cast<const bf16x8 *>(wa_hi) + wave * 64 + lane;
  const bf16x8 *Wd_lo = SPLIT ? reinterpret_cast<const bf16x8 *>(wd_lo) + wave * 64 + lane : nullptr;
  const bf16x8 *Wa_lo = SPLIT ? reinterpret_cast<const bf16x8 *>(wa_lo) + wave * 64 + lane : nullptr;
  // B (weight) fragments come from L2 (~500+ cycles): keep a 4-deep register prefetch ring ahead of the MFMAs
  constexpr int PF = 4;
  bf16x8 rbd[PF], rba[PF], rbdl[PF], rbal[PF];
#pragma unroll
  for (int i = 0; i < PF; ++i) {
    rbd[i] = Wd_hi[i * 512];
    rba[i] = Wa_hi[i * 512];
    if (SPLIT) {
      rbdl[i] = Wd_lo[i * 512];
      rbal[i] = Wa_lo[i * 512];
    }
  }
#pragma unroll
  for (int ks = 0; ks < GE_DIM / 16; ++ks) {
    const int kidx = ks * 2 + khalf;  // which group of 8 k's this lane holds
    const int slot = ks % PF;
    const bf16x8 bd = rbd[slot], ba = rba[slot];
    bf16x8 bdl, bal;
    if (SPLIT) {
      bdl = rbdl[slot];
      bal = rbal[slot];
    }
    if (ks + PF < GE_DIM / 16) {
      rbd[slot] = Wd_hi[(ks + PF) * 512];
      rba[slot] = Wa_hi[(ks + PF) * 512];
      if (SPLIT) {
        rbdl[slot] = Wd_lo[(ks + PF) * 512];
        rbal[slot] = Wa_lo[(ks + PF) * 512];
      }
    }
#if GE_PIN
    // all A fragments of the k-step read as one batch, and neither they nor the ring's loads above may sink below this point: hipcc
    // otherwise moves every read next to its MFMAs (the 4-deep ring arrived in the ISA as `s_waitcnt vmcnt(1)` right behind the loads)
    bf16x8 af[4], afl[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      af[s] = *reinterpret_cast<const bf16x8 *>(Ahi + ge_swz(s * GE_PAIRS + arow, kidx * 16));
      if (SPLIT) afl[s] = *reinterpret_cast<const bf16x8 *>(Alo + ge_swz(s * GE_PAIRS + arow, kidx * 16));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8 bw = s == 0 ? bd : ba;
      acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s], bw, acc[s], 0, 0, 0);
      if (SPLIT) {
        const bf16x8 bl = s == 0 ? bdl : bal;
        acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s], bl, acc[s], 0, 0, 0);
        acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afl[s], bw, acc[s], 0, 0, 0);
      }
    }
#else
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8 a = *reinterpret_cast<const bf16x8 *>(Ahi + ge_swz(s * GE_PAIRS + arow, kidx * 16));
      const bf16x8 bw = s == 0 ? bd : ba;
      acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bw, acc[s], 0, 0, 0);
      if (SPLIT) {
        const bf16x8 al = *reinterpret_cast<const bf16x8 *>(Alo + ge_swz(s * GE_PAIRS + arow, kidx * 16));
        const bf16x8 bl = s == 0 ? bdl : bal;
        acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bl, acc[s], 0, 0, 0);
        acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bw, acc[s], 0, 0, 0);
      }
    }
#endif
  }

  // ---------------- phase 3: E = d + reduce_k(a_k) + (b_d + b_a)
  // (one base address per lane + a row stride, the tail check only in a cloud's last tile, the reduction
  // flag hoisted out of the register loop: the epilogue was the largest VALU block of the kernel)
  const float bsum = bias[ch];
  const size_t obase = ((size_t)b * nn + pair0 + 4 * khalf) * GE_DIM + ch;
  const bool full = pair0 + GE_PAIRS <= nn;  // wave-uniform
  auto epilogue = [&](auto mean_tag) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2);  // 32x32 C/D layout (+ 4 * khalf, in the base)
      const float a = decltype(mean_tag)::value ? (acc[1][r] + acc[2][r] + acc[3][r]) * (1.f / 3.f)
                                                : fmaxf(fmaxf(acc[1][r], acc[2][r]), acc[3][r]);
      const float v = acc[0][r] + a + bsum;
      if (full || pair0 + 4 * khalf + row < nn) {
        if (OUT_BF16)
          reinterpret_cast<u16 *>(out_)[obase + (size_t)row * GE_DIM] = (u16)cvt_pk_bf16(v, v);
        else
          reinterpret_cast<float *>(out_)[obase + (size_t)row * GE_DIM] = v;
      }
    }
  };
  if (reduce_mean)
    epilogue(std::true_type{});
  else
    epilogue(std::false_type{});
}

// ---- Round 5: the embedding as TABLE LOOKUPS.  proj(sinus(x)) is a smooth function of ONE scalar per output channel -- a sum of 128
// sinusoids of frequencies 10000^(-i/128) <= 1 --, so instead of generating 256 sinusoids per index and contracting them with a 256 x 256
// matrix (131 kflop per index, four indices per pair: what the kernel above does on the matrix cores) the two functions
//     f_d(x) = W_d sinus(x),   f_a(x) = W_a sinus(x)          (256 channels each, biases added at the end)
// are tabulated once per weight version at spacing h = 1 / GT_HINV (fp32, `ops.geo_embedding`) and evaluated by NP-point Lagrange
// interpolation.  NP = 4 (the autocast forward, bf16 result): error <= 0.0234 h^4 max|d4f/dx4| = 9e-5 max|d4f/dx4| at h = 1/4, the fourth
// derivative bounded by sum_i w_i^4 (|W_sin| + |W_cos|) ~ 0.5 for 1/16-sized weights -- two orders below the bf16 result's resolution.
// NP = 6 (the fp32 forward): error <= 0.0235 h^6 max|d6f/dx6| = 5.7e-6 max|d6f/dx6| -- fp32 class, as the split-operand kernel above.
// The tables (distance range [0, 16): 64 + NP - 1 rows, angle range [0, pi factor_a]: ~48 + NP rows, 1 KiB per row) live in LDS; a distance
// index past the LDS range reads its rows from the full table in global memory, one past that the defining sum (wave-uniform branches).
// A wave takes a contiguous share of the pairs: 64 lanes compute the four indices, table rows and weights of 64 columns j at once, then
// every pair is 4 lookups x NP rows x 16 B per lane (4 channels) and 16 NP fma; the result row (512 B / 1 KiB) leaves as one coalesced store.
constexpr int GT_HINV = 4;  // table nodes per unit index

typedef float gt_f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const gt_f4 gt_lds_f4;

__device__ __forceinline__ float gt_bcast(float v, int l) {
#if defined(GT_ABL) && GT_ABL == 1   // (timing probe, scripts/ubench/geo_table_abl.cpp: no weight broadcasts)
  return v;
#endif
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// sum_m w[m] T[m] for this lane's four channels; the weights are lane l's, broadcast (SGPRs).  Scalar v_fma_f32 on purpose: packed fp32
// instructions return wrong values next to another kernel's MFMA waves (tests/test_abi.py).
template <int NP>
__device__ __forceinline__ void gt_lerp(const gt_f4 (&t)[NP], const float (&w)[NP], int l, float (&v)[4]) {
  float s = gt_bcast(w[0], l);
  v[0] = t[0].x * s, v[1] = t[0].y * s, v[2] = t[0].z * s, v[3] = t[0].w * s;
#pragma unroll
  for (int m = 1; m < NP; ++m) {
    s = gt_bcast(w[m], l);
    v[0] = fmaf(t[m].x, s, v[0]), v[1] = fmaf(t[m].y, s, v[1]), v[2] = fmaf(t[m].z, s, v[2]), v[3] = fmaf(t[m].w, s, v[3]);
  }
}

// Lagrange weights of the nodes -(NP / 2 - 1) .. NP / 2 at f in [0, 1)
template <int NP>
__device__ __forceinline__ void gt_weights(float f, float (&w)[NP]) {
  constexpr int LO = NP / 2 - 1;
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    float num = 1.f, den = 1.f;
#pragma unroll
    for (int m = 0; m < NP; ++m)
      if (m != j) {
        num *= f - (float)(m - LO);
        den *= (float)(j - m);
      }
    w[j] = num * (1.f / den);
  }
}

// TRAIN (round 6, the forward of the training step): additionally writes, per output element, WHICH of the three angle terms was the
// maximum (one byte per channel, first maximum on ties: where torch.max sends the gradient) for geo_embed_table_bwd_kernel.
template <bool OUT_BF16, bool MEAN, int NP, bool TRAIN = false>
__global__ __launch_bounds__(1024) void geo_embed_table_kernel(const float *__restrict__ pts, const int32_t *__restrict__ knn,
                                                               const float *__restrict__ tab_d, int rows_d, const float *__restrict__ tab_a,
                                                               int rows_a, const float *__restrict__ bias, const float *__restrict__ w_d,
                                                               const float *__restrict__ div_term, int B, int n, float sigma_d,
                                                               float factor_a, void *__restrict__ out_, uint32_t *__restrict__ amax = nullptr) {
  extern __shared__ float4 gt_smem[];
  const int rd_l = min(rows_d, 16 * GT_HINV + NP - 1);  // distance rows [0, rd_l) (index range [0, 16)), then the angle rows
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
  {  // tables -> LDS, eight loads in flight per thread
    const int nd = rd_l * 64, ne = nd + rows_a * 64;
    for (int e0 = tid; e0 < ne; e0 += 8 * (int)blockDim.x) {
      float4 r[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + u * (int)blockDim.x;
        if (e < ne) r[u] = e < nd ? reinterpret_cast<const float4 *>(tab_d)[e] : reinterpret_cast<const float4 *>(tab_a)[e - nd];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + u * (int)blockDim.x;
        if (e < ne) gt_smem[e] = r[u];
      }
    }
  }
  __syncthreads();
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)gt_smem + lane * 16;  // this lane's 4 channels of row 0
  const gt_f4 b4 = reinterpret_cast<const gt_f4 *>(bias)[lane];
  // row r of a table holds x = (r - (NP / 2 - 1)) / HINV: the NP rows from floor(x HINV) on are the nodes around x
  const float a_max = (float)(rows_a - NP) / (float)GT_HINV;  // (angle indices are <= pi factor_a by construction; clamped for safety)
  const float d_max = (float)(rows_d - NP) / (float)GT_HINV;
  // every wave takes an equal contiguous share of the B n n pairs (rows of 197 columns do not divide into waves evenly)
  const long P = (long)B * n * n, W = (long)gridDim.x * nwaves, per = (P + W - 1) / W;
  const long p0 = ((long)blockIdx.x * nwaves + wave) * per, p1 = min(P, p0 + per);
  for (long p = p0; p < p1;) {
    const long row = p / n;
    const int j0 = (int)(p - row * n), cols = (int)min((long)min(64, n - j0), p1 - p);
    const int b = (int)(row / n), i = (int)(row - (long)b * n);
    const float *Pt = pts + (size_t)b * n * 3;
    const int32_t *KNN = knn + (size_t)row * 3;
    const float xi = Pt[i * 3], yi = Pt[i * 3 + 1], zi = Pt[i * 3 + 2];
    // ---- phase A, one column per lane: the four indices (the formulas of geo_embed_kernel), their table rows and Lagrange weights
    const int j = min(j0 + lane, n - 1);
    const float ax = Pt[j * 3] - xi, ay = Pt[j * 3 + 1] - yi, az = Pt[j * 3 + 2] - zi;  // anchor = p_j - p_i
    float idx[4];
    idx[0] = sqrtf(ax * ax + ay * ay + az * az) / sigma_d;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int q = KNN[k];
      const float rx = Pt[q * 3] - xi, ry = Pt[q * 3 + 1] - yi, rz = Pt[q * 3 + 2] - zi;  // ref = p_knn - p_i
      const float cx = ry * az - rz * ay, cy = rz * ax - rx * az, cz = rx * ay - ry * ax;
      const float sinv = sqrtf(cx * cx + cy * cy + cz * cz);
      const float cosv = 0.0f + (rx * ax + ry * ay + rz * az);  // (+0: atan2(0, +0) = 0 for i == j as torch.sum gives it)
      idx[1 + k] = fminf(atan2f(sinv, cosv) * factor_a, a_max);
    }
    int off[4];       // LDS byte offset of the first of the NP rows
    float w[4][NP];   // Lagrange weights at f = frac(x HINV)
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const float x = idx[s4] * (float)GT_HINV, fl = floorf(x);
      const int r0 = (int)fl;
      gt_weights<NP>(x - fl, w[s4]);
      if (s4 == 0)
        off[0] = (idx[0] <= d_max) ? (r0 + NP - 1 < rd_l ? r0 * 1024 : -1 - r0) : (int)0x80000000;   // >= 0 LDS, < 0 global row, INT_MIN: the sum
      else
        off[s4] = (rd_l + r0) * 1024;
    }
    char *orow = reinterpret_cast<char *>(out_) + ((size_t)row * n + j0) * 256 * (OUT_BF16 ? 2 : 4) + lane * (OUT_BF16 ? 8 : 16);
    // ---- phase B, one pair at a time, four channels per lane
    for (int l = 0; l < cols; ++l) {
      float v[4][4];
      const int od = __builtin_amdgcn_readlane(off[0], l);
      // the first angle's rows before the distance lookup (always LDS): their latency runs under it
      gt_f4 ta[NP], tb[NP];
      auto lds_rows = [&](int k, gt_f4 (&dst)[NP]) {
        gt_lds_f4 *t = (gt_lds_f4 *)(uintptr_t)(lds0 + __builtin_amdgcn_readlane(off[k], l));
#pragma unroll
        for (int m = 0; m < NP; ++m) dst[m] = t[64 * m];
#if defined(GT_ABL) && GT_ABL == 2   // (timing probe: no LDS reads)
#pragma unroll
        for (int m = 0; m < NP; ++m) dst[m] = gt_f4{(float)(uintptr_t)t, 1.f, 2.f, 3.f};
#endif
      };
      lds_rows(1, ta);
      if (od >= 0) {
        gt_lds_f4 *t = (gt_lds_f4 *)(uintptr_t)(lds0 + od);
#pragma unroll
        for (int m = 0; m < NP; ++m) tb[m] = t[64 * m];
        gt_lerp<NP>(tb, w[0], l, v[0]);
      } else if (od != (int)0x80000000) {   // past the LDS-resident rows: the full table in global memory
        const gt_f4 *g = reinterpret_cast<const gt_f4 *>(tab_d) + (size_t)(-1 - od) * 64 + lane;
#pragma unroll
        for (int m = 0; m < NP; ++m) tb[m] = g[64 * m];
        gt_lerp<NP>(tb, w[0], l, v[0]);
      } else {
        // (rare) a distance index past the table -- clouds that are not radius-normalised --: the defining sum itself,
        // f_d(x)[c] = sum_i W[c][2i] sin(x w_i) + W[c][2i + 1] cos(x w_i), for this lane's four channels.  (NaN coordinates come here too.)
        const float xd = gt_bcast(idx[0], l);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int f = 0; f < 128; ++f) {
          const float om = xd * div_term[f];
          const float sv = sinf(om), cv = cosf(om);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float2 wv = *reinterpret_cast<const float2 *>(w_d + (size_t)(lane * 4 + c) * 256 + 2 * f);
            acc[c] += wv.x * sv + wv.y * cv;
          }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) v[0][c] = acc[c];
      }
      lds_rows(2, tb);
      gt_lerp<NP>(ta, w[1], l, v[1]);
      lds_rows(3, ta);
      gt_lerp<NP>(tb, w[2], l, v[2]);
      gt_lerp<NP>(ta, w[3], l, v[3]);
      float o[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float a = MEAN ? (v[1][c] + v[2][c] + v[3][c]) * (1.f / 3.f) : fmaxf(fmaxf(v[1][c], v[2][c]), v[3][c]);
        o[c] = v[0][c] + a + b4[c];
      }
#if defined(GT_ABL) && GT_ABL == 3   // (timing probe: one store per chunk)
      if (l != cols - 1 || o[0] == 123.f) continue;
#endif
      if (OUT_BF16)
        *reinterpret_cast<uint2 *>(orow + (size_t)l * 512) = make_uint2(cvt_pk_bf16(o[0], o[1]), cvt_pk_bf16(o[2], o[3]));
      else
        *reinterpret_cast<gt_f4 *>(orow + (size_t)l * 1024) = gt_f4{o[0], o[1], o[2], o[3]};
      if (TRAIN && !MEAN) {
        uint32_t am = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const uint32_t k = (v[1][c] >= v[2][c] && v[1][c] >= v[3][c]) ? 0u : (v[2][c] >= v[3][c] ? 1u : 2u);
          am |= k << (8 * c);
        }
        amax[((size_t)row * n + j0 + l) * 64 + lane] = am;
      }
    }
    p += cols;
  }
}

// ---- Round 6: the gradient of the table form, for the training step (transformer.py:303-350 under autograd).  E = sum_m w_m T[r + m] is
// linear in the tables, so dT[r + m][c] += w_m dE[p][c]: the backward pass is the forward kernel's mirror -- the same indices, rows and
// Lagrange weights -- and instead of reading NP table rows per lookup it ADDS into a gradient table of the same shape that lives in LDS.
// No atomics (a first form with ds_add_f32 from 16 waves took 9.4 ms per launch: ~190 cycles per LDS float-atomic instruction): a workgroup
// is FOUR waves, wave w owns channels [64 w, 64 w + 64) of the table -- its own 256-byte column block of every row, one channel per lane --
// and walks ALL pairs of the workgroup's share in order with plain ds_read / fma / ds_write (program order of one wave: deterministic).
// The maximum over the three angle terms routes dE to the term the forward recorded (`amax`): a lane adds into the rows of ITS channel's
// term only (per-lane row offset and weights).  The workgroup's table leaves as one coalesced copy into `ws[blockIdx.x]`; the host sums
// the workgroups' tables and maps them back to the weights (dW = dT^T S_grid, db = sum_r dT[r]: the Lagrange weights of a lookup add
// up to one).  Distance indices past the LDS-resident rows (radius-normalised clouds never have them) go to the full-size table `full_d`
// with global atomics.  MEAN: a third of dE to each term.
template <bool MEAN, int NP>
__global__ __launch_bounds__(256) void geo_embed_table_bwd_kernel(const float *__restrict__ pts, const int32_t *__restrict__ knn, int rows_d, int rows_a, int B, int n,
                                                                  float sigma_d, float factor_a, const float *__restrict__ dE, const uint8_t *__restrict__ amax,
                                                                  float *__restrict__ ws, float *__restrict__ full_d, int *__restrict__ past_table) {
  extern __shared__ float4 gt_smem[];
  const int rd_l = min(rows_d, 16 * GT_HINV + NP - 1);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nrows = rd_l + rows_a;
  for (int e = tid; e < nrows * 64; e += blockDim.x) gt_smem[e] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  float *const tab = reinterpret_cast<float *>(gt_smem) + wave * nrows * 64 + lane;  // this lane's channel of the wave's column block, row r at + 64 r
  const int ch = wave * 64 + lane;
  const float a_max = (float)(rows_a - NP) / (float)GT_HINV;
  const float d_max = (float)(rows_d - NP) / (float)GT_HINV;
  const long P = (long)B * n * n, per = (P + gridDim.x - 1) / gridDim.x;
  const long p0 = (long)blockIdx.x * per, p1 = min(P, p0 + per);   // (all four waves walk the same pairs)
  for (long p = p0; p < p1;) {
    const long row = p / n;
    const int j0 = (int)(p - row * n), cols = (int)min((long)min(64, n - j0), p1 - p);
    const int b = (int)(row / n), i = (int)(row - (long)b * n);
    const float *Pt = pts + (size_t)b * n * 3;
    const int32_t *KNN = knn + (size_t)row * 3;
    const float xi = Pt[i * 3], yi = Pt[i * 3 + 1], zi = Pt[i * 3 + 2];
    const int j = min(j0 + lane, n - 1);
    const float ax = Pt[j * 3] - xi, ay = Pt[j * 3 + 1] - yi, az = Pt[j * 3 + 2] - zi;
    float idx[4];
    idx[0] = sqrtf(ax * ax + ay * ay + az * az) / sigma_d;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int q = KNN[k];
      const float rx = Pt[q * 3] - xi, ry = Pt[q * 3 + 1] - yi, rz = Pt[q * 3 + 2] - zi;
      const float cx = ry * az - rz * ay, cy = rz * ax - rx * az, cz = rx * ay - ry * ax;
      const float sinv = sqrtf(cx * cx + cy * cy + cz * cz);
      const float cosv = 0.0f + (rx * ax + ry * ay + rz * az);
      idx[1 + k] = fminf(atan2f(sinv, cosv) * factor_a, a_max);
    }
    int off[4];       // first of the NP rows in the gradient table, in rows (distance: < 0 = -1 - row of the full table; INT_MIN: past it)
    float w[4][NP];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const float x = idx[s4] * (float)GT_HINV, fl = floorf(x);
      const int r0 = (int)fl;
      gt_weights<NP>(x - fl, w[s4]);
      if (s4 == 0)
        off[0] = (idx[0] <= d_max) ? (r0 + NP - 1 < rd_l ? r0 : -1 - r0) : (int)0x80000000;
      else
        off[s4] = rd_l + r0;
    }
    const float *grow = dE + ((size_t)row * n + j0) * 256 + ch;
    const uint8_t *arow = amax + ((size_t)row * n + j0) * 256 + ch;
    float g_next = grow[0];
    uint32_t am_next = MEAN ? 0u : arow[0];
    for (int l = 0; l < cols; ++l) {
      const float g = g_next;
      const uint32_t am = am_next;
      if (l + 1 < cols) {  // the next pair's gradient element rides under this pair's table updates
        g_next = grow[(size_t)(l + 1) * 256];
        if (!MEAN) am_next = arow[(size_t)(l + 1) * 256];
      }
      const int od = __builtin_amdgcn_readlane(off[0], l);
      if (od >= 0) {
        float *t = tab + od * 64;
        float cur[NP];
#pragma unroll
        for (int m = 0; m < NP; ++m) cur[m] = t[m * 64];
#pragma unroll
        for (int m = 0; m < NP; ++m) t[m * 64] = fmaf(gt_bcast(w[0][m], l), g, cur[m]);
      } else if (od != (int)0x80000000) {
        float *t = full_d + (size_t)(-1 - od) * 256 + ch;
#pragma unroll
        for (int m = 0; m < NP; ++m) atomicAdd(t + m * 256, gt_bcast(w[0][m], l) * g);
      } else if (lane == 0) {
        *past_table = 1;  // a distance index past the table (never for radius-normalised clouds): the host refuses the result
      }
      if (MEAN) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          float *t = tab + __builtin_amdgcn_readlane(off[1 + k], l) * 64;
          float cur[NP];
#pragma unroll
          for (int m = 0; m < NP; ++m) cur[m] = t[m * 64];
#pragma unroll
          for (int m = 0; m < NP; ++m) t[m * 64] = fmaf(gt_bcast(w[1 + k][m], l), g * (1.f / 3.f), cur[m]);
        }
      } else {
        // this lane's channel took its maximum from term `am`: that term's rows and weights (lane-varying)
        const int o0 = __builtin_amdgcn_readlane(off[1], l), o1 = __builtin_amdgcn_readlane(off[2], l), o2 = __builtin_amdgcn_readlane(off[3], l);
        float *t = tab + (am == 0 ? o0 : am == 1 ? o1 : o2) * 64;
        float cur[NP];
#pragma unroll
        for (int m = 0; m < NP; ++m) cur[m] = t[m * 64];
#pragma unroll
        for (int m = 0; m < NP; ++m) {
          const float s0 = gt_bcast(w[1][m], l), s1 = gt_bcast(w[2][m], l), s2 = gt_bcast(w[3][m], l);
          t[m * 64] = fmaf(am == 0 ? s0 : am == 1 ? s1 : s2, g, cur[m]);
        }
      }
    }
    p += cols;
  }
  __syncthreads();
  // ws[blockIdx.x][r][c]: the wave-major LDS image back to (rows, 256)
  float *o = ws + (size_t)blockIdx.x * nrows * 256;
  for (int r = 0; r < nrows; ++r) o[(size_t)r * 256 + ch] = tab[r * 64];
}

}  // namespace unopose

using namespace unopose;

extern "C" {

int unopose_geo_embedding(const float *points, int B, int n, const void *wd_hi, const void *wd_lo, const void *wa_hi,
                          const void *wa_lo, const float *bias_sum, const float *div_term, float sigma_d,
                          float factor_a, int reduce_mean, int split, int out_bf16, int32_t *knn_ws, void *out,
                          unopose_stream_t stream) {
  UNOPOSE_REQUIRE(points && wd_hi && wa_hi && bias_sum && div_term && knn_ws && out, "geo_embedding: null pointer");
  UNOPOSE_REQUIRE(!split || (wd_lo && wa_lo), "geo_embedding: split precision needs the lo weight parts");
  UNOPOSE_REQUIRE(B >= 0 && n >= 4 && B <= 65535, "geo_embedding: bad sizes (n must be >= 4 for 3-NN)");
  if (B == 0) return UNOPOSE_OK;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(geo_knn_kernel, dim3(cdiv(n, 256), B), dim3(256), 0, s, points, n, knn_ws);
  int rc = check_launch("geo_knn");
  if (rc) return rc;
  dim3 grid(cdiv((long)n * n, GE_PAIRS), B);
  const size_t lds = (size_t)4 * GE_PAIRS * 512 * (split ? 2 : 1) + 4 * GE_PAIRS * sizeof(float);
  const u16 *dh = (const u16 *)wd_hi, *dl = (const u16 *)wd_lo, *ah = (const u16 *)wa_hi, *al = (const u16 *)wa_lo;
#define UNOPOSE_GE_LAUNCH(SP, OB)                                                                               \
  hipLaunchKernelGGL((geo_embed_kernel<SP, OB>), grid, dim3(GE_THREADS), lds, s, points, knn_ws, dh, dl, ah, al, \
                     bias_sum, div_term, n, sigma_d, factor_a, reduce_mean, out)
  if (split) {
    static bool opt_a[64], opt_b[64];
    if (lds_optin(opt_a, (const void *)geo_embed_kernel<true, false>, lds, "geo_embedding") != UNOPOSE_OK ||
        lds_optin(opt_b, (const void *)geo_embed_kernel<true, true>, lds, "geo_embedding") != UNOPOSE_OK)
      return UNOPOSE_ELAUNCH;
    if (out_bf16) UNOPOSE_GE_LAUNCH(true, true); else UNOPOSE_GE_LAUNCH(true, false);
  } else {
    if (out_bf16) UNOPOSE_GE_LAUNCH(false, true); else UNOPOSE_GE_LAUNCH(false, false);
  }
#undef UNOPOSE_GE_LAUNCH
  return check_launch("geo_embedding");
}

int unopose_geo_embedding_table(const float *points, int B, int n, const float *tab_d, int rows_d, const float *tab_a, int rows_a,
                                const float *bias_sum, const float *w_d, const float *div_term, int hinv, int npoint, float sigma_d, float factor_a, int reduce_mean, int out_bf16, int32_t *knn_ws,
                                void *out, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(points && tab_d && tab_a && bias_sum && w_d && div_term && knn_ws && out, "geo_embedding_table: null pointer");
  UNOPOSE_REQUIRE(B >= 0 && n >= 4 && B <= 65535, "geo_embedding_table: bad sizes (n must be >= 4 for 3-NN)");
  UNOPOSE_REQUIRE(hinv == GT_HINV && (npoint == 4 || npoint == 6) && rows_d >= 8 && rows_a >= 8,
                  "geo_embedding_table: tables must be spaced 1 / %d (got 1 / %d) for 4- or 6-point interpolation (got %d)", GT_HINV, hinv, npoint);
  const int rdl = 16 * GT_HINV + npoint - 1;
  const size_t lds = ((size_t)(rows_d < rdl ? rows_d : rdl) + rows_a) * 1024;
  UNOPOSE_REQUIRE(lds <= 150 * 1024, "geo_embedding_table: %d angle rows do not fit the LDS-resident table", rows_a);
  if (B == 0) return UNOPOSE_OK;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(geo_knn_kernel, dim3(cdiv(n, 256), B), dim3(256), 0, s, points, n, knn_ws);
  int rc = check_launch("geo_knn");
  if (rc) return rc;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) cus = 256;
  const long pairs = (long)B * n * n;
  const int grid = (int)(pairs < (long)cus * 256 ? (pairs + 255) / 256 : cus);  // one 16-wave workgroup per CU (the tables fill its LDS), >= 16 pairs per wave
#define UNOPOSE_GT_LAUNCH(BF, MEAN, NPT)                                                                                                       \
  do {                                                                                                                                         \
    static bool opt[64];                                                                                                                       \
    if (lds_optin(opt, (const void *)geo_embed_table_kernel<BF, MEAN, NPT>, lds, "geo_embedding_table") != UNOPOSE_OK) return UNOPOSE_ELAUNCH; \
    hipLaunchKernelGGL((geo_embed_table_kernel<BF, MEAN, NPT>), dim3(grid), dim3(1024), lds, s, points, knn_ws, tab_d, rows_d, tab_a, rows_a,  \
                       bias_sum, w_d, div_term, B, n, sigma_d, factor_a, out);                                                                 \
  } while (0)
#define UNOPOSE_GT_ORDER(BF, MEAN)         \
  do {                                     \
    if (npoint == 4)                       \
      UNOPOSE_GT_LAUNCH(BF, MEAN, 4);      \
    else                                   \
      UNOPOSE_GT_LAUNCH(BF, MEAN, 6);      \
  } while (0)
  if (out_bf16 && reduce_mean) UNOPOSE_GT_ORDER(true, true);
  else if (out_bf16) UNOPOSE_GT_ORDER(true, false);
  else if (reduce_mean) UNOPOSE_GT_ORDER(false, true);
  else UNOPOSE_GT_ORDER(false, false);
#undef UNOPOSE_GT_ORDER
#undef UNOPOSE_GT_LAUNCH
  return check_launch("geo_embedding_table");
}

static int gt_train_grid(int B, int n) {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) cus = 256;
  const long pairs = (long)B * n * n;
  return (int)(pairs < (long)cus * 256 ? (pairs + 255) / 256 : cus);
}

int unopose_geo_embedding_train_workgroups(int B, int n) { return gt_train_grid(B, n); }

int unopose_geo_embedding_train_forward(const float *points, int B, int n, const float *tab_d, int rows_d, const float *tab_a, int rows_a, const float *bias_sum,
                                        const float *w_d, const float *div_term, int hinv, float sigma_d, float factor_a, int reduce_mean, int32_t *knn,
                                        float *out, void *amax, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(points && tab_d && tab_a && bias_sum && w_d && div_term && knn && out && amax, "geo_embedding_train_forward: null pointer");
  UNOPOSE_REQUIRE(B >= 0 && n >= 4 && B <= 65535, "geo_embedding_train_forward: bad sizes (n must be >= 4 for 3-NN)");
  UNOPOSE_REQUIRE(hinv == GT_HINV && rows_d >= 8 && rows_a >= 8, "geo_embedding_train_forward: tables must be spaced 1 / %d (got 1 / %d)", GT_HINV, hinv);
  const int rdl = 16 * GT_HINV + 6 - 1;
  const size_t lds = ((size_t)(rows_d < rdl ? rows_d : rdl) + rows_a) * 1024;
  UNOPOSE_REQUIRE(lds <= 150 * 1024, "geo_embedding_train_forward: %d angle rows do not fit the LDS-resident table", rows_a);
  if (B == 0) return UNOPOSE_OK;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(geo_knn_kernel, dim3(cdiv(n, 256), B), dim3(256), 0, s, points, n, knn);
  int rc = check_launch("geo_knn");
  if (rc) return rc;
  const int grid = gt_train_grid(B, n);
#define UNOPOSE_GTT_LAUNCH(MEAN)                                                                                                                          \
  do {                                                                                                                                                    \
    static bool opt[64];                                                                                                                                  \
    if (lds_optin(opt, (const void *)geo_embed_table_kernel<false, MEAN, 6, true>, lds, "geo_embedding_train_forward") != UNOPOSE_OK) return UNOPOSE_ELAUNCH; \
    hipLaunchKernelGGL((geo_embed_table_kernel<false, MEAN, 6, true>), dim3(grid), dim3(1024), lds, s, points, knn, tab_d, rows_d, tab_a, rows_a, bias_sum, w_d, \
                       div_term, B, n, sigma_d, factor_a, (void *)out, (uint32_t *)amax);                                                                 \
  } while (0)
  if (reduce_mean) UNOPOSE_GTT_LAUNCH(true); else UNOPOSE_GTT_LAUNCH(false);
#undef UNOPOSE_GTT_LAUNCH
  return check_launch("geo_embedding_train_forward");
}

int unopose_geo_embedding_train_backward(const float *points, const int32_t *knn, int B, int n, int rows_d, int rows_a, int hinv, float sigma_d, float factor_a,
                                         int reduce_mean, const float *dE, const void *amax, float *ws, float *full_d, int *past_table,
                                         unopose_stream_t stream) {
  UNOPOSE_REQUIRE(points && knn && dE && amax && ws && full_d && past_table, "geo_embedding_train_backward: null pointer");
  UNOPOSE_REQUIRE(B >= 0 && n >= 4 && B <= 65535 && hinv == GT_HINV && rows_d >= 8 && rows_a >= 8, "geo_embedding_train_backward: bad sizes");
  const int rdl = 16 * GT_HINV + 6 - 1;
  const size_t lds = ((size_t)(rows_d < rdl ? rows_d : rdl) + rows_a) * 1024;
  UNOPOSE_REQUIRE(lds <= 150 * 1024, "geo_embedding_train_backward: %d angle rows do not fit the LDS-resident table", rows_a);
  if (B == 0) return UNOPOSE_OK;
  hipStream_t s = (hipStream_t)stream;
  const int grid = gt_train_grid(B, n);
#define UNOPOSE_GTB_LAUNCH(MEAN)                                                                                                                \
  do {                                                                                                                                          \
    static bool opt[64];                                                                                                                        \
    if (lds_optin(opt, (const void *)geo_embed_table_bwd_kernel<MEAN, 6>, lds, "geo_embedding_train_backward") != UNOPOSE_OK) return UNOPOSE_ELAUNCH; \
    hipLaunchKernelGGL((geo_embed_table_bwd_kernel<MEAN, 6>), dim3(grid), dim3(256), lds, s, points, knn, rows_d, rows_a, B, n, sigma_d, factor_a, dE, \
                       (const uint8_t *)amax, ws, full_d, past_table);                                                                         \
  } while (0)
  if (reduce_mean) UNOPOSE_GTB_LAUNCH(true); else UNOPOSE_GTB_LAUNCH(false);
#undef UNOPOSE_GTB_LAUNCH
  return check_launch("geo_embedding_train_backward");
}

}  // extern "C"
