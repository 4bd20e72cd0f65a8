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

}  // extern "C"
