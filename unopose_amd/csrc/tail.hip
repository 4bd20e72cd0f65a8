// Fused "tail" of the post-LN transformer layers for gfx950 (C ABI part 2).
//
// Everything after the attention core of TransformerLayer / RPETransformerLayer / LinearTransformerLayer
// (core/unopose/model/transformer.py:151-193: AttentionLayer.linear + residual + LayerNorm, then
// AttentionOutput: expand 256->512, ReLU, squeeze 512->256, residual, LayerNorm) in ONE kernel:
//     r   = LN1(h Wl^T + bl + x)
//     out = LN2(r + relu(r We^T + be) Ws^T + bs)
// The reference (and a GEMM-library formulation) runs 3 GEMMs + ~8 element-wise / norm kernels with five
// round trips of the (rows,256|512) activations through HBM.  Here a workgroup of 4 wavefronts owns 64
// token rows; each wave keeps its 16 tokens' activations in registers for the whole chain, computed
// TRANSPOSED on v_mfma_f32_16x16x32_bf16 (D[channel][token] = W[channel][k] X[k][token]) so that
//   * LayerNorm statistics of a token are in-lane sums + two cross-lane exchanges, and
//   * a layer's C/D registers are directly the next layer's B operand (k order permuted to the C/D
//     register map; the matching weight fragments are two 8-byte reads).
// Weights stream through LDS once per workgroup (shared by the 4 waves), chunk by chunk; the 512-wide
// hidden activation never exists beyond one 32-channel chunk.
#include "common.h"

namespace unopose {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

__device__ __forceinline__ u16 tl_f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (u16)(u >> 16);
}
__device__ __forceinline__ float tl_bf2f(u16 h) { return __uint_as_float(((uint32_t)h) << 16); }

__device__ __forceinline__ bf16x8 tl_pack(const f32x4 &a, const f32x4 &b) {
  union { bf16x8 v; u16 u[8]; } p;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    p.u[e] = tl_f2bf(a[e]);
    p.u[4 + e] = tl_f2bf(b[e]);
  }
  return p.v;
}
// A-operand fragment in the permuted k order of a chained layer: 8 values = W[row][32s+4g+{0..3}] and
// W[row][32s+16+4g+{0..3}]  (row pointer already offset to column 32s)
__device__ __forceinline__ bf16x8 tl_wfrag(const u16 *wrow, int g) {
  union { bf16x8 v; bf16x4 h[2]; } f;
  f.h[0] = *reinterpret_cast<const bf16x4 *>(wrow + 4 * g);
  f.h[1] = *reinterpret_cast<const bf16x4 *>(wrow + 16 + 4 * g);
  return f.v;
}

constexpr int TL_D = 256, TL_H = 512;
// LDS chunk buffers (bf16): Wl: 256x(256+8); per FFN chunk: We 32x(256+8) and Ws 256x(32+8)
constexpr int TL_LDW = TL_D + 8;   // padded row length of Wl / We chunks
constexpr int TL_LDS2 = 32 + 8;    // padded row length of a Ws chunk

__device__ __forceinline__ void tl_ln(f32x4 (&y)[16], const float *__restrict__ w, const float *__restrict__ b, int g,
                                      float eps) {
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 16; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) s += y[t][r];
  s += __shfl_xor(s, 16);
  s += __shfl_xor(s, 32);
  const float mean = s * (1.f / TL_D);
  float q = 0.f;
#pragma unroll
  for (int t = 0; t < 16; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float d = y[t][r] - mean;
      q += d * d;
    }
  q += __shfl_xor(q, 16);
  q += __shfl_xor(q, 32);
  const float rstd = rsqrtf(q * (1.f / TL_D) + eps);
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const float4 wv = *reinterpret_cast<const float4 *>(w + t * 16 + 4 * g);
    const float4 bv = *reinterpret_cast<const float4 *>(b + t * 16 + 4 * g);
    y[t][0] = (y[t][0] - mean) * rstd * wv.x + bv.x;
    y[t][1] = (y[t][1] - mean) * rstd * wv.y + bv.y;
    y[t][2] = (y[t][2] - mean) * rstd * wv.z + bv.z;
    y[t][3] = (y[t][3] - mean) * rstd * wv.w + bv.w;
  }
}

constexpr int TL_WAVES = 8;                 // 8 waves x 16 tokens = 128 token rows per workgroup
constexpr int TL_THREADS = TL_WAVES * 64;

// h, x, out: (rows, 256) bf16.  Weights bf16 row-major [out][in]; biases / LN params fp32.
__global__ __launch_bounds__(TL_THREADS) void transformer_tail_kernel(
    const u16 *__restrict__ h, const u16 *__restrict__ x, long rows, const u16 *__restrict__ Wl,
    const float *__restrict__ bl, const float *__restrict__ ln1w, const float *__restrict__ ln1b,
    const u16 *__restrict__ We, const float *__restrict__ be, const u16 *__restrict__ Ws,
    const float *__restrict__ bs, const float *__restrict__ ln2w, const float *__restrict__ ln2b, float eps,
    u16 *__restrict__ out) {
  extern __shared__ float4 smem4[];
  u16 *sW = reinterpret_cast<u16 *>(smem4);          // phase 1: Wl chunks; phase 2: We / Ws chunks
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tk = lane & 15, g = lane >> 4;            // token column, lane group
  const long row0 = (long)blockIdx.x * (TL_WAVES * 16) + wave * 16;
  const long trow = min(row0 + tk, rows - 1);

  // B operand of the first GEMM straight from memory: h[token][32s + 8g .. +7]
  bf16x8 hb[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) hb[s] = *reinterpret_cast<const bf16x8 *>(h + trow * TL_D + s * 32 + g * 8);

  // ---- y1 = Wl h + bl + x ; D[channel][token]; Wl streamed through LDS in 4 chunks of 64 output rows.
  // Register-staged double buffering: the next chunk's global loads are issued before the MFMAs of the
  // current chunk and written to LDS after them.
  constexpr int CH_WL = 64 * TL_LDW;
  uint4 pre[4];
  auto wl_load = [&](int qc) {  // 64 rows x 32 uint4 = 2048 uint4 over 512 threads
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * TL_THREADS, r = e >> 5, c8 = e & 31;
      pre[i] = *reinterpret_cast<const uint4 *>(Wl + (size_t)(qc * 64 + r) * TL_D + c8 * 8);
    }
  };
  auto wl_store = [&](int buf) {
    u16 *d = sW + buf * CH_WL;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * TL_THREADS, r = e >> 5, c8 = e & 31;
      *reinterpret_cast<uint4 *>(d + r * TL_LDW + c8 * 8) = pre[i];
    }
  };
  wl_load(0);
  wl_store(0);
  __syncthreads();
  f32x4 y[16];
#pragma unroll
  for (int qc = 0; qc < 4; ++qc) {
    if (qc + 1 < 4) wl_load(qc + 1);
    const u16 *cW = sW + (qc & 1) * CH_WL;
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      const int t = qc * 4 + tt;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const u16 *wr = cW + (tt * 16 + tk) * TL_LDW + g * 8;  // A: row = out channel t*16 + (lane&15)
#pragma unroll
      for (int s = 0; s < 8; ++s)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8 *>(wr + s * 32), hb[s], acc, 0,
                                                      0, 0);
      // C/D: lane (token tk, group g) register r <-> channel t*16 + 4g + r
      const float4 bv = *reinterpret_cast<const float4 *>(bl + t * 16 + 4 * g);
      const uint2 xv = *reinterpret_cast<const uint2 *>(x + trow * TL_D + t * 16 + 4 * g);
      acc[0] += bv.x + tl_bf2f((u16)(xv.x & 0xFFFF));
      acc[1] += bv.y + tl_bf2f((u16)(xv.x >> 16));
      acc[2] += bv.z + tl_bf2f((u16)(xv.y & 0xFFFF));
      acc[3] += bv.w + tl_bf2f((u16)(xv.y >> 16));
      y[t] = acc;
    }
    if (qc + 1 < 4) wl_store((qc + 1) & 1);
    __syncthreads();
  }
  tl_ln(y, ln1w, ln1b, g, eps);
  // r (bf16) as B-operand fragments of the expand GEMM: k-step s <- tiles 2s, 2s+1.  The fp32 copy is
  // dropped here (the second residual re-reads these bf16 values), which frees 64 VGPRs.
  bf16x8 rb[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) rb[s] = tl_pack(y[2 * s], y[2 * s + 1]);

  // ---- FFN in 16 chunks of 32 hidden channels; chunk = We rows (32x256) + Ws columns (256x32) = 32 KB
  f32x4 acc2[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) acc2[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int CH_WE = 32 * TL_LDW;             // u16 elements
  constexpr int CH_WS = TL_D * TL_LDS2;
  constexpr int CH = CH_WE + CH_WS;
  auto ff_load = [&](int c) {  // We: 32x32 uint4 = 1024 (2 per thread); Ws: 256x4 = 1024 (2 per thread)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + i * TL_THREADS, r = e >> 5, c8 = e & 31;
      pre[i] = *reinterpret_cast<const uint4 *>(We + (size_t)(c * 32 + r) * TL_D + c8 * 8);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + i * TL_THREADS, r = e >> 2, c8 = e & 3;
      pre[2 + i] = *reinterpret_cast<const uint4 *>(Ws + (size_t)r * TL_H + c * 32 + c8 * 8);
    }
  };
  auto ff_store = [&](int buf) {
    u16 *dWe = sW + buf * CH, *dWs = dWe + CH_WE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + i * TL_THREADS, r = e >> 5, c8 = e & 31;
      *reinterpret_cast<uint4 *>(dWe + r * TL_LDW + c8 * 8) = pre[i];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + i * TL_THREADS, r = e >> 2, c8 = e & 3;
      *reinterpret_cast<uint4 *>(dWs + r * TL_LDS2 + c8 * 8) = pre[2 + i];
    }
  };
  ff_load(0);
  ff_store(0);
  __syncthreads();
  for (int c = 0; c < 16; ++c) {
    const int buf = c & 1;
    if (c + 1 < 16) ff_load(c + 1);
    const u16 *cWe = sW + buf * CH, *cWs = cWe + CH_WE;
    // e = relu(We[chunk] r + be): two 16-channel tiles, K = 256 in the permuted order
    f32x4 e0 = {0.f, 0.f, 0.f, 0.f}, e1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      e0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tl_wfrag(cWe + tk * TL_LDW + s * 32, g), rb[s], e0, 0, 0, 0);
      e1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tl_wfrag(cWe + (16 + tk) * TL_LDW + s * 32, g), rb[s], e1, 0, 0, 0);
    }
    const float4 b0 = *reinterpret_cast<const float4 *>(be + c * 32 + 4 * g);
    const float4 b1 = *reinterpret_cast<const float4 *>(be + c * 32 + 16 + 4 * g);
    e0[0] = fmaxf(e0[0] + b0.x, 0.f); e0[1] = fmaxf(e0[1] + b0.y, 0.f);
    e0[2] = fmaxf(e0[2] + b0.z, 0.f); e0[3] = fmaxf(e0[3] + b0.w, 0.f);
    e1[0] = fmaxf(e1[0] + b1.x, 0.f); e1[1] = fmaxf(e1[1] + b1.y, 0.f);
    e1[2] = fmaxf(e1[2] + b1.z, 0.f); e1[3] = fmaxf(e1[3] + b1.w, 0.f);
    const bf16x8 eb = tl_pack(e0, e1);  // one k-step (32 hidden channels) of the squeeze GEMM
#pragma unroll
    for (int t = 0; t < 16; ++t)
      acc2[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tl_wfrag(cWs + (t * 16 + tk) * TL_LDS2, g), eb, acc2[t], 0, 0,
                                                        0);
    if (c + 1 < 16) ff_store(buf ^ 1);
    __syncthreads();
  }
  // ---- out = LN2(r + y2 + bs), r re-read from its bf16 fragments
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const float4 bv = *reinterpret_cast<const float4 *>(bs + t * 16 + 4 * g);
    union { bf16x8 v; u16 u[8]; } rf;
    rf.v = rb[t >> 1];
    const int o = (t & 1) * 4;
    y[t][0] = tl_bf2f(rf.u[o + 0]) + acc2[t][0] + bv.x;
    y[t][1] = tl_bf2f(rf.u[o + 1]) + acc2[t][1] + bv.y;
    y[t][2] = tl_bf2f(rf.u[o + 2]) + acc2[t][2] + bv.z;
    y[t][3] = tl_bf2f(rf.u[o + 3]) + acc2[t][3] + bv.w;
  }
  tl_ln(y, ln2w, ln2b, g, eps);
  if (row0 + tk < rows) {
    u16 *o = out + (row0 + tk) * TL_D + 4 * g;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      uint2 v;
      v.x = (uint32_t)tl_f2bf(y[t][0]) | ((uint32_t)tl_f2bf(y[t][1]) << 16);
      v.y = (uint32_t)tl_f2bf(y[t][2]) | ((uint32_t)tl_f2bf(y[t][3]) << 16);
      *reinterpret_cast<uint2 *>(o + t * 16) = v;
    }
  }
}

}  // namespace unopose

using namespace unopose;

extern "C" {

int unopose_transformer_tail(const void *h, const void *x, long rows, const void *Wl, const float *bl,
                             const float *ln1w, const float *ln1b, const void *We, const float *be, const void *Ws,
                             const float *bs, const float *ln2w, const float *ln2b, float eps, void *out,
                             unopose_stream_t stream) {
  UNOPOSE_REQUIRE(h && x && Wl && bl && ln1w && ln1b && We && be && Ws && bs && ln2w && ln2b && out,
                  "transformer_tail: null pointer");
  UNOPOSE_REQUIRE(rows >= 0, "transformer_tail: bad sizes");
  if (rows == 0) return UNOPOSE_OK;
  const size_t lds_wl = (size_t)2 * 64 * TL_LDW * 2;
  const size_t lds_ch = (size_t)2 * (32 * TL_LDW + TL_D * TL_LDS2) * 2;
  const size_t lds = lds_wl > lds_ch ? lds_wl : lds_ch;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void *)transformer_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    attr_set = true;
  }
  dim3 grid((unsigned)((rows + TL_WAVES * 16 - 1) / (TL_WAVES * 16)));
  hipLaunchKernelGGL(transformer_tail_kernel, grid, dim3(TL_THREADS), lds, (hipStream_t)stream, (const u16 *)h,
                     (const u16 *)x, rows, (const u16 *)Wl, bl, ln1w, ln1b, (const u16 *)We, be, (const u16 *)Ws, bs,
                     ln2w, ln2b, eps, (u16 *)out);
  return check_launch("transformer_tail");
}

}  // extern "C"
