// DINOv2 ViT patch attention for gfx950 (C ABI part 2).
//
// Replaces timm's Attention core, softmax(q k^T / sqrt(64)) v per head, that the reference drives at
// core/unopose/model/oneref_feature_extraction.py:38-41 (12 heads x 64, T = 5 + (S/14)^2 tokens).
// Flash-style: the T x T score matrix never exists in memory.  One wavefront owns 32 query tokens of one
// (image, head) and walks the keys in tiles of 32 with an online softmax:
//   * S^T = K Q^T on v_mfma_f32_32x32x16_bf16 ("swapped" product): in the C/D layout a lane then holds
//     16 keys of ONE query, so the softmax reductions are in-register plus one cross-half exchange;
//   * O^T += V^T P^T with P fed straight from those registers: accumulator registers 8s..8s+7 are one
//     16-key MFMA k-step in a fixed permuted key order, and V^T (channel-major, built once per call) is
//     read in the same order, so P never moves between lanes;
//   * O^T is transposed through 4 KiB of LDS so every token row is stored as 128 contiguous bytes.
#include "common.h"

namespace unopose {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

__device__ __forceinline__ u16 va_f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (u16)(u >> 16);
}

// qkv: (B, T, 3, H, 64) bf16 (the fused qkv Linear output); vt: (B, H, 64, TP) bf16, TP % 32 == 0,
// zero beyond T; out: (B, T, H*64) bf16.
__global__ __launch_bounds__(256) void vit_attn_kernel(const u16 *__restrict__ qkv, const u16 *__restrict__ vt,
                                                       int T, int TP, int H, float scale_log2e,
                                                       u16 *__restrict__ out) {
  __shared__ __attribute__((aligned(16))) u16 Ot[4][32][72];  // per wave: 32 tokens x 64 channels (+8 pad)
  const int b = blockIdx.z, h = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q0 = (blockIdx.x * 4 + wave) * 32;
  if (q0 >= T) return;
  const int col = lane & 31, hb = lane >> 5;
  const int C3 = 3 * H * 64;
  const u16 *base = qkv + (size_t)b * T * C3;
  // Q as the B operand: lane (query col, hb) holds 8 consecutive channels per k-step
  bf16x8 qf[4];
  {
    const int tq = min(q0 + col, T - 1);
    const u16 *qp = base + (size_t)tq * C3 + h * 64 + hb * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8 *>(qp + ks * 16);
  }
  f32x16 o[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m_run = -3e38f, l_run = 0.f;
  const u16 *VT = vt + ((size_t)b * H + h) * 64 * TP;

  for (int k0 = 0; k0 < T; k0 += 32) {
    // ---- S^T tile: rows = 32 keys, cols = 32 queries
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    {
      const int tk = min(k0 + col, T - 1);
      const u16 *kp = base + (size_t)tk * C3 + (H + h) * 64 + hb * 8;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8 *>(kp + ks * 16);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
      }
    }
    // ---- online softmax for this lane's query over its 16 keys (+ the other half-wave's 16)
    float mx = -3e38f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * hb;
      s[r] = key < T ? s[r] * scale_log2e : -3e38f;
      mx = fmaxf(mx, s[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = exp2f(m_run - m_new);
    float ls = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = exp2f(s[r] - m_new);
      ls += s[r];
    }
    ls += __shfl_xor(ls, 32);
    l_run = l_run * alpha + ls;
    m_run = m_new;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
    // ---- O^T += V^T P^T ; k-step s2 = accumulator registers 8*s2 .. 8*s2+7 (permuted key order)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      union { bf16x8 v; u16 u[8]; } pf;
#pragma unroll
      for (int e = 0; e < 8; ++e) pf.u[e] = va_f2bf(s[s2 * 8 + e]);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        // A = V^T: row = channel t*32 + col, keys k0 + 16 s2 + 4 hb + {0..3} and + 8 + {0..3}
        const u16 *vp = VT + (size_t)(t * 32 + col) * TP + k0 + s2 * 16 + 4 * hb;
        union { bf16x8 v; bf16x4 h4[2]; } vf;
        vf.h4[0] = *reinterpret_cast<const bf16x4 *>(vp);
        vf.h4[1] = *reinterpret_cast<const bf16x4 *>(vp + 8);
        o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.v, pf.v, o[t], 0, 0, 0);
      }
    }
  }
  // ---- normalise, transpose through LDS, store token rows
  const float inv = 1.f / l_run;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hb;
      Ot[wave][col][c] = va_f2bf(o[t][r] * inv);
    }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // 32 rows x 128 B: each lane moves 16 B; 8 lanes cover one row
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = it * 8 + (lane >> 3), seg = lane & 7;
    const int tq = q0 + row;
    if (tq < T) {
      const uint4 v = *reinterpret_cast<const uint4 *>(&Ot[wave][row][seg * 8]);
      *reinterpret_cast<uint4 *>(out + ((size_t)b * T + tq) * (H * 64) + h * 64 + seg * 8) = v;
    }
  }
}

}  // namespace unopose

using namespace unopose;

extern "C" {

int unopose_vit_attention(const void *qkv, const void *vt, int B, int T, int TP, int H, void *out,
                          unopose_stream_t stream) {
  UNOPOSE_REQUIRE(qkv && vt && out, "vit_attention: null pointer");
  UNOPOSE_REQUIRE(B >= 0 && T >= 1 && H >= 1 && TP >= T && TP % 32 == 0 && B <= 65535 && H <= 65535,
                  "vit_attention: bad sizes (TP must be a multiple of 32 >= T)");
  if (B == 0) return UNOPOSE_OK;
  const float scale_log2e = 0.125f * 1.4426950408889634f;  // 1/sqrt(64) * log2(e)
  dim3 grid(cdiv(T, 128), H, B);
  hipLaunchKernelGGL(vit_attn_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const u16 *)qkv, (const u16 *)vt, T,
                     TP, H, scale_log2e, (u16 *)out);
  return check_launch("vit_attention");
}

}  // extern "C"
