// DINOv2 ViT patch attention for gfx950 (C ABI part 2).
//
// Replaces timm's Attention core, softmax(q k^T / sqrt(64)) v per head, that the reference drives at
// core/unopose/model/oneref_feature_extraction.py:38-41 (12 heads x 64, T = 5 + (S/14)^2 tokens).
// Flash-style: the T x T score matrix never exists in memory.  One wavefront owns 32 query tokens of one
// (image, head) and walks the keys in tiles of 32 with an online softmax:
//   * S^T = K Q^T on v_mfma_f32_32x32x16_bf16 ("swapped" product): in the C/D layout a lane then holds
//     16 keys of ONE query, so the softmax reductions are in-register plus one cross-half exchange;
//   * O^T += V^T P^T with P fed straight from those registers: accumulator registers 8s..8s+7 are one
//     16-key MFMA k-step in a fixed permuted key order, and V^T (transposed on its way into LDS) is
//     read in the same order, so P never moves between lanes;
//   * K / V of a 128-key chunk are staged once per workgroup in LDS and shared by its 4 waves;
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

constexpr int VA_CHUNK = 128;            // keys staged per LDS chunk
constexpr int VA_LDK = 64 + 8;           // padded row of the K chunk  [key][channel]
constexpr int VA_LDV = VA_CHUNK + 8;     // padded row of the V^T chunk [channel][key]

// qkv: (B, T, 3, H, 64) bf16 (the fused qkv Linear output); out: (B, T, H*64) bf16.
// The 4 waves of a workgroup share one (image, head): K and V of a 128-key chunk are loaded once with
// coalesced 16-byte reads, K kept row-major and V transposed on the way into LDS (so the caller needs no
// V^T copy), then every wave runs its 32 queries against the chunk.
__device__ __forceinline__ uint32_t va_cvt_pk(float a, float b) {  // packed RNE fp32 -> bf16 (gfx950)
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4))) void vit_attn_kernel(const u16 *__restrict__ qkv, int T, int H, float scale_log2e,
                                                       u16 *__restrict__ out) {
  // K chunk [key][ch] + V^T chunk [ch][key]; the same bytes serve as the output transpose buffer at the end
  __shared__ __attribute__((aligned(16))) u16 smem[VA_CHUNK * VA_LDK + 64 * VA_LDV];
  u16 (*Ks)[VA_LDK] = reinterpret_cast<u16 (*)[VA_LDK]>(smem);
  u16 (*Vs)[VA_LDV] = reinterpret_cast<u16 (*)[VA_LDV]>(smem + VA_CHUNK * VA_LDK);
  u16 (*Ot)[32][72] = reinterpret_cast<u16 (*)[32][72]>(smem);
  static_assert(4 * 32 * 72 <= VA_CHUNK * VA_LDK + 64 * VA_LDV, "output staging must fit the chunk buffers");
  const int b = blockIdx.z, h = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q0 = (blockIdx.x * 4 + wave) * 32;
  const bool active = q0 < T;  // inactive waves still help staging and hit every barrier
  const int col = lane & 31, hb = lane >> 5;
  const int C3 = 3 * H * 64;
  const u16 *base = qkv + (size_t)b * T * C3;
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
  // running max in the exp2 domain (score * scale * log2 e) and running sum
  float m_run = -3e38f, l_run = 0.f;

  // register-staged prefetch of a chunk: 128 keys x 128 B of K and of V, 8 lanes per key row (coalesced)
  uint4 pk[4], pv[4];
  auto chunk_load = [&](int c0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * 256, key = e >> 3, c8 = e & 7;
      const u16 *src = base + (size_t)min(c0 + key, T - 1) * C3 + h * 64 + c8 * 8;
      pk[i] = *reinterpret_cast<const uint4 *>(src + H * 64);
      pv[i] = *reinterpret_cast<const uint4 *>(src + 2 * H * 64);
    }
  };
  auto chunk_store = [&](int c0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * 256, key = e >> 3, c8 = e & 7;
      *reinterpret_cast<uint4 *>(&Ks[key][c8 * 8]) = pk[i];
      union { uint4 v; u16 u[8]; } vv;
      vv.v = pv[i];
      const bool valid = c0 + key < T;  // keys beyond T contribute V = 0 (their P is 0 as well)
#pragma unroll
      for (int j = 0; j < 8; ++j) Vs[c8 * 8 + j][key] = valid ? vv.u[j] : (u16)0;
    }
  };
  chunk_load(0);
  for (int c0 = 0; c0 < T; c0 += VA_CHUNK) {
    __syncthreads();  // previous chunk fully consumed
    chunk_store(c0);
    if (c0 + VA_CHUNK < T) chunk_load(c0 + VA_CHUNK);  // in flight under this chunk's MFMAs
    __syncthreads();
    if (!active) continue;
    const int nk = min(VA_CHUNK, T - c0);
    for (int kt = 0; kt < nk; kt += 32) {
      const int k0 = c0 + kt;
      // ---- S^T tile: rows = 32 keys, cols = 32 queries
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8 *>(&Ks[kt + col][ks * 16 + hb * 8]);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
      }
      if (k0 + 32 > T) {  // only the last tile can hold keys >= T (wave-uniform branch)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (k0 + (r & 3) + 8 * (r >> 2) + 4 * hb >= T) s[r] = -3e38f;
      }
      // ---- online softmax for this lane's query over its 16 keys (+ the other half-wave's 16)
      float mx = s[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32)) * scale_log2e;  // scale > 0: max commutes with it
      const float m_new = fmaxf(m_run, mx);
      float ls = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], scale_log2e, -m_new));
        ls += s[r];
      }
      ls += __shfl_xor(ls, 32);
      if (__any(m_new > m_run)) {  // rescale the accumulators only when some query's max moved
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        l_run *= alpha;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
        m_run = m_new;
      }
      l_run += ls;
      // ---- O^T += V^T P^T ; k-step s2 = accumulator registers 8*s2 .. 8*s2+7 (permuted key order)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        union { bf16x8 v; uint32_t w[4]; } pf;
#pragma unroll
        for (int e = 0; e < 4; ++e) pf.w[e] = va_cvt_pk(s[s2 * 8 + 2 * e], s[s2 * 8 + 2 * e + 1]);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          // A = V^T: row = channel t*32 + col, keys kt + 16 s2 + 4 hb + {0..3} and + 8 + {0..3}
          const u16 *vp = &Vs[t * 32 + col][kt + s2 * 16 + 4 * hb];
          union { bf16x8 v; bf16x4 h4[2]; } vf;
          vf.h4[0] = *reinterpret_cast<const bf16x4 *>(vp);
          vf.h4[1] = *reinterpret_cast<const bf16x4 *>(vp + 8);
          o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.v, pf.v, o[t], 0, 0, 0);
        }
      }
    }
  }
  __syncthreads();  // every wave is done with the K / V chunk before it becomes the output buffer
  if (!active) return;
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

int unopose_vit_attention(const void *qkv, int B, int T, int H, void *out, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(qkv && out, "vit_attention: null pointer");
  UNOPOSE_REQUIRE(B >= 0 && T >= 1 && H >= 1 && B <= 65535 && H <= 65535, "vit_attention: bad sizes");
  if (B == 0) return UNOPOSE_OK;
  const float scale_log2e = 0.125f * 1.4426950408889634f;  // 1/sqrt(64) * log2(e)
  dim3 grid(cdiv(T, 128), H, B);
  hipLaunchKernelGGL(vit_attn_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const u16 *)qkv, T, H, scale_log2e,
                     (u16 *)out);
  return check_launch("vit_attention");
}

}  // extern "C"
