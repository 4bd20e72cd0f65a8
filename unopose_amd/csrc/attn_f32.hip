// fp32-class attention kernels for gfx950 (C ABI part 2): the SAME algorithms as attn.hip (token
// attention of the correspondence transformer) and vit_attn.hip (ViT patch attention), for float32
// inputs / outputs, so that the fp32 configuration the reference runs by default
// (configs/main_cfg.py:87-89, test.amp.enabled=False) -- and all golden parity tests -- also go through
// hand-written HIP.  gfx950 has no reduced-precision fast path for fp32 matrix inputs (no xf32), and its
// exact fp32 MFMA runs at 1/16 of the bf16 rate; instead every operand is split on the fly into
// hi + lo bfloat16 parts and each product is issued as 3 bf16 MFMAs (a_hi b_hi + a_hi b_lo + a_lo b_hi,
// fp32 accumulation): ~2^-16 relative error per product, i.e. fp32-class results at 5x the fp32 MFMA rate.
#include "common.h"

namespace unopose {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint32_t af_cvt_pk(float a, float b) {
  return cvt_pk_bf16_f32(a, b);
}
struct HL {
  bf16x8 hi, lo;
};
__device__ __forceinline__ HL af_split(const float (&v)[8]) {
  union { bf16x8 v; uint32_t w[4]; } H, L;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const uint32_t h = af_cvt_pk(v[2 * e], v[2 * e + 1]);
    H.w[e] = h;
    L.w[e] = af_cvt_pk(v[2 * e] - __uint_as_float(h << 16), v[2 * e + 1] - __uint_as_float(h & 0xFFFF0000u));
  }
  return HL{H.v, L.v};
}
__device__ __forceinline__ HL af_load8(const float *p) {  // 8 consecutive fp32 (32-byte aligned)
  const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + 4);
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  return af_split(v);
}
__device__ __forceinline__ HL af_zero() {
  float z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  return af_split(z);
}
#define AF_MFMA3_16(acc, A, B)                                              \
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A.hi, B.hi, acc, 0, 0, 0);  \
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A.hi, B.lo, acc, 0, 0, 0);  \
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A.lo, B.hi, acc, 0, 0, 0)
#define AF_MFMA3_32(acc, A, B)                                              \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.hi, B.hi, acc, 0, 0, 0);  \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.hi, B.lo, acc, 0, 0, 0);  \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.lo, B.hi, acc, 0, 0, 0)

template <int XOR>
__device__ __forceinline__ float af_swz(float v) {
  return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), (XOR << 10) | 0x1F));
}

constexpr int AF_NT = 14, AF_MP = AF_NT * 16;
#ifndef AF_PF_DEPTH
#define AF_PF_DEPTH 4  // operand pairs in flight ahead of the MFMAs of the token attention
#endif

// ---- token attention (see attn.hip for the scheme): one wave = 4 query rows x 4 heads ----------------
template <bool RPE>
__global__ __launch_bounds__(256) void token_attn_f32_kernel(const float *__restrict__ q, int ldq,
                                                             const float *__restrict__ k, int ldk,
                                                             const float *__restrict__ vt,
                                                             const float *__restrict__ qp, int ldqp,
                                                             const float *__restrict__ E, int n, int m, float scale,
                                                             float *__restrict__ out) {
  __shared__ __attribute__((aligned(16))) float Pl[4][16][AF_MP + 4];
  const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = (blockIdx.x * 4 + wave) * 4;
  if (n0 >= n) return;
  const int li = lane & 15, kg = lane >> 4;
  const int a_nl = li >> 2, a_h = li & 3;
  const bool a_valid = n0 + a_nl < n;
  const float *Q = q + ((size_t)b * n + n0 + a_nl) * ldq;
  const float *K = k + (size_t)b * m * ldk;
  f32x4 acc[AF_NT];
#pragma unroll
  for (int t = 0; t < AF_NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nt_valid = (m + 15) >> 4;
  // Operand streams straight from global memory (K is L2-resident, the geometric embedding E is a 2.5 GB HBM stream per launch): the
  // raw 32-byte pieces of the next AF_PF (k-step, key tile) pairs are in flight while the current one is split and multiplied -- the
  // compiler had issued every pair of loads directly in front of its three MFMAs (one exposed memory round trip per 3 MFMAs).
  struct Raw {
    float4 a, b;
  };
  auto ld8 = [](const float *p) { return Raw{*reinterpret_cast<const float4 *>(p), *reinterpret_cast<const float4 *>(p + 4)}; };
  auto sp8 = [](const Raw &r) {
    const float v[8] = {r.a.x, r.a.y, r.a.z, r.a.w, r.b.x, r.b.y, r.b.z, r.b.w};
    return af_split(v);
  };
  constexpr int AF_PF = AF_PF_DEPTH, NPAIR = 8 * AF_NT;
  {
    auto addr = [&](int i) { return K + (size_t)min((i % AF_NT) * 16 + li, m - 1) * ldk + (i / AF_NT) * 32 + kg * 8; };
    Raw ring[AF_PF];
#pragma unroll
    for (int i = 0; i < AF_PF; ++i) ring[i] = ld8(addr(i));
    HL a = af_zero();
#pragma unroll
    for (int i = 0; i < NPAIR; ++i) {
      const int ks = i / AF_NT, t = i % AF_NT;
      if (t == 0) {
        const int kk = ks * 32 + kg * 8;
        a = af_zero();
        if (a_valid && (kk >> 6) == a_h) a = af_load8(Q + kk);
      }
      const Raw cur = ring[i % AF_PF];
      if (i + AF_PF < NPAIR) ring[i % AF_PF] = ld8(addr(i + AF_PF));
      if (t < nt_valid) {
        const HL bv = sp8(cur);
        AF_MFMA3_16(acc[t], a, bv);
      }
    }
  }
  if (RPE) {
    for (int nl = 0; nl < 4; ++nl) {
      if (n0 + nl >= n) break;
      const float *QP = qp + ((size_t)b * n + n0 + nl) * ldqp + a_h * 256;
      const float *En = E + ((size_t)b * n + n0 + nl) * (size_t)m * 256;
      auto addr = [&](int i) { return En + (size_t)min((i % AF_NT) * 16 + li, m - 1) * 256 + (i / AF_NT) * 32 + kg * 8; };
      Raw ring[AF_PF];
#pragma unroll
      for (int i = 0; i < AF_PF; ++i) ring[i] = ld8(addr(i));
      Raw araw = ld8(QP + kg * 8);  // the query-side fragment of k-step 0; the next one is fetched a whole k-step ahead
      HL a = af_zero();
#pragma unroll
      for (int i = 0; i < NPAIR; ++i) {
        const int ks = i / AF_NT, t = i % AF_NT;
        if (t == 0) {
          a = af_zero();
          if (a_nl == nl) a = sp8(araw);
          if (ks + 1 < 8) araw = ld8(QP + (ks + 1) * 32 + kg * 8);
        }
        const Raw cur = ring[i % AF_PF];
        if (i + AF_PF < NPAIR) ring[i % AF_PF] = ld8(addr(i + AF_PF));
        if (t < nt_valid) {
          const HL bv = sp8(cur);
          AF_MFMA3_16(acc[t], a, bv);
        }
      }
    }
  }
  float mx[4] = {-3e38f, -3e38f, -3e38f, -3e38f};
#pragma unroll
  for (int t = 0; t < AF_NT; ++t) {
    const bool ok = t * 16 + li < m;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      acc[t][h] = ok ? acc[t][h] * scale : -3e38f;
      mx[h] = fmaxf(mx[h], acc[t][h]);
    }
  }
  float sm[4];
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    float v = mx[h];
    v = fmaxf(v, af_swz<1>(v)); v = fmaxf(v, af_swz<2>(v)); v = fmaxf(v, af_swz<4>(v)); v = fmaxf(v, af_swz<8>(v));
    mx[h] = v;
    sm[h] = 0.f;
  }
#pragma unroll
  for (int t = 0; t < AF_NT; ++t) {
    const bool ok = t * 16 + li < m;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const float p = ok ? expf(acc[t][h] - mx[h]) : 0.f;
      acc[t][h] = p;
      sm[h] += p;
    }
  }
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    float v = sm[h];
    v += af_swz<1>(v); v += af_swz<2>(v); v += af_swz<4>(v); v += af_swz<8>(v);
    sm[h] = 1.f / v;
  }
#pragma unroll
  for (int t = 0; t < AF_NT; ++t)
#pragma unroll
    for (int h = 0; h < 4; ++h) Pl[wave][kg * 4 + h][t * 16 + li] = acc[t][h] * sm[h];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  HL pa[AF_MP / 32];
#pragma unroll
  for (int ks = 0; ks < AF_MP / 32; ++ks) pa[ks] = af_load8(&Pl[wave][li][ks * 32 + kg * 8]);
  const float *VT = vt + (size_t)b * 256 * AF_MP;
  {
    constexpr int NKS = AF_MP / 32, NP2 = 16 * NKS;
    auto addr = [&](int i) { return VT + (size_t)((i / NKS) * 16 + li) * AF_MP + kg * 8 + (i % NKS) * 32; };
    Raw ring[AF_PF];
#pragma unroll
    for (int i = 0; i < AF_PF; ++i) ring[i] = ld8(addr(i));
    f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NP2; ++i) {
      const int nt = i / NKS, ks = i % NKS;
      if (ks == 0) o = f32x4{0.f, 0.f, 0.f, 0.f};
      const Raw cur = ring[i % AF_PF];
      if (i + AF_PF < NP2) ring[i % AF_PF] = ld8(addr(i + AF_PF));
      const HL bv = sp8(cur);
      AF_MFMA3_16(o, pa[ks], bv);
      if (ks == NKS - 1 && n0 + kg < n) out[((size_t)b * n + n0 + kg) * 256 + nt * 16 + li] = o[nt >> 2];
    }
  }
}

// ---- ViT attention (see vit_attn.hip): flash-style; K / V^T chunks in LDS ALREADY SPLIT into bf16 hi / lo planes ---------------
// (the staging threads split every element once per workgroup; round 2 kept fp32 in LDS and every wave re-split every fragment at
// every use -- ~190 of the ~340 vector instructions of a 32-key tile.  Same hi / lo values, so the results are bit-identical.)
constexpr int AV_CHUNK = 64, AV_LDH = 64 + 8, AV_LDVH = AV_CHUNK + 8;  // u16 row strides of the K planes / the V^T planes
typedef unsigned short u16f;

// SPLIT: the output is written in the split layout of csrc/gemm_f32.hip (per token and 32-channel block one 128-byte line
// [hi (32 bf16) | lo (32 bf16)]) -- the operand form of the projection GEMM that follows, instead of fp32 + a split pass.
template <bool SPLIT>
__global__ __launch_bounds__(256) void vit_attn_f32_kernel(const float *__restrict__ qkv, int T, int H, float scale_log2e,
                                                           float *__restrict__ out) {
  // K hi | K lo | V^T hi | V^T lo planes (4 x 9 KiB); the same memory is re-used as the output transpose buffer at the end
  constexpr int PLANE_K = AV_CHUNK * AV_LDH, PLANE_V = 64 * AV_LDVH;
  __shared__ __attribute__((aligned(16))) u16f smem16[2 * PLANE_K + 2 * PLANE_V];
  u16f *Kh = smem16, *Kl = smem16 + PLANE_K, *Vh = smem16 + 2 * PLANE_K, *Vl = Vh + PLANE_V;
  float (*Ot)[32][68] = reinterpret_cast<float (*)[32][68]>(smem16);
  static_assert(4 * 32 * 68 * 4 <= (2 * PLANE_K + 2 * PLANE_V) * 2, "output staging must fit the chunk buffers");
  const int b = blockIdx.z, h = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q0 = (blockIdx.x * 4 + wave) * 32;
  const bool active = q0 < T;
  const int col = lane & 31, hb = lane >> 5;
  const int C3 = 3 * H * 64;
  const float *base = qkv + (size_t)b * T * C3;
  HL qf[4];
  {
    const float *qp = base + (size_t)min(q0 + col, T - 1) * C3 + h * 64 + hb * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = af_load8(qp + ks * 16);
  }
  f32x16 o[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m_run = -3e38f, l_run = 0.f;
  for (int c0 = 0; c0 < T; c0 += AV_CHUNK) {
    __syncthreads();
    // 64 keys x 64 channels fp32 for K and V: 1024 float4 each, 4 per thread; split once here
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * 256, key = e >> 4, c4 = e & 15;
      const float *src = base + (size_t)min(c0 + key, T - 1) * C3 + h * 64 + c4 * 4;
      const float4 kv = *reinterpret_cast<const float4 *>(src + H * 64);
      float4 vv = *reinterpret_cast<const float4 *>(src + 2 * H * 64);
      if (c0 + key >= T) vv = make_float4(0.f, 0.f, 0.f, 0.f);
      uint2 kh, kl;
      kh.x = af_cvt_pk(kv.x, kv.y);
      kh.y = af_cvt_pk(kv.z, kv.w);
      kl.x = af_cvt_pk(kv.x - __uint_as_float(kh.x << 16), kv.y - __uint_as_float(kh.x & 0xFFFF0000u));
      kl.y = af_cvt_pk(kv.z - __uint_as_float(kh.y << 16), kv.w - __uint_as_float(kh.y & 0xFFFF0000u));
      *reinterpret_cast<uint2 *>(Kh + key * AV_LDH + c4 * 4) = kh;
      *reinterpret_cast<uint2 *>(Kl + key * AV_LDH + c4 * 4) = kl;
      const uint32_t vh0 = af_cvt_pk(vv.x, vv.y), vh1 = af_cvt_pk(vv.z, vv.w);
      const uint32_t vl0 = af_cvt_pk(vv.x - __uint_as_float(vh0 << 16), vv.y - __uint_as_float(vh0 & 0xFFFF0000u));
      const uint32_t vl1 = af_cvt_pk(vv.z - __uint_as_float(vh1 << 16), vv.w - __uint_as_float(vh1 & 0xFFFF0000u));
      Vh[(c4 * 4 + 0) * AV_LDVH + key] = (u16f)(vh0 & 0xFFFF);
      Vh[(c4 * 4 + 1) * AV_LDVH + key] = (u16f)(vh0 >> 16);
      Vh[(c4 * 4 + 2) * AV_LDVH + key] = (u16f)(vh1 & 0xFFFF);
      Vh[(c4 * 4 + 3) * AV_LDVH + key] = (u16f)(vh1 >> 16);
      Vl[(c4 * 4 + 0) * AV_LDVH + key] = (u16f)(vl0 & 0xFFFF);
      Vl[(c4 * 4 + 1) * AV_LDVH + key] = (u16f)(vl0 >> 16);
      Vl[(c4 * 4 + 2) * AV_LDVH + key] = (u16f)(vl1 & 0xFFFF);
      Vl[(c4 * 4 + 3) * AV_LDVH + key] = (u16f)(vl1 >> 16);
    }
    __syncthreads();
    if (!active) continue;
    const int nk = min(AV_CHUNK, T - c0);
    for (int kt = 0; kt < nk; kt += 32) {
      const int k0 = c0 + kt;
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int off = (kt + col) * AV_LDH + ks * 16 + hb * 8;
        const HL kf{*reinterpret_cast<const bf16x8 *>(Kh + off), *reinterpret_cast<const bf16x8 *>(Kl + off)};
        AF_MFMA3_32(s, kf, qf[ks]);
      }
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
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        float pv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) pv[e] = s[s2 * 8 + e];
        const HL pf = af_split(pv);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          // keys kt + 16 s2 + 4 hb + {0..3} and + 8 + {0..3}: two 8-byte reads per plane
          const int off = (t * 32 + col) * AV_LDVH + kt + s2 * 16 + 4 * hb;
          union { bf16x8 v; uint2 h2[2]; } VH, VL;
          VH.h2[0] = *reinterpret_cast<const uint2 *>(Vh + off);
          VH.h2[1] = *reinterpret_cast<const uint2 *>(Vh + off + 8);
          VL.h2[0] = *reinterpret_cast<const uint2 *>(Vl + off);
          VL.h2[1] = *reinterpret_cast<const uint2 *>(Vl + off + 8);
          const HL vf{VH.v, VL.v};
          AF_MFMA3_32(o[t], vf, pf);
        }
      }
    }
  }
  __syncthreads();  // every wave is done with the K / V chunk before it becomes the output buffer
  if (!active) return;
  const float inv = 1.f / l_run;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) Ot[wave][col][t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hb] = o[t][r] * inv;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // 32 rows x 256 B: 16 lanes per row, 16 B each
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int row = it * 4 + (lane >> 4), seg = lane & 15;
    if (q0 + row < T) {
      const float4 v = *reinterpret_cast<const float4 *>(&Ot[wave][row][seg * 4]);
      if (SPLIT) {
        const int c = h * 64 + seg * 4;
        uint2 hi, lo;
        hi.x = cvt_pk_bf16_f32(v.x, v.y);
        hi.y = cvt_pk_bf16_f32(v.z, v.w);
        lo.x = cvt_pk_bf16_f32(v.x - __uint_as_float(hi.x << 16), v.y - __uint_as_float(hi.x & 0xffff0000u));
        lo.y = cvt_pk_bf16_f32(v.z - __uint_as_float(hi.y << 16), v.w - __uint_as_float(hi.y & 0xffff0000u));
        char *line = reinterpret_cast<char *>(out) + ((size_t)b * T + q0 + row) * (size_t)(H * 64) * 4 + (size_t)(c >> 5) * 128 + (c & 31) * 2;
        *reinterpret_cast<uint2 *>(line) = hi;
        *reinterpret_cast<uint2 *>(line + 64) = lo;
      } else {
        *reinterpret_cast<float4 *>(out + ((size_t)b * T + q0 + row) * (H * 64) + h * 64 + seg * 4) = v;
      }
    }
  }
}

}  // namespace unopose

using namespace unopose;

extern "C" {

int unopose_token_attention_f32(const float *q, int ldq, const float *k, int ldk, const float *vt, const float *qp,
                                int ldqp, const float *E, int B, int n, int m, float scale, float *out,
                                unopose_stream_t stream) {
  UNOPOSE_REQUIRE(q && k && vt && out, "token_attention_f32: null pointer");
  UNOPOSE_REQUIRE((qp == nullptr) == (E == nullptr), "token_attention_f32: qp and E go together");
  UNOPOSE_REQUIRE(B >= 0 && n >= 1 && m >= 1 && m <= AF_MP && B <= 65535,
                  "token_attention_f32: m=%d exceeds the %d-key tile", m, AF_MP);
  UNOPOSE_REQUIRE(ldq >= 256 && ldk >= 256 && ldq % 8 == 0 && ldk % 8 == 0 && (!E || (ldqp >= 1024 && ldqp % 8 == 0)),
                  "token_attention_f32: row strides must be multiples of 8 elements");
  if (B == 0) return UNOPOSE_OK;
  dim3 grid(cdiv(n, 16), B);
  hipStream_t s = (hipStream_t)stream;
  if (E)
    hipLaunchKernelGGL(token_attn_f32_kernel<true>, grid, dim3(256), 0, s, q, ldq, k, ldk, vt, qp, ldqp, E, n, m, scale,
                       out);
  else
    hipLaunchKernelGGL(token_attn_f32_kernel<false>, grid, dim3(256), 0, s, q, ldq, k, ldk, vt, (const float *)nullptr,
                       0, (const float *)nullptr, n, m, scale, out);
  return check_launch("token_attention_f32");
}

int unopose_vit_attention_f32(const float *qkv, int B, int T, int H, float *out, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(qkv && out, "vit_attention_f32: null pointer");
  UNOPOSE_REQUIRE(B >= 0 && T >= 1 && H >= 1 && B <= 65535 && H <= 65535, "vit_attention_f32: bad sizes");
  if (B == 0) return UNOPOSE_OK;
  dim3 grid(cdiv(T, 128), H, B);
  hipLaunchKernelGGL(vit_attn_f32_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, qkv, T, H,
                     0.125f * 1.4426950408889634f, out);
  return check_launch("vit_attention_f32");
}

int unopose_vit_attention_f32_split(const float *qkv, int B, int T, int H, void *out_split, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(qkv && out_split, "vit_attention_f32_split: null pointer");
  UNOPOSE_REQUIRE(B >= 0 && T >= 1 && H >= 1 && B <= 65535 && H <= 65535, "vit_attention_f32_split: bad sizes");
  if (B == 0) return UNOPOSE_OK;
  dim3 grid(cdiv(T, 128), H, B);
  hipLaunchKernelGGL(vit_attn_f32_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, qkv, T, H, 0.125f * 1.4426950408889634f,
                     (float *)out_split);
  return check_launch("vit_attention_f32_split");
}

}  // extern "C"
