// Experiment: csrc/vit_attn.hip's 64-queries-per-wave kernel with K / V staged by LDS-DMA (no staging registers) into a 3-deep ring,
// the K rows XOR-swizzled on the SOURCE address, and the score MFMAs of tile t + 1 issued before the softmax of tile t.
// Same arithmetic as the product kernel (swapped product, deferred reference point, end-of-tile fix-up).
#include "common.h"
#include "gemm_common.h"
#include <cstdlib>

namespace unopose {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u16 va_f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (u16)(u >> 16);
}

constexpr int VA_CHUNK = 128;
constexpr int VA_KBYTES = VA_CHUNK * 128;      // K chunk: [key][64 ch] bf16, 16-byte chunk c of row r at c ^ ((r >> 1) & 7)
constexpr int VA_VSUBB = 128 * 32 + 128;       // bytes per V sub-tile [128 keys][16 channels] + bank skew
constexpr int VA_BUFB = VA_KBYTES + 4 * VA_VSUBB;  // 33280 B
constexpr float VA_DEFER = 8.f;
#ifndef VA_FILL
#define VA_FILL 6
#endif
#ifndef VA_MODE
#define VA_MODE 1  // 1: two score sets, the score MFMAs of tile t + 1 before the softmax of tile t; 2: one score set, fragment prefetch
#endif
#ifndef VA_ABL
#define VA_ABL 0  // timing ablations of VA_MODE 2 (wrong results): 1 no score MFMAs, 2 no P.V MFMAs, 4 no exp, 8 no per-chunk barrier
#endif
#ifndef VA_QB
#define VA_QB 2  // 32-query blocks per wave
#endif
#ifndef VA_NW
#define VA_NW 8  // wavefronts per workgroup (one workgroup per CU: NW / 4 waves per SIMD)
#endif
#ifndef VA_ONES
#define VA_ONES 0  // 1: row sums through the matrix pipe (modes 0 / 1)
#endif
#ifndef VA_NBUF
#define VA_NBUF 3  // chunk buffers in LDS (2: the next chunk is in flight during the current one; 3: two chunks ahead)
#endif
#ifndef VA_SCHED
#define VA_SCHED 1
#endif
#define VA_IL1 __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, VA_FILL, 0);
#define VA_INTERLEAVE VA_IL1 VA_IL1 VA_IL1 VA_IL1 VA_IL1 VA_IL1 VA_IL1 VA_IL1 VA_IL1 VA_IL1 VA_IL1 VA_IL1 VA_IL1 VA_IL1 VA_IL1 VA_IL1

template <int QB, int NW, int NBUF>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : (NW == 2 ? 4 : 1)) void vit_attn2_kernel(const u16 *__restrict__ qkv, int T, int H, int BH, int nq,
                                                                                                      float scale_log2e, u16 *__restrict__ out) {
  constexpr int NPC = 32 / NW;  // LDS-DMA pieces per wave and chunk: NPC / 2 of K, NPC / 2 of V
  static_assert(NW == 2 || NW == 4 || NW == 8 || NW == 16, "2, 4, 8 or 16 wavefronts");
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  u16 (*Ot)[32][72] = reinterpret_cast<u16 (*)[32][72]>(smem);
  static_assert(NW * 32 * 72 * 2 * (VA_QB == 2 ? 1 : 1) <= NBUF * VA_BUFB, "output staging must fit the chunk buffers");
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int bh = (slot / nq) * 8 + xcd, qblk = slot % nq;
  if (bh >= BH) return;
  const int b = bh / H, h = bh % H;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q0 = (qblk * NW + wave) * (32 * QB);
  const bool active = q0 < T;
  const int col = lane & 31, hb = lane >> 5;
  const int C3 = 3 * H * 64;
  const u16 *base = qkv + (size_t)b * T * C3;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;

  // ---- LDS-DMA sources of this wave's pieces (2 of K, 2 of V per chunk): byte offsets from the image's first token; rows past
  //      the image fall outside the descriptor and arrive as zeros
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)((size_t)T * C3 * 2), 0x00020000);
  uint32_t kvo[NPC / 2], vvo[NPC / 2], kdst[NPC / 2], vdst[NPC / 2];
#pragma unroll
  for (int i = 0; i < NPC / 2; ++i) {
    const int p = (NPC / 2) * wave + i;
    const int krow = 8 * p + (lane >> 3), gch = (lane & 7) ^ ((krow >> 1) & 7);
    kvo[i] = (uint32_t)(krow * C3 * 2 + (H * 64 + h * 64) * 2 + gch * 16);
    kdst[i] = (uint32_t)(p * 1024);
    const int sub = p >> 2, vkey = 32 * (p & 3) + (lane >> 1);
    vvo[i] = (uint32_t)(vkey * C3 * 2 + (2 * H * 64 + h * 64 + sub * 16 + (lane & 1) * 8) * 2);
    vdst[i] = (uint32_t)(VA_KBYTES + sub * VA_VSUBB + (p & 3) * 1024);
  }
  auto issue_chunk = [&](int c, int bslot) {
    const uint32_t b0 = lds0 + bslot * VA_BUFB, ro = (uint32_t)(c * VA_CHUNK) * (uint32_t)(C3 * 2);
#pragma unroll
    for (int i = 0; i < NPC / 2; ++i) gemm_dma16(b0 + kdst[i], kvo[i] + ro, rs, 0);
#pragma unroll
    for (int i = 0; i < NPC / 2; ++i) gemm_dma16(b0 + vdst[i], vvo[i] + ro, rs, 0);
  };
  const int nchunks = (T + VA_CHUNK - 1) / VA_CHUNK;
  issue_chunk(0, 0);
  if (NBUF == 3 && nchunks > 1) issue_chunk(1, 1);

  bf16x8 qf[QB][4];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int tq = min(q0 + qb * 32 + col, T - 1);
    const u16 *qp = base + (size_t)tq * C3 + h * 64 + hb * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[qb][ks] = *reinterpret_cast<const bf16x8 *>(qp + ks * 16);
  }
#pragma unroll
  for (int qb = 0; qb < QB; ++qb)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qf[qb][ks]));
  f32x16 o[QB][2];
  float m_run[QB], l_run[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    m_run[qb] = -3e38f;
    l_run[qb] = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[qb][t][r] = 0.f;
  }
  f32x16 lacc[QB];
  union { bf16x8 v; uint32_t w[4]; } ones;
#pragma unroll
  for (int e = 0; e < 4; ++e) ones.w[e] = 0x3f803f80u;
#pragma unroll
  for (int qb = 0; qb < QB; ++qb)
#pragma unroll
    for (int r = 0; r < 16; ++r) lacc[qb][r] = 0.f;
  // K fragment of key row kt + col, k-step ks: 16-byte chunk 2 ks + hb, swizzled by the row (kt is a multiple of 32)
  uint32_t koff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) koff[ks] = (uint32_t)(col * 128 + (((2 * ks + hb) ^ ((col >> 1) & 7)) << 4));
  const uint32_t vlane_off = (uint32_t)(VA_KBYTES + ((lane >> 4) & 1) * VA_VSUBB + (4 * hb + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8);

  auto qk_tile = [&](const char *buf, int kt, f32x16 (&s)[QB]) {
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[qb][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 kf = *reinterpret_cast<const bf16x8 *>(buf + kt * 128 + koff[ks]);
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) s[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[qb][ks], s[qb], 0, 0, 0);
    }
  };
  // VA_ONES: the row sums ride the matrix pipe -- lacc += ONES . P^T (every row of the 32 x 32 result is the column sum of the bf16 P the
  // P.V product also uses), so the softmax loses its 16 adds, the cross-half exchange of the sum and the per-tile alpha; alpha is only
  // computed in the rare re-reference path, where lacc is fixed up exactly like O.
  auto softmax_pv_tile = [&](const char *buf, int kt, f32x16 (&s)[QB]) {
    const char *vlane = buf + vlane_off;
    union PF { bf16x8 v; uint32_t w[4]; } pf[QB][2];
    float alpha[QB], m_old[QB];
    bool moved = false;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      float mx = s[qb][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[qb][r]);
      const auto sm = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      mx = fmaxf(__uint_as_float(sm[0]), __uint_as_float(sm[1])) * scale_log2e;
      const bool grow = mx > m_run[qb] + VA_DEFER;
      const float m_use = grow ? mx : m_run[qb];
      m_old[qb] = m_run[qb];
      if (!VA_ONES) alpha[qb] = __builtin_amdgcn_exp2f(m_run[qb] - m_use);
      m_run[qb] = m_use;
      moved |= grow;
      float ls = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[qb][r] = __builtin_amdgcn_exp2f(fmaf(s[qb][r], scale_log2e, -m_use));
        if (!VA_ONES) ls += s[qb][r];
      }
      if (!VA_ONES) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(ls), __float_as_uint(ls), false, false);
        l_run[qb] = fmaf(l_run[qb], alpha[qb], __uint_as_float(sw[0]) + __uint_as_float(sw[1]));
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int e = 0; e < 4; ++e) pf[qb][s2].w[e] = cvt_pk_bf16_f32(s[qb][s2 * 8 + 2 * e], s[qb][s2 * 8 + 2 * e + 1]);
    }
    auto v_frag = [&](int s2, int t) {
      const char *vp = vlane + t * 2 * VA_VSUBB + (kt + s2 * 16) * 32;
      union { bf16x8 v; s16x4 h4[2]; } vf;
      vf.h4[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(vp));
      vf.h4[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(vp + 8 * 32));
      return vf.v;
    };
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bf16x8 vf = v_frag(s2, t);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) o[qb][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[qb][s2].v, o[qb][t], 0, 0, 0);
      }
      if (VA_ONES) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) lacc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones.v, pf[qb][s2].v, lacc[qb], 0, 0, 0);
      }
    }
    if (__builtin_expect(__any(moved), 0)) {
      if (VA_ONES) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
          alpha[qb] = __builtin_amdgcn_exp2f(m_old[qb] - m_run[qb]);
          f32x16 d;
#pragma unroll
          for (int r = 0; r < 16; ++r) d[r] = 0.f;
          d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones.v, pf[qb][0].v, d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones.v, pf[qb][1].v, d, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 16; ++r) lacc[qb][r] = fmaf(lacc[qb][r] - d[r], alpha[qb], d[r]);
        }
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bf16x8 v0 = v_frag(0, t), v1 = v_frag(1, t);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
          f32x16 d;
#pragma unroll
          for (int r = 0; r < 16; ++r) d[r] = 0.f;
          d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pf[qb][0].v, d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pf[qb][1].v, d, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 16; ++r) o[qb][t][r] = fmaf(o[qb][t][r] - d[r], alpha[qb], d[r]);
        }
      }
    }
  };

  // ---- VA_MODE 2: one score set; the K fragments of tile t + 1 and the V fragments of tile t are read into registers at the TOP of
  //      tile t (behind the score MFMAs' issue), a whole softmax ahead of their use
  auto load_kf = [&](const char *buf, int kt, bf16x8 (&kf)[4]) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kf[ks] = *reinterpret_cast<const bf16x8 *>(buf + kt * 128 + koff[ks]);
  };
  auto tile_pref = [&](const char *buf, int kt, bf16x8 (&kc)[4], bf16x8 (&kn)[4], bool pre_next, bool mask, int c0) {
    const char *vlane = buf + vlane_off;
    union VF { bf16x8 v; s16x4 h4[2]; } vf[2][2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const char *vp = vlane + t * 2 * VA_VSUBB + (kt + s2 * 16) * 32;
        vf[s2][t].h4[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(vp));
        vf[s2][t].h4[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(vp + 8 * 32));
      }
    f32x16 s[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[qb][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (VA_ABL & 1) {  // no score MFMAs: lane-varying stand-ins keep the softmax alive
#pragma unroll
          for (int r = 0; r < 4; ++r) s[qb][4 * ks + r] = __uint_as_float(((const uint32_t *)&kc[ks])[r] & 0x3f7fffffu);
        } else {
          s[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kc[ks], qf[qb][ks], s[qb], 0, 0, 0);
        }
      }
    }
    if (pre_next) load_kf(buf, kt + 32, kn);
    __builtin_amdgcn_sched_barrier(0);
    if (mask) {
#pragma unroll
      for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (c0 + kt + (r & 3) + 8 * (r >> 2) + 4 * hb >= T) s[qb][r] = -3e38f;
    }
    union PF { bf16x8 v; uint32_t w[4]; } pf[QB][2];
    float alpha[QB];
    bool moved = false;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      float mx = s[qb][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[qb][r]);
      const auto sm = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      mx = fmaxf(__uint_as_float(sm[0]), __uint_as_float(sm[1])) * scale_log2e;
      const bool grow = mx > m_run[qb] + VA_DEFER;
      const float m_use = grow ? mx : m_run[qb];
      alpha[qb] = __builtin_amdgcn_exp2f(m_run[qb] - m_use);
      m_run[qb] = m_use;
      moved |= grow;
      float ls = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[qb][r] = (VA_ABL & 4) ? fmaf(s[qb][r], scale_log2e, -m_use) : __builtin_amdgcn_exp2f(fmaf(s[qb][r], scale_log2e, -m_use));
        ls += s[qb][r];
      }
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(ls), __float_as_uint(ls), false, false);
      l_run[qb] = fmaf(l_run[qb], alpha[qb], __uint_as_float(sw[0]) + __uint_as_float(sw[1]));
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int e = 0; e < 4; ++e) pf[qb][s2].w[e] = cvt_pk_bf16_f32(s[qb][s2 * 8 + 2 * e], s[qb][s2 * 8 + 2 * e + 1]);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          if (VA_ABL & 2) {  // no P.V MFMAs: fold P and V into the accumulator with two VALU ops so that neither is dead
            o[qb][t][0] += __uint_as_float(pf[qb][s2].w[0] ^ pf[qb][s2].w[1] ^ pf[qb][s2].w[2] ^ pf[qb][s2].w[3]);
            o[qb][t][1] += __uint_as_float(((const uint32_t *)&vf[s2][t].v)[0] ^ ((const uint32_t *)&vf[s2][t].v)[1] ^ ((const uint32_t *)&vf[s2][t].v)[2] ^ ((const uint32_t *)&vf[s2][t].v)[3]);
          } else {
            o[qb][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[s2][t].v, pf[qb][s2].v, o[qb][t], 0, 0, 0);
          }
        }
    }
    if (__builtin_expect(__any(moved), 0)) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
          f32x16 d;
#pragma unroll
          for (int r = 0; r < 16; ++r) d[r] = 0.f;
          d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0][t].v, pf[qb][0].v, d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[1][t].v, pf[qb][1].v, d, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 16; ++r) o[qb][t][r] = fmaf(o[qb][t][r] - d[r], alpha[qb], d[r]);
        }
    }
  };
  auto chunk_compute2 = [&](int c0, const char *buf) {
    const int nk = min(VA_CHUNK, T - c0);
    const int nt = (nk + 31) >> 5;  // tiles of this chunk; only the sequence's last one can be partial
    bf16x8 ka[4], kb[4];
    load_kf(buf, 0, ka);
    for (int t = 0; t < nt; t += 2) {
      tile_pref(buf, t * 32, ka, kb, t + 1 < nt, (t + 1) * 32 > nk, c0);
      if (t + 1 < nt) tile_pref(buf, (t + 1) * 32, kb, ka, t + 2 < nt, (t + 2) * 32 > nk, c0);
    }
  };

  auto chunk_compute = [&](int c0, const char *buf) {
    if (VA_MODE == 2) return chunk_compute2(c0, buf);
    const int nk = min(VA_CHUNK, T - c0);
    int kt = 0;
    if (VA_MODE == 0) {
      for (; kt + 32 <= nk; kt += 32) {
        f32x16 s[QB];
        qk_tile(buf, kt, s);
        softmax_pv_tile(buf, kt, s);
      }
    } else if (nk >= 32) {
      f32x16 sa[QB], sb[QB];
      qk_tile(buf, 0, sa);
      for (;;) {
        const bool more_b = kt + 64 <= nk;
        if (more_b) qk_tile(buf, kt + 32, sb);
        softmax_pv_tile(buf, kt, sa);
        if (VA_SCHED) { VA_INTERLEAVE }
        kt += 32;
        if (!more_b) break;
        const bool more_a = kt + 64 <= nk;
        if (more_a) qk_tile(buf, kt + 32, sa);
        softmax_pv_tile(buf, kt, sb);
        if (VA_SCHED) { VA_INTERLEAVE }
        kt += 32;
        if (!more_a) break;
      }
    }
    if (kt < nk) {
      f32x16 s[QB];
      qk_tile(buf, kt, s);
#pragma unroll
      for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (c0 + kt + (r & 3) + 8 * (r >> 2) + 4 * hb >= T) s[qb][r] = -3e38f;
      softmax_pv_tile(buf, kt, s);
    }
  };

  // chunk 0 landed (chunk 1 may still fly), every wave's pieces visible
  auto wait_all_but_one_chunk = [&]() {
    if (NPC == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (NPC == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (NPC == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  };
  if (NBUF == 3 && nchunks > 1) wait_all_but_one_chunk();
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  int bslot = 0;
  if (NBUF == 1) {  // one chunk buffer: load, wait, compute -- the CU's other workgroups compute meanwhile
    for (int c = 0; c < nchunks; ++c) {
      if (c > 0) {
        issue_chunk(c, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      if (active) chunk_compute(c * VA_CHUNK, smem);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  } else
  for (int c = 0; c < nchunks; ++c) {
    int nslot = bslot + NBUF - 1;
    if (nslot >= NBUF) nslot -= NBUF;
    if (c + NBUF - 1 < nchunks) issue_chunk(c + NBUF - 1, nslot);  // (that buffer was last read in chunk c - 1, before the barrier that ended it)
    if (active) chunk_compute(c * VA_CHUNK, smem + bslot * VA_BUFB);
    if (c + 1 < nchunks) {
      if (NBUF == 3 && c + 2 < nchunks) wait_all_but_one_chunk();
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!(VA_ABL & 8)) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    bslot = bslot + 1 == NBUF ? 0 : bslot + 1;
  }
  if (!active) return;
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const float inv = 1.f / (VA_ONES ? lacc[qb][0] : l_run[qb]);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ch = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hb;
        Ot[wave][col][ch] = va_f2bf(o[qb][t][r] * inv);
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + (lane >> 3), seg = lane & 7;
      const int tq = q0 + qb * 32 + row;
      if (tq < T) {
        const uint4 v = *reinterpret_cast<const uint4 *>(&Ot[wave][row][seg * 8]);
        *reinterpret_cast<uint4 *>(out + ((size_t)b * T + tq) * (H * 64) + h * 64 + seg * 8) = v;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

}  // namespace unopose

using namespace unopose;

extern "C" int unopose_vit_attention(const void *qkv, int B, int T, int H, void *out, unopose_stream_t stream) {
  constexpr int QB = VA_QB, NW = VA_NW, NBUF = VA_NBUF;
  static bool opt[64];
  const size_t lds = (size_t)NBUF * VA_BUFB;
  if (lds_optin(opt, reinterpret_cast<const void *>(&vit_attn2_kernel<QB, NW, NBUF>), lds, "vit_attention") != UNOPOSE_OK) return UNOPOSE_ELAUNCH;
  const int BH = B * H, nq = cdiv(T, 32 * NW * QB);
  const float scale_log2e = 0.125f * 1.4426950408889634f;
  const long blocks = (long)cdiv(BH, 8) * nq * 8;
  hipLaunchKernelGGL((vit_attn2_kernel<QB, NW, NBUF>), dim3((unsigned)blocks), dim3(NW * 64), lds, (hipStream_t)stream, (const u16 *)qkv, T, H, BH, nq, scale_log2e,
                     (u16 *)out);
  return check_launch("vit_attention");
}
