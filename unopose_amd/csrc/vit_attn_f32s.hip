// DINOv2 ViT patch attention on fp32-class data for gfx950 (C ABI part 2): the reference's default precision
// (configs/main_cfg.py:87-89, no autocast; timm Attention core at core/unopose/model/oneref_feature_extraction.py:38-41).
//
// Input AND output are in the SPLIT layout of csrc/gemm_f32.hip (per token and 32-channel block one 128-byte line
// [hi (32 bf16) | lo (32 bf16)], x = hi + lo to 2^-17): the qkv GEMM writes it in its epilogue and the projection GEMM reads it, so
// this kernel never sees fp32 data and never splits an operand itself -- round 3's kernel (attn_f32.hip) read fp32 qkv and every one
// of the 11 workgroups of an (image, head) re-split the whole K / V on its way into LDS, V^T through 2-byte LDS stores.
// Structure: 12 waves x 32 queries per workgroup (3 waves per SIMD); K / V of a 128-key chunk (hi and lo planes, 65 KiB) arrive in
// double-buffered LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`, 64 pieces of 1 KiB per chunk, per-lane source addresses pick the planes
// apart; the K rows are XOR-swizzled on the SOURCE address, the V image is the [sub-tile][key][16 channels] form ds_read_b64_tr_b16
// wants) -- no staging registers, no LDS stores, one counted wait + barrier per chunk; S^T = K Q^T with the operands swapped so that a
// lane holds 16 keys of ONE query, deferred reference point + end-of-tile fix-up (vit_attn.hip), P from registers (split into hi / lo
// there), three MFMAs per product (hi.hi + hi.lo + lo.hi, fp32 accumulation) in both contractions.
// Round 4: 1293 us (8 waves, register-staged chunks) -> 1086 us at 64 images x 12 heads x 1374 tokens, same box.
#include "common.h"
#include "gemm_common.h"

namespace unopose {

typedef short s16x4 __attribute__((ext_vector_type(4)));

#ifndef VS_CHUNK_KEYS
#define VS_CHUNK_KEYS 128
#endif
constexpr int VS_CHUNK = VS_CHUNK_KEYS;  // keys staged per LDS chunk (128, or 64: half the LDS, two workgroups per CU)
constexpr int VS_KPLANE = VS_CHUNK * 128;  // bytes of a K plane [key][64 channels] bf16; 16-byte chunk c of row r sits at c ^ ((r >> 1) & 7)
constexpr int VS_VSUBB = VS_CHUNK * 32 + 128;   // bytes per V sub-tile [128 keys][16 channels] (+ bank skew), 4 per plane
constexpr int VS_VPLANE = 4 * VS_VSUBB;
constexpr int VS_VOFF = 2 * VS_KPLANE;     // K hi | K lo | V hi | V lo
constexpr int VS_BUFB = 2 * VS_KPLANE + 2 * VS_VPLANE;  // 66560 B
constexpr float VS_DEFER = 8.f;  // log2 of the largest P the deferred rescale lets through
#ifndef VS_MODE
#define VS_MODE 0  // 0: score MFMAs, softmax, P.V per tile; 1: the score MFMAs of tile t + 1 issued before the softmax of tile t (measured: slower at 3 waves / SIMD)
#endif
#ifndef VS_BATCH_READS
#define VS_BATCH_READS 2  // 1: all fragments of a product read before its MFMAs; 2: the V fragments already before the softmax arithmetic (1101 -> 1086 -> 1079 us)
#endif
#ifndef VS_NW
#define VS_NW 12   // wavefronts per workgroup = 3 per SIMD (154 VGPRs); 8: 1189 us, 12: 1086 us at 64 x 12 x 1374 (scripts/ubench/vit_attn_var.py)
#endif

struct HLf {
  bf16x8 h, l;
};
#define VS_MFMA3(acc, a, b)                                          \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, b.h, acc, 0, 0, 0); \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.l, acc, 0, 0, 0); \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, acc, 0, 0, 0)

// qkv_s: (B, T) token rows of 3 * H * 64 values in the split layout (3 * H * 256 bytes per token: q blocks | k blocks | v blocks,
// head h = blocks 2h, 2h + 1 of each third); out_s: (B, T) rows of H * 64 values in the split layout.
template <int NW>
__global__ __launch_bounds__(NW * 64, VS_CHUNK == 64 ? 2 : 1) void vit_attn_f32s_kernel(const char *__restrict__ qkv_s, int T, int H,
                                                                                                     int BH, int nq, float scale_log2e,
                                                                                                     char *__restrict__ out_s) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  float (*Ot)[32][68] = reinterpret_cast<float (*)[32][68]>(smem);
  static_assert(NW * 32 * 68 * 4 <= 2 * VS_BUFB, "output staging must fit the chunk buffers");
  constexpr int NT = NW * 64;
  static_assert(NT == 384 || NT == 512 || NT == 768 || NT == 1024, "6, 8, 12 or 16 wavefronts per workgroup");
  // workgroup -> (image, head, query block): all query blocks of one (image, head) share id % 8 (one XCD's L2 holds its K / V)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int bh = (slot / nq) * 8 + xcd, qblk = slot % nq;
  if (bh >= BH) return;
  const int b = bh / H, h = bh % H;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q0 = (qblk * NW + wave) * 32;
  const bool active = q0 < T;  // inactive waves still help staging and hit every barrier
  const int col = lane & 31, hb = lane >> 5;
  const size_t RB = (size_t)3 * H * 256;  // bytes per token row
  const char *base = qkv_s + (size_t)b * T * RB;
  const int qoff = h * 256, koff = (H + h) * 256, voff = (2 * H + h) * 256;  // the head's 2 blocks = 256 contiguous bytes [hi0 lo0 hi1 lo1]
  // ---- LDS-DMA staging: per chunk 32 pieces of K (2 planes x 16 groups of 8 key rows) and 32 of V (2 planes x 4 sub-tiles x 4 groups of
  //      32 keys), 1 KiB each; wave w issues K pieces 4w .. 4w + 3 and V pieces 4w .. 4w + 3.  Rows past the image arrive as zeros.
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)((size_t)T * RB), 0x00020000);
  constexpr int NPC = VS_CHUNK / 4, KP = VS_CHUNK / 8, VG = VS_CHUNK / 32;  // pieces of K (and of V) per chunk; K pieces / V key groups per plane
  constexpr int NP = (NPC + NW - 1) / NW;  // pieces of K (and of V) per wave: piece ids wave, wave + NW, ... < NPC
  uint32_t kvo[NP], vvo[NP], kdst[NP], vdst[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int kp = min(wave + i * NW, NPC - 1), p = kp / KP, rg = kp % KP;
    const int krow = 8 * rg + (lane >> 3), gch = (lane & 7) ^ ((krow >> 1) & 7);
    kvo[i] = (uint32_t)(krow * (int)RB + koff + (gch >> 2) * 128 + (gch & 3) * 16 + p * 64);
    kdst[i] = (uint32_t)(p * VS_KPLANE + rg * 1024);
    const int pv = kp / (4 * VG), sub = (kp / VG) & 3, kq = kp % VG, vkey = 32 * kq + (lane >> 1), ch = sub * 16 + (lane & 1) * 8;
    vvo[i] = (uint32_t)(vkey * (int)RB + voff + (ch >> 5) * 128 + (ch & 31) * 2 + pv * 64);
    vdst[i] = (uint32_t)(VS_VOFF + pv * VS_VPLANE + sub * VS_VSUBB + kq * 1024);
  }
  auto issue_chunk = [&](int c, int bslot) {
    const uint32_t b0 = lds0 + bslot * VS_BUFB, ro = (uint32_t)(c * VS_CHUNK) * (uint32_t)RB;
#pragma unroll
    for (int i = 0; i < NP; ++i)
      if (wave + i * NW < NPC) gemm_dma16(b0 + kdst[i], kvo[i] + ro, rs, 0);  // (wave-uniform)
#pragma unroll
    for (int i = 0; i < NP; ++i)
      if (wave + i * NW < NPC) gemm_dma16(b0 + vdst[i], vvo[i] + ro, rs, 0);
  };
  const int nchunks = (T + VS_CHUNK - 1) / VS_CHUNK;
  issue_chunk(0, 0);
  HLf qf[4];
  {
    const char *qp = base + (size_t)min(q0 + col, T - 1) * RB + qoff;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int c = ks * 16 + hb * 8;  // channels c .. c + 7
      const char *p = qp + (c >> 5) * 128 + (c & 31) * 2;
      qf[ks].h = *reinterpret_cast<const bf16x8 *>(p);
      qf[ks].l = *reinterpret_cast<const bf16x8 *>(p + 64);
    }
  }
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qf[ks].h), "+v"(qf[ks].l));  // the Q loads are complete HERE (see vit_attn.hip)
  f32x16 o[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m_run = -3e38f, l_run = 0.f;

  // this lane's slot in the transpose read: sub-tile (lane>>4)&1, key row 4 hb + ((lane&15)>>2), 8-byte chunk lane&3
  const uint32_t vlane_off = (uint32_t)(VS_VOFF + ((lane >> 4) & 1) * VS_VSUBB + (4 * hb + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8);
  uint32_t kfo[4];  // K fragment of key row kt + col, k-step ks: 16-byte chunk 2 ks + hb, swizzled by the row
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) kfo[ks] = (uint32_t)(col * 128 + (((2 * ks + hb) ^ ((col >> 1) & 7)) << 4));

  auto qk_tile = [&](const char *buf, int c0, int kt, bool partial, f32x16 &s) {
    // S^T = K Q^T: rows = 32 keys, cols = 32 queries
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#if VS_BATCH_READS
    // all eight K fragments first, then the twelve MFMAs (the compiler otherwise reads every k-step's pair right in front of its MFMAs:
    // four exposed LDS round trips instead of one)
    HLf kf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const char *kp = buf + kt * 128 + kfo[ks];
      kf[ks].h = *reinterpret_cast<const bf16x8 *>(kp);
      kf[ks].l = *reinterpret_cast<const bf16x8 *>(kp + VS_KPLANE);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { VS_MFMA3(s, kf[ks], qf[ks]); }
#else
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const char *kp = buf + kt * 128 + kfo[ks];
      const HLf kf{*reinterpret_cast<const bf16x8 *>(kp), *reinterpret_cast<const bf16x8 *>(kp + VS_KPLANE)};
      VS_MFMA3(s, kf, qf[ks]);
    }
#endif
    if (partial) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (c0 + kt + (r & 3) + 8 * (r >> 2) + 4 * hb >= T) s[r] = -3e38f;
    }
  };
  auto softmax_pv_tile = [&](const char *buf, int kt, f32x16 &s) {
#if VS_BATCH_READS == 2
    HLf vfe[2][2];  // (experiment: the V fragments read before the softmax arithmetic)
    {
      const char *vl0 = buf + vlane_off;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const char *vp = vl0 + t * 2 * VS_VSUBB + (kt + s2 * 16) * 32;
          union { bf16x8 v; s16x4 h4[2]; } vh, vl;
          vh.h4[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(vp));
          vh.h4[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(vp + 8 * 32));
          vl.h4[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(vp + VS_VPLANE));
          vl.h4[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(vp + VS_VPLANE + 8 * 32));
          vfe[s2][t] = HLf{vh.v, vl.v};
        }
      __builtin_amdgcn_sched_barrier(0);
    }
#endif
    // online softmax with the deferred reference point (vit_attn.hip): per-lane select, no branch before the P.V MFMAs
    float mx = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
    const auto sm = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
    mx = fmaxf(__uint_as_float(sm[0]), __uint_as_float(sm[1])) * scale_log2e;
    const bool grow = mx > m_run + VS_DEFER;
    const float m_use = grow ? mx : m_run;
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
    m_run = m_use;
    float ls = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], scale_log2e, -m_use));
      ls += s[r];
    }
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(ls), __float_as_uint(ls), false, false);
    l_run = fmaf(l_run, alpha, __uint_as_float(sw[0]) + __uint_as_float(sw[1]));
    // P = hi + lo, packed as the B operand of the two 16-key k-steps
    union PF {
      bf16x8 v;
      uint32_t w[4];
    } ph[2], pl[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a = s[s2 * 8 + 2 * e], c = s[s2 * 8 + 2 * e + 1];
        const uint32_t hw = cvt_pk_bf16_f32(a, c);
        ph[s2].w[e] = hw;
        pl[s2].w[e] = cvt_pk_bf16_f32(a - __uint_as_float(hw << 16), c - __uint_as_float(hw & 0xffff0000u));
      }
    const char *vlane = buf + vlane_off;
    auto v_frag = [&](int s2, int t) {
      const char *vp = vlane + t * 2 * VS_VSUBB + (kt + s2 * 16) * 32;
      union {
        bf16x8 v;
        s16x4 h4[2];
      } vh, vl;
      vh.h4[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(vp));
      vh.h4[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(vp + 8 * 32));
      vl.h4[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(vp + VS_VPLANE));
      vl.h4[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(vp + VS_VPLANE + 8 * 32));
      return HLf{vh.v, vl.v};
    };
#if VS_BATCH_READS
    HLf vfa[2][2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#if VS_BATCH_READS == 2
        vfa[s2][t] = vfe[s2][t];
#else
        vfa[s2][t] = v_frag(s2, t);
#endif
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const HLf pf{ph[s2].v, pl[s2].v};
        VS_MFMA3(o[t], vfa[s2][t], pf);
      }
#else
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const HLf vf = v_frag(s2, t);
        const HLf pf{ph[s2].v, pl[s2].v};
        VS_MFMA3(o[t], vf, pf);
      }
#endif
    if (__builtin_expect(__any(grow), 0)) {  // rare after the first tile: O = (O - D) alpha + D with D = this tile's P.V (vit_attn.hip)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x16 d;
#pragma unroll
        for (int r = 0; r < 16; ++r) d[r] = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const HLf vf = v_frag(s2, t);
          const HLf pf{ph[s2].v, pl[s2].v};
          VS_MFMA3(d, vf, pf);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = fmaf(o[t][r] - d[r], alpha, d[r]);
      }
    }
  };
  auto chunk_compute = [&](int c0, const char *buf) {
    const int nk = min(VS_CHUNK, T - c0);
    const int nt = (nk + 31) >> 5;  // tiles of this chunk; only the sequence's last one can be partial
    if (VS_MODE == 0) {
      for (int t = 0; t < nt; ++t) {
        f32x16 s;
        qk_tile(buf, c0, t * 32, (t + 1) * 32 > nk, s);
        softmax_pv_tile(buf, t * 32, s);
      }
    } else {
      f32x16 sa, sb;
      qk_tile(buf, c0, 0, 32 > nk, sa);
      for (int t = 0; t < nt; t += 2) {
        if (t + 1 < nt) qk_tile(buf, c0, (t + 1) * 32, (t + 2) * 32 > nk, sb);
        softmax_pv_tile(buf, t * 32, sa);
        if (t + 1 < nt) {
          if (t + 2 < nt) qk_tile(buf, c0, (t + 2) * 32, (t + 3) * 32 > nk, sa);
          softmax_pv_tile(buf, (t + 1) * 32, sb);
        }
      }
    }
  };

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  for (int c = 0; c < nchunks; ++c) {
    if (c + 1 < nchunks) issue_chunk(c + 1, (c + 1) & 1);  // (that buffer was last read in chunk c - 1, before the barrier that ended it)
    if (active) chunk_compute(c * VS_CHUNK, smem + (c & 1) * VS_BUFB);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  if (!active) return;
  // ---- normalise, transpose through LDS, store token rows in the split layout
  const float inv = 1.f / l_run;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) Ot[wave][col][t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hb] = o[t][r] * inv;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // 32 rows x 64 channels: 16 lanes per row, 4 channels each -> 8 bytes of hi and 8 bytes of lo
  const size_t ORB = (size_t)H * 256;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int row = it * 4 + (lane >> 4), seg = lane & 15;
    if (q0 + row < T) {
      const float4 v = *reinterpret_cast<const float4 *>(&Ot[wave][row][seg * 4]);
      uint2 hi, lo;
      hi.x = cvt_pk_bf16_f32(v.x, v.y);
      hi.y = cvt_pk_bf16_f32(v.z, v.w);
      lo.x = cvt_pk_bf16_f32(v.x - __uint_as_float(hi.x << 16), v.y - __uint_as_float(hi.x & 0xffff0000u));
      lo.y = cvt_pk_bf16_f32(v.z - __uint_as_float(hi.y << 16), v.w - __uint_as_float(hi.y & 0xffff0000u));
      const int c = seg * 4;  // channel inside the head
      char *line = out_s + ((size_t)b * T + q0 + row) * ORB + h * 256 + (c >> 5) * 128 + (c & 31) * 2;
      *reinterpret_cast<uint2 *>(line) = hi;
      *reinterpret_cast<uint2 *>(line + 64) = lo;
    }
  }
}

}  // namespace unopose

using namespace unopose;

extern "C" {

int unopose_vit_attention_f32_ss(const void *qkv_split, int B, int T, int H, void *out_split, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(qkv_split && out_split, "vit_attention_f32_ss: null pointer");
  UNOPOSE_REQUIRE(B >= 0 && T >= 1 && H >= 1 && (long)B * H * cdiv(T, 256) < (1L << 28), "vit_attention_f32_ss: bad sizes");
  if (B == 0) return UNOPOSE_OK;
  constexpr int NW = VS_NW;
  static bool opt[64];
  const size_t lds = (size_t)2 * VS_BUFB;
  if (lds_optin(opt, reinterpret_cast<const void *>(&vit_attn_f32s_kernel<NW>), lds, "vit_attention_f32_ss") != UNOPOSE_OK) return UNOPOSE_ELAUNCH;
  const int BH = B * H, nq = cdiv(T, 32 * NW);
  const long blocks = (long)cdiv(BH, 8) * nq * 8;
  hipLaunchKernelGGL((vit_attn_f32s_kernel<NW>), dim3((unsigned)blocks), dim3(NW * 64), lds, (hipStream_t)stream, (const char *)qkv_split, T, H, BH,
                     nq, 0.125f * 1.4426950408889634f, (char *)out_split);
  return check_launch("vit_attention_f32_ss");
}

}  // extern "C"
