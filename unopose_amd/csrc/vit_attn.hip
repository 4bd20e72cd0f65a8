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
#include "gemm_common.h"
#include <cstdlib>

namespace unopose {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u16 va_f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (u16)(u >> 16);
}

constexpr int VA_CHUNK = 128;            // keys staged per LDS chunk
constexpr int VA_LDK = 64 + 8;           // padded row of the K chunk  [key][channel]
// V chunk image: 4 sub-tiles [128 keys][16 channels] (row = 32 B), sub-tile stride 4224 B.  Read with the
// gfx950 transpose read ds_read_b64_tr_b16: a 16-lane group fetches a [4 keys][16 channels] block (128
// contiguous bytes = 32 banks) and lane c receives the 4 keys of channel c -- the A fragment of V^T
// without ever transposing V in LDS.  Lanes 16-31 read the next sub-tile, 4224 B = 32 banks further on.
constexpr int VA_VSUB = 128 * 16 + 64;   // u16 per sub-tile (4096 B + 128 B bank skew)
typedef short s16x4 __attribute__((ext_vector_type(4)));

// qkv: (B, T, 3, H, 64) bf16 (the fused qkv Linear output); out: (B, T, H*64) bf16.
// The 4 waves of a workgroup share one (image, head): K and V of a 128-key chunk are loaded once with
// coalesced 16-byte reads, K kept row-major and V transposed on the way into LDS (so the caller needs no
// V^T copy), then every wave runs its 32 queries against the chunk.
__device__ __forceinline__ uint32_t va_cvt_pk(float a, float b) {  // packed RNE fp32 -> bf16 (gfx950)
  return cvt_pk_bf16_f32(a, b);
}

constexpr float VA_DEFER = 8.f;  // log2 of the largest P the deferred rescale lets through
constexpr int VA_BUF = VA_CHUNK * VA_LDK + 4 * VA_VSUB;  // u16 per chunk buffer (K rows + V sub-tiles)

// QB = 32-query blocks per wave (a workgroup owns 128 QB queries of one (image, head)); NBUF = LDS chunk
// buffers.  QB = 2 reads every K / V fragment once for two independent score tiles -- half the LDS traffic
// per MFMA and two dependency chains per wave for the scheduler to interleave (QK^T of one block under
// the softmax VALU work of the other); it takes ~2x the registers, so it runs 2 waves/SIMD with a
// double-buffered chunk (one barrier per chunk).  QB = 1 keeps short sequences (T = 261) from wasting
// half-empty workgroups.
// Workgroup -> (image, head, query block): the linear id is dealt round-robin over the 8 XCDs by the
// dispatcher, so id%8 picks the XCD and all query blocks of one (image, head) are given the same id%8:
// its K / V are then fetched into ONE XCD's L2 instead of every L2.
// NW = wavefronts per workgroup: the K / V chunk staged in LDS is shared by NW * 32 QB queries, so a
// larger workgroup amortises the staging (global loads, LDS writes, barriers: ~1/3 of the kernel at NW = 4
// by ablation) over twice the MFMA work.
// Ask the scheduler for an MFMA : VALU interleave in the pipelined tile (one matrix instruction, then up to VA_FILL
// vector instructions, 16 times): the matrix pipe hides a few VALU issues per MFMA only if they sit in its shadow.
#ifndef VA_FILL
#define VA_FILL 6
#endif
#define VA_IL1 __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, VA_FILL, 0);
#define VA_INTERLEAVE VA_IL1 VA_IL1 VA_IL1 VA_IL1 VA_IL1 VA_IL1 VA_IL1 VA_IL1 VA_IL1 VA_IL1 VA_IL1 VA_IL1 VA_IL1 VA_IL1 VA_IL1 VA_IL1

template <int QB, int NBUF, int NW, bool PIPE = false>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(QB == 1 ? 3 : (QB == 4 ? 1 : 2), QB == 1 ? 4 : (QB == 4 ? 1 : 2)))) void vit_attn_kernel(const u16 *__restrict__ qkv, int T, int H, int BH, int nq,
                                                       float scale_log2e, u16 *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) u16 smem[];
  u16 (*Ot)[32][72] = reinterpret_cast<u16 (*)[32][72]>(smem);
  static_assert(NW * 32 * 72 <= NBUF * VA_BUF, "output staging must fit the chunk buffers");
  constexpr int NT = NW * 64;  // threads
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int bh = (slot / nq) * 8 + xcd, qblk = slot % nq;
  if (bh >= BH) return;
  const int b = bh / H, h = bh % H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q0 = (qblk * NW + wave) * (32 * QB);
  const bool active = q0 < T;  // inactive waves still help staging and hit every barrier
  const int col = lane & 31, hb = lane >> 5;
  const int C3 = 3 * H * 64;
  const u16 *base = qkv + (size_t)b * T * C3;
  bf16x8 qf[QB][4];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int tq = min(q0 + qb * 32 + col, T - 1);
    const u16 *qp = base + (size_t)tq * C3 + h * 64 + hb * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[qb][ks] = *reinterpret_cast<const bf16x8 *>(qp + ks * 16);
  }
  // Pin the Q loads as complete HERE: otherwise the compiler waits for them with vmcnt(0) at their first
  // use inside the tile loop, which also drains the K / V chunk prefetch issued in the meantime.
#pragma unroll
  for (int qb = 0; qb < QB; ++qb)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qf[qb][ks]));
  f32x16 o[QB][2];
  // running max in the exp2 domain (score * scale * log2 e) and running sum
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

  // register-staged prefetch of a chunk: 128 keys x 128 B of K and of V.
  //   K: 8 lanes per key row (contiguous 128 B; the 16-byte LDS stores of 8 lanes cover all 32 banks)
  //   V: 8 lanes = 4 keys x the two 16-byte halves of one 16-channel sub-tile row (same property)
  // (named registers, not arrays: an array captured by the lambdas below ends up in scratch)
  uint4 pk0, pk1, pk2, pk3, pv0, pv1, pv2, pv3;
#define VA_LOAD(i, PK, PV)                                                                                       \
  {                                                                                                              \
    const int e = tid + i * NT, key = e >> 3, c8 = e & 7;                                                        \
    PK = *reinterpret_cast<const uint4 *>(base + (size_t)min(c0 + key, T - 1) * C3 + H * 64 + h * 64 + c8 * 8);  \
    const int vkey = (e >> 5) * 4 + ((e >> 1) & 3), vc = ((e >> 3) & 3) * 16 + (e & 1) * 8;                      \
    PV = *reinterpret_cast<const uint4 *>(base + (size_t)min(c0 + vkey, T - 1) * C3 + 2 * H * 64 + h * 64 + vc); \
  }
#define VA_STORE(i, PK, PV)                                                                  \
  {                                                                                          \
    const int e = tid + i * NT, key = e >> 3, c8 = e & 7;                                    \
    *reinterpret_cast<uint4 *>(buf + key * VA_LDK + c8 * 8) = PK;                            \
    const int vkey = (e >> 5) * 4 + ((e >> 1) & 3);                                          \
    /* keys beyond T contribute V = 0 (their P is 0 as well) */                              \
    *reinterpret_cast<uint4 *>(buf + VA_CHUNK * VA_LDK + ((e >> 3) & 3) * VA_VSUB + vkey * 16 + (e & 1) * 8) = \
        c0 + vkey < T ? PV : make_uint4(0u, 0u, 0u, 0u);                                     \
  }
  // 1024 16-byte pieces of K and of V per chunk: 1024 / NT per thread
  auto chunk_load = [&](int c0) {
    VA_LOAD(0, pk0, pv0) VA_LOAD(1, pk1, pv1)
    if (NT == 256) { VA_LOAD(2, pk2, pv2) VA_LOAD(3, pk3, pv3) }
  };
  auto chunk_store = [&](int c0, u16 *buf) {
    VA_STORE(0, pk0, pv0) VA_STORE(1, pk1, pv1)
    if (NT == 256) { VA_STORE(2, pk2, pv2) VA_STORE(3, pk3, pv3) }
  };
  static_assert(NT == 256 || NT == 512, "4 or 8 wavefronts per workgroup");
#undef VA_LOAD
#undef VA_STORE
  // this lane's slot in the transpose read: sub-tile (lane>>4)&1, key row 4 hb + ((lane&15)>>2), 8-byte chunk lane&3
  const int vlane_off = VA_CHUNK * VA_LDK + ((lane >> 4) & 1) * VA_VSUB + (4 * hb + ((lane & 15) >> 2)) * 16 + (lane & 3) * 4;

  // ---- pieces of one 32-key tile ------------------------------------------------------------------
  // S^T = K Q^T: rows = 32 keys, cols = 32 queries, for every query block of the wave
  auto qk_tile = [&](const u16 *buf, int kt, f32x16 (&s)[QB]) {
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[qb][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 kf = *reinterpret_cast<const bf16x8 *>(buf + (kt + col) * VA_LDK + ks * 16 + hb * 8);
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) s[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[qb][ks], s[qb], 0, 0, 0);
    }
  };
  // Online softmax + P.V of one tile, written so that the COMMON path is one basic block from the score
  // MFMAs to the P.V MFMAs (a wave-uniform rescale branch in the middle of the tile cost 1/3 of the kernel
  // by ablation: it splits the block the scheduler interleaves MFMA, exp and VALU work in).
  //  * Deferred reference point: a query's reference m_run moves only when its tile maximum exceeds it by
  //    more than 2^VA_DEFER; until then P = exp2(s - m_run) may reach 2^VA_DEFER instead of 1 -- harmless in
  //    the fp32 accumulators, bf16 rounds P with the same RELATIVE error at any magnitude, and O / l does not
  //    depend on the reference point.  The choice is a per-lane select, not a branch.
  //  * l is kept exact branch-free: l = l * alpha + sum(P) with alpha = exp2(m_old - m_new) (= 1 if unmoved).
  //  * O is accumulated as if alpha were 1; if ANY query of the wave moved (first tile, then rare) a fix-up
  //    at the END of the tile recomputes this tile's P.V into a scratch accumulator D and sets
  //    O = (O - D) * alpha + D.  (O - D) is the old accumulator up to one rounding of the much larger new
  //    terms, and it is scaled by alpha <= 2^-VA_DEFER (or by 0 on the first tile, where O - D is exactly 0).
  auto softmax_pv_tile = [&](const u16 *buf, int kt, f32x16 (&s)[QB]) {
    const u16 *vlane = buf + vlane_off;
    union PF { bf16x8 v; uint32_t w[4]; } pf[QB][2];
    float alpha[QB];
    bool moved = false;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      float mx = s[qb][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[qb][r]);
      const auto sm = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      mx = fmaxf(__uint_as_float(sm[0]), __uint_as_float(sm[1])) * scale_log2e;  // scale > 0: max commutes with it
      const bool grow = mx > m_run[qb] + VA_DEFER;
      const float m_use = grow ? mx : m_run[qb];
      alpha[qb] = __builtin_amdgcn_exp2f(m_run[qb] - m_use);
      m_run[qb] = m_use;
      moved |= grow;
      float ls = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[qb][r] = __builtin_amdgcn_exp2f(fmaf(s[qb][r], scale_log2e, -m_use));
        ls += s[qb][r];
      }
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(ls), __float_as_uint(ls), false, false);
      l_run[qb] = fmaf(l_run[qb], alpha[qb], __uint_as_float(sw[0]) + __uint_as_float(sw[1]));
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int e = 0; e < 4; ++e) pf[qb][s2].w[e] = va_cvt_pk(s[qb][s2 * 8 + 2 * e], s[qb][s2 * 8 + 2 * e + 1]);
    }
    // ---- O^T += V^T P^T ; k-step s2 = accumulator registers 8*s2 .. 8*s2+7 of S (permuted key order that
    // the V fragment reads follow); A = V^T: row = channel t*32 + col, keys kt + 16 s2 + 4 hb + {0..3}, + 8
    auto v_frag = [&](int s2, int t) {
      const u16 *vp = vlane + t * 2 * VA_VSUB + (kt + s2 * 16) * 16;
      union { bf16x8 v; s16x4 h4[2]; } vf;
      vf.h4[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(vp));
      vf.h4[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(vp + 8 * 16));
      return vf.v;
    };
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bf16x8 vf = v_frag(s2, t);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
          o[qb][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[qb][s2].v, o[qb][t], 0, 0, 0);
      }
    if (__builtin_expect(__any(moved), 0)) {  // rare after the first tile: re-reference the accumulators of the queries that moved
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

  auto chunk_compute = [&](int c0, const u16 *buf) {
    const int nk = min(VA_CHUNK, T - c0);  // valid keys of this chunk
    int kt = 0;
    if (PIPE) {
      // software pipeline over the chunk's full tiles: the K Q^T MFMAs of tile t+1 are issued BEFORE the softmax of tile t
      // (two named score sets, static indexing), so the matrix pipe works under the exp / max / sum VALU stream
      if (nk >= 32) {
        f32x16 sa[QB], sb[QB];
        qk_tile(buf, 0, sa);
        for (;;) {
          const bool more_b = kt + 64 <= nk;
          if (more_b) qk_tile(buf, kt + 32, sb);
          softmax_pv_tile(buf, kt, sa);
          VA_INTERLEAVE
          kt += 32;
          if (!more_b) break;
          const bool more_a = kt + 64 <= nk;
          if (more_a) qk_tile(buf, kt + 32, sa);
          softmax_pv_tile(buf, kt, sb);
          VA_INTERLEAVE
          kt += 32;
          if (!more_a) break;
        }
      }
    } else {
      for (; kt + 32 <= nk; kt += 32) {  // full tiles: no masking, no branch between the MFMA groups
        f32x16 s[QB];
        qk_tile(buf, kt, s);
        softmax_pv_tile(buf, kt, s);
      }
    }
    if (kt < nk) {  // the one partial tile of the sequence: keys >= T are masked out
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

  const int nchunks = (T + VA_CHUNK - 1) / VA_CHUNK;
  if (NBUF == 1) {
    chunk_load(0);
    for (int c = 0; c < nchunks; ++c) {
      __syncthreads();  // previous chunk fully consumed
      chunk_store(c * VA_CHUNK, smem);
      if (c + 1 < nchunks) chunk_load((c + 1) * VA_CHUNK);  // in flight under this chunk's MFMAs
      __syncthreads();
      if (active) chunk_compute(c * VA_CHUNK, smem);
    }
  } else {
    chunk_load(0);
    chunk_store(0, smem);
    if (nchunks > 1) chunk_load(VA_CHUNK);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
      if (active) chunk_compute(c * VA_CHUNK, smem + (c & 1) * VA_BUF);
      if (c + 1 < nchunks) {  // the other buffer was last read before the previous barrier
        chunk_store((c + 1) * VA_CHUNK, smem + ((c + 1) & 1) * VA_BUF);
        if (c + 2 < nchunks) chunk_load((c + 2) * VA_CHUNK);
      }
      __syncthreads();
    }
  }
  __syncthreads();  // every wave is done with the K / V chunks before they become the output buffer
  if (!active) return;
  // ---- normalise, transpose through LDS, store token rows
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const float inv = 1.f / l_run[qb];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hb;
        Ot[wave][col][c] = va_f2bf(o[qb][t][r] * inv);
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // 32 rows x 128 B: each lane moves 16 B; 8 lanes cover one row
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


// ------------------------------------------------------------------------------------------------------------------------------
// Round 4: the long-sequence form (T >= 1024).  Same arithmetic as vit_attn_kernel<2, ...> above -- 64 queries per wave, swapped
// product, deferred reference point, end-of-tile fix-up -- with two structural changes:
//  * K / V chunks arrive by LDS-DMA (`buffer_load_dwordx4 ... lds`: 32 one-KiB pieces per 128-key chunk; the per-lane source addresses
//    swizzle the K rows (16-byte chunk c of row r at c ^ ((r >> 1) & 7): conflict-free ds_read_b128 without padding) and lay V out as
//    the [sub-tile][key][16 channels] image ds_read_b64_tr_b16 wants; keys past the sequence fall outside the buffer descriptor and
//    arrive as zeros): no staging registers (233 -> 213 VGPRs), no ds_write, one counted wait + barrier per chunk;
//  * 4 waves per workgroup and TWO workgroups per CU (66.5 KiB of LDS each): with one 8-wave workgroup per CU the Q loads, the first
//    chunk and the output pass of every workgroup ran with nothing else on the CU -- 9 workgroups per CU and launch, ~10 % of the time.
// Same box, 64 x 12 x 1374: 489.6 us (vit_attn_kernel<2, 2, 8>) -> 472 (8 waves, LDS-DMA, 2 buffers) -> 450-469 us (this kernel);
// a third chunk buffer (one workgroup per CU again): 622 us; results equal to the old kernel's (scripts/ubench/vit_attn_var.py).
constexpr int VD_KBYTES = VA_CHUNK * 128;          // K chunk: [key][64 ch] bf16, swizzled
constexpr int VD_VSUBB = 128 * 32 + 128;           // bytes per V sub-tile [128 keys][16 channels] + bank skew
constexpr int VD_BUFB = VD_KBYTES + 4 * VD_VSUBB;  // 33 280 B per chunk buffer

template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void vit_attn_kernel_dma(const u16 *__restrict__ qkv, int T, int H, int BH, int nq, float scale_log2e,
                                                               u16 *__restrict__ out) {
  constexpr int QB = 2, NBUF = 2, NPC = 32 / NW;  // NPC: LDS-DMA pieces per wave and chunk, half of them K, half V
  extern __shared__ __attribute__((aligned(1024))) char smemd[];
  u16 (*Ot)[32][72] = reinterpret_cast<u16 (*)[32][72]>(smemd);
  static_assert(NW * 32 * 72 * 2 <= NBUF * VD_BUFB, "output staging must fit the chunk buffers");
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int bh = (slot / nq) * 8 + xcd, qblk = slot % nq;
  if (bh >= BH) return;
  const int b = bh / H, h = bh % H;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q0 = (qblk * NW + wave) * (32 * QB);
  const bool active = q0 < T;  // inactive waves still issue their DMA pieces and hit every barrier
  const int col = lane & 31, hb = lane >> 5;
  const int C3 = 3 * H * 64;
  const u16 *base = qkv + (size_t)b * T * C3;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smemd;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)((size_t)T * C3 * 2), 0x00020000);
  uint32_t kvo[NPC / 2], vvo[NPC / 2], kdst[NPC / 2], vdst[NPC / 2];
#pragma unroll
  for (int i = 0; i < NPC / 2; ++i) {
    const int p = (NPC / 2) * wave + i;  // piece 0 .. 15 of K and of V
    const int krow = 8 * p + (lane >> 3), gch = (lane & 7) ^ ((krow >> 1) & 7);
    kvo[i] = (uint32_t)(krow * C3 * 2 + (H * 64 + h * 64) * 2 + gch * 16);
    kdst[i] = (uint32_t)(p * 1024);
    const int sub = p >> 2, vkey = 32 * (p & 3) + (lane >> 1);
    vvo[i] = (uint32_t)(vkey * C3 * 2 + (2 * H * 64 + h * 64 + sub * 16 + (lane & 1) * 8) * 2);
    vdst[i] = (uint32_t)(VD_KBYTES + sub * VD_VSUBB + (p & 3) * 1024);
  }
  auto issue_chunk = [&](int c, int bslot) {  // (the key-row offset rides in the VGPR offset: the descriptor's range check covers it)
    const uint32_t b0 = lds0 + bslot * VD_BUFB, ro = (uint32_t)(c * VA_CHUNK) * (uint32_t)(C3 * 2);
#pragma unroll
    for (int i = 0; i < NPC / 2; ++i) gemm_dma16(b0 + kdst[i], kvo[i] + ro, rs, 0);
#pragma unroll
    for (int i = 0; i < NPC / 2; ++i) gemm_dma16(b0 + vdst[i], vvo[i] + ro, rs, 0);
  };
  const int nchunks = (T + VA_CHUNK - 1) / VA_CHUNK;
  issue_chunk(0, 0);

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
    for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qf[qb][ks]));  // the Q loads are complete HERE (see above)
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
  uint32_t koff[4];  // K fragment of key row kt + col (kt a multiple of 32), k-step ks: 16-byte chunk 2 ks + hb, swizzled by the row
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) koff[ks] = (uint32_t)(col * 128 + (((2 * ks + hb) ^ ((col >> 1) & 7)) << 4));
  const uint32_t vlane_off = (uint32_t)(VD_KBYTES + ((lane >> 4) & 1) * VD_VSUBB + (4 * hb + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8);

  auto tile = [&](const char *buf, int c0, int kt, bool partial) {
    f32x16 s[QB];
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
    if (partial) {  // the one partial tile of the sequence: keys >= T are masked out
#pragma unroll
      for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (c0 + kt + (r & 3) + 8 * (r >> 2) + 4 * hb >= T) s[qb][r] = -3e38f;
    }
    const char *vlane = buf + vlane_off;
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
        s[qb][r] = __builtin_amdgcn_exp2f(fmaf(s[qb][r], scale_log2e, -m_use));
        ls += s[qb][r];
      }
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(ls), __float_as_uint(ls), false, false);
      l_run[qb] = fmaf(l_run[qb], alpha[qb], __uint_as_float(sw[0]) + __uint_as_float(sw[1]));
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int e = 0; e < 4; ++e) pf[qb][s2].w[e] = va_cvt_pk(s[qb][s2 * 8 + 2 * e], s[qb][s2 * 8 + 2 * e + 1]);
    }
    auto v_frag = [&](int s2, int t) {
      const char *vp = vlane + t * 2 * VD_VSUBB + (kt + s2 * 16) * 32;
      union { bf16x8 v; s16x4 h4[2]; } vf;
      vf.h4[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(vp));
      vf.h4[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(vp + 8 * 32));
      return vf.v;
    };
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bf16x8 vf = v_frag(s2, t);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) o[qb][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[qb][s2].v, o[qb][t], 0, 0, 0);
      }
    if (__builtin_expect(__any(moved), 0)) {  // rare after the first tile: re-reference the accumulators of the queries that moved
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

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // chunk 0 landed; every wave's pieces visible after the barrier
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  for (int c = 0; c < nchunks; ++c) {
    if (c + 1 < nchunks) issue_chunk(c + 1, (c + 1) & 1);  // (that buffer was last read in chunk c - 1, before the barrier that ended it)
    if (active) {
      const int c0 = c * VA_CHUNK, nk = min(VA_CHUNK, T - c0);
      const char *buf = smemd + (c & 1) * VD_BUFB;
      int kt = 0;
      for (; kt + 32 <= nk; kt += 32) tile(buf, c0, kt, false);
      if (kt < nk) tile(buf, c0, kt, true);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  if (!active) return;
  // ---- normalise, transpose through LDS, store token rows
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const float inv = 1.f / l_run[qb];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) Ot[wave][col][t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hb] = va_f2bf(o[qb][t][r] * inv);
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

static int launch_vit_attn_dma(const void *qkv, int B, int T, int H, void *out, hipStream_t stream) {
  constexpr int NW = 4;
  static bool opt[64];
  const size_t lds = (size_t)2 * VD_BUFB;
  if (lds_optin(opt, reinterpret_cast<const void *>(&vit_attn_kernel_dma<NW>), lds, "vit_attention") != UNOPOSE_OK) return UNOPOSE_ELAUNCH;
  const int BH = B * H, nq = cdiv(T, 32 * NW * 2);
  const long blocks = (long)cdiv(BH, 8) * nq * 8;
  hipLaunchKernelGGL((vit_attn_kernel_dma<NW>), dim3((unsigned)blocks), dim3(NW * 64), lds, stream, (const u16 *)qkv, T, H, BH, nq,
                     0.125f * 1.4426950408889634f, (u16 *)out);
  return check_launch("vit_attention");
}

template <int QB, int NBUF, int NW, bool PIPE = false>
static int launch_vit_attn(const void *qkv, int B, int T, int H, void *out, hipStream_t stream) {
  static bool opt[64];  // > 64 KiB of LDS needs the opt-in, per device (idempotent; benign if raced)
  const size_t lds = (size_t)NBUF * VA_BUF * sizeof(u16);
  if (lds_optin(opt, reinterpret_cast<const void *>(&vit_attn_kernel<QB, NBUF, NW, PIPE>), lds, "vit_attention") != UNOPOSE_OK) return UNOPOSE_ELAUNCH;
  const int BH = B * H, nq = cdiv(T, 32 * NW * QB);
  const float scale_log2e = 0.125f * 1.4426950408889634f;  // 1/sqrt(64) * log2(e)
  const long blocks = (long)cdiv(BH, 8) * nq * 8;
  hipLaunchKernelGGL((vit_attn_kernel<QB, NBUF, NW, PIPE>), dim3((unsigned)blocks), dim3(NW * 64), lds, stream, (const u16 *)qkv, T,
                     H, BH, nq, scale_log2e, (u16 *)out);
  return check_launch("vit_attention");
}

extern "C" {

int unopose_vit_attention(const void *qkv, int B, int T, int H, void *out, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(qkv && out, "vit_attention: null pointer");
  UNOPOSE_REQUIRE(B >= 0 && T >= 1 && H >= 1 && (long)B * H * cdiv(T, 128) < (1L << 31), "vit_attention: bad sizes");
  if (B == 0) return UNOPOSE_OK;
#ifdef UNOPOSE_PROBE_BUILD  // A/B routes exist in probe builds only (UNOPOSE_EXTRA_HIPCC_FLAGS=-DUNOPOSE_PROBE_BUILD; scripts/ubench/vit_attn_modes.py)
  static const int nw_env = getenv("UNOPOSE_VIT_NW") ? atoi(getenv("UNOPOSE_VIT_NW")) : 8;
  static const int qb4_env = getenv("UNOPOSE_VIT_QB4") ? atoi(getenv("UNOPOSE_VIT_QB4")) : 0;  // experiment: 1 wave / SIMD, 128 queries / wave
  if (T >= 1024 && qb4_env == 1) return launch_vit_attn<4, 2, 4>(qkv, B, T, H, out, (hipStream_t)stream);
  static const int dma_env = getenv("UNOPOSE_VIT_DMA") ? atoi(getenv("UNOPOSE_VIT_DMA")) : 1;  // 0: round 3's register-staged 8-wave kernel (A/B)
  static const int pipe_env = getenv("UNOPOSE_VIT_PIPE") ? atoi(getenv("UNOPOSE_VIT_PIPE")) : 0;
  if (T >= 1024 && dma_env != 1 && pipe_env == 1) return launch_vit_attn<2, 2, 8, true>(qkv, B, T, H, out, (hipStream_t)stream);
  if (T >= 1024 && dma_env != 1 && nw_env != 8) return launch_vit_attn<2, 2, 4>(qkv, B, T, H, out, (hipStream_t)stream);
  if (T >= 1024 && dma_env != 1) return launch_vit_attn<2, 2, 8>(qkv, B, T, H, out, (hipStream_t)stream);
#endif
  // T >= 1024: LDS-DMA staging, 4-wave workgroups, two per CU; a qkv image past 2 GiB per crop (32-bit DMA offsets): the register-staged kernel
  if (T >= 1024 && (size_t)T * 3 * H * 64 * 2 < (1UL << 31)) return launch_vit_attn_dma(qkv, B, T, H, out, (hipStream_t)stream);
  if (T >= 1024) return launch_vit_attn<2, 2, 8>(qkv, B, T, H, out, (hipStream_t)stream);
  if (T >= 512) return launch_vit_attn<2, 2, 4>(qkv, B, T, H, out, (hipStream_t)stream);
  return launch_vit_attn<1, 1, 4>(qkv, B, T, H, out, (hipStream_t)stream);
}

}  // extern "C"
