// Token attention of the correspondence transformer for gfx950 (C ABI part 2).
//
// Replaces the attention cores of MultiHeadAttention (cross) and RPEMultiHeadAttention (self, with
// relative-position embedding) -- core/unopose/model/transformer.py:130-148 and :386-405 -- for the
// 197-token coarse sequences (4 heads x 64):
//     P = softmax((q k^T + q proj_p(E)[n,m]) / 8),  out = P v
// The reference materialises proj_p(E) as a (B,4,n,m,64) tensor (5.09 GFLOP + 40 MB per cloud and layer).
// Here the RPE term is folded, q.(W_p e + b_p) = (q W_p).e + const (the constant cancels in the
// softmax), and ONE kernel streams E[n] (the only HBM-sized operand) exactly once: a wavefront owns four
// query rows x four heads = the 16 rows of a v_mfma_f32_16x16x32_bf16 tile, so q k^T, the four per-row
// (q W_p) E[n]^T products, the softmax (row == 16-lane DPP row) and P v all run on one set of
// accumulators without touching HBM in between.
#include "gemm_common.h"  // (gemm_dma16: one 1-KiB LDS-DMA piece)

namespace unopose {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

__device__ __forceinline__ u16 f2bf_rn(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (u16)(u >> 16);
}

#ifndef TA_DPP_BUTTERFLY
#define TA_DPP_BUTTERFLY 1  // 0: the butterfly through ds_swizzle (LDS crossbar); see DESIGN.md section 7
#endif
template <int XOR>
__device__ __forceinline__ float swz_xor(float v) {  // butterfly step inside a 16-lane row, no LDS memory touched
#if TA_DPP_BUTTERFLY
  // DPP: quad_perm [1,0,3,2] / [2,3,0,1] for xor 1 / 2; row_half_mirror / row_mirror for xor 4 / 8 -- applied in this order every
  // lane's partner group holds the value the xor partner would (the groups are uniform by then), so sums and maxima are unchanged
  constexpr int ctrl = XOR == 1 ? 0xB1 : XOR == 2 ? 0x4E : XOR == 4 ? 0x141 : 0x140;
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xF, 0xF, false));
#else
  return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), (XOR << 10) | 0x1F));
#endif
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, swz_xor<1>(v));
  v = fmaxf(v, swz_xor<2>(v));
  v = fmaxf(v, swz_xor<4>(v));
  v = fmaxf(v, swz_xor<8>(v));
  return v;
}
__device__ __forceinline__ float row16_sum(float v) {
  v += swz_xor<1>(v);
  v += swz_xor<2>(v);
  v += swz_xor<4>(v);
  v += swz_xor<8>(v);
  return v;
}

__device__ __forceinline__ bf16x8 zero8() {
  bf16x8 z;
#pragma unroll
  for (int i = 0; i < 8; ++i) z[i] = (__bf16)0.f;
  return z;
}

constexpr int TA_NT = 14;          // key tiles of 16 -> up to 224 keys
constexpr int TA_MP = TA_NT * 16;  // padded key count (V^T and the P staging use this stride)

// RW = query rows per wavefront (4 fills the 16-row MFMA tile; 2 doubles the number of wavefronts that
// stream E concurrently -- the kernel is bound by HBM latency x occupancy, not by the matrix pipe)
template <bool RPE, int RW>
__global__ __launch_bounds__(256, 2) void token_attn_kernel(const u16 *__restrict__ q, int ldq,
                                                         const u16 *__restrict__ k, int ldk,
                                                         const u16 *__restrict__ vt, const u16 *__restrict__ qp,
                                                         int ldqp, const u16 *__restrict__ E, int n, int m,
                                                         float scale, u16 *__restrict__ out) {
  __shared__ __attribute__((aligned(16))) u16 Pl[4][16][TA_MP];
  __shared__ __attribute__((aligned(16))) u16 Al[RPE ? 4 : 1][8][64][8];  // RPE: A fragments of the current query row
  const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = (blockIdx.x * 4 + wave) * RW;  // first of this wave's RW query rows
  if (n0 >= n) return;
  const int li = lane & 15, kg = lane >> 4;     // A: row li, k-group kg | B: column li, k-group kg
  const int a_nl = li >> 2, a_h = li & 3;       // A-operand row = (query row a_nl, head a_h)
  const bool a_valid = a_nl < RW && n0 + a_nl < n;
  const u16 *Q = q + ((size_t)b * n + n0 + a_nl) * ldq;
  const u16 *K = k + (size_t)b * m * ldk;

  f32x4 acc[TA_NT];
#pragma unroll
  for (int t = 0; t < TA_NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- q k^T: block-diagonal A (row (n_l,h) only sees head h's 64 channels), B = K.
  // K is L2-resident but every fragment load still costs an L2 round trip: the 14 key-tile loads of
  // k-step ks+1 are all issued before the 14 MFMAs of k-step ks (double-buffered), not one by one.
  {
    bf16x8 qa[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int kk = ks * 32 + kg * 8;
      qa[ks] = zero8();
      if (a_valid && (kk >> 6) == a_h) qa[ks] = *reinterpret_cast<const bf16x8 *>(Q + kk);
    }
    bf16x8 kb[2][TA_NT];
    auto load_k = [&](int ks, bf16x8 (&dst)[TA_NT]) {
#pragma unroll
      for (int t = 0; t < TA_NT; ++t) {
        // keys >= m read the last key's row instead of branching per load: their scores are masked below
        const int mm = min(t * 16 + li, m - 1);
        dst[t] = *reinterpret_cast<const bf16x8 *>(K + (size_t)mm * ldk + ks * 32 + kg * 8);
      }
    };
    load_k(0, kb[0]);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if (ks + 1 < 8) load_k(ks + 1, kb[(ks + 1) & 1]);
#pragma unroll
      for (int t = 0; t < TA_NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[ks], kb[ks & 1][t], acc[t], 0, 0, 0);
    }
  }
  // ---- RPE term: for each of the 4 query rows, (q W_p)[n] . E[n,m,:]
  if (RPE) {
    for (int nl = 0; nl < RW; ++nl) {
      if (n0 + nl >= n) break;  // wave-uniform
      const u16 *QP = qp + ((size_t)b * n + n0 + nl) * ldqp + a_h * 256;
      const u16 *En = E + ((size_t)b * n + n0 + nl) * (size_t)m * 256;
      // this query row's folded (q W_p) fragments live in LDS (8 KiB per wave), not in 32 VGPRs: the
      // registers buy a second E tile in flight instead (the kernel is bound by HBM latency x bytes in
      // flight: 2 waves/SIMD x 16 KiB each)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        bf16x8 av = zero8();
        if (a_nl == nl) av = *reinterpret_cast<const bf16x8 *>(QP + ks * 32 + kg * 8);
        *reinterpret_cast<bf16x8 *>(&Al[wave][ks][lane][0]) = av;
      }
      // software-pipelined over key tiles, two tiles (2 x 8 KiB per wave) ahead of the MFMAs -- E is the
      // only HBM-sized stream of this kernel
      const int nt_valid = (m + 15) >> 4;
      bf16x8 buf[3][8];
      auto load_tile = [&](int t, bf16x8 (&dst)[8]) {
        const u16 *Er = En + (size_t)min(t * 16 + li, m - 1) * 256 + kg * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) dst[ks] = *reinterpret_cast<const bf16x8 *>(Er + ks * 32);
      };
      load_tile(0, buf[0]);
      if (1 < nt_valid) load_tile(1, buf[1]);
#pragma unroll
      for (int t = 0; t < TA_NT; ++t) {
        if (t >= nt_valid) continue;  // uniform: tile entirely beyond the keys
        if (t + 2 < nt_valid) load_tile(t + 2, buf[(t + 2) % 3]);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          const bf16x8 av = *reinterpret_cast<const bf16x8 *>(&Al[wave][ks][lane][0]);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, buf[t % 3][ks], acc[t], 0, 0, 0);
        }
      }
    }
  }
  // ---- softmax over keys: C/D layout row = kg*4 + reg = (query row kg, head reg), column = key li
  float mx[4] = {-3e38f, -3e38f, -3e38f, -3e38f};
#pragma unroll
  for (int t = 0; t < TA_NT; ++t) {
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
    mx[h] = row16_max(mx[h]);
    sm[h] = 0.f;
  }
#pragma unroll
  for (int t = 0; t < TA_NT; ++t) {
    const bool ok = t * 16 + li < m;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const float p = ok ? __expf(acc[t][h] - mx[h]) : 0.f;
      acc[t][h] = p;
      sm[h] += p;
    }
  }
#pragma unroll
  for (int h = 0; h < 4; ++h) sm[h] = 1.f / row16_sum(sm[h]);
  // ---- P (bf16) -> LDS in A-operand order: row (query row, head), column key
#pragma unroll
  for (int t = 0; t < TA_NT; ++t)
#pragma unroll
    for (int h = 0; h < 4; ++h) Pl[wave][kg * 4 + h][t * 16 + li] = f2bf_rn(acc[t][h] * sm[h]);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // ---- P v: B = V^T (channel-major, zero-padded keys); only the head-matching quarter is kept
  bf16x8 pa[TA_MP / 32];
#pragma unroll
  for (int ks = 0; ks < TA_MP / 32; ++ks)
    pa[ks] = *reinterpret_cast<const bf16x8 *>(&Pl[wave][li][ks * 32 + kg * 8]);
  const u16 *VT = vt + (size_t)b * 256 * TA_MP;
  constexpr int PK = TA_MP / 32;
  bf16x8 vb[2][PK];
  auto load_v = [&](int nt, bf16x8 (&dst)[PK]) {
    const u16 *Vr = VT + (size_t)(nt * 16 + li) * TA_MP + kg * 8;
#pragma unroll
    for (int ks = 0; ks < PK; ++ks) dst[ks] = *reinterpret_cast<const bf16x8 *>(Vr + ks * 32);
  };
  load_v(0, vb[0]);
#pragma unroll
  for (int nt = 0; nt < 16; ++nt) {
    if (nt + 1 < 16) load_v(nt + 1, vb[(nt + 1) & 1]);  // next channel tile's V^T fragments in flight
    f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < PK; ++ks) o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa[ks], vb[nt & 1][ks], o, 0, 0, 0);
    // D row = (query row kg, head reg); channel tile nt belongs to head nt >> 2
    if (kg < RW && n0 + kg < n) out[((size_t)b * n + n0 + kg) * 256 + nt * 16 + li] = f2bf_rn(o[nt >> 2]);
  }
}

// ---- Round 5: the RPE self-attention with the embedding stream staged through LDS by LDS-DMA.
// token_attn_kernel<true> loads E in FRAGMENT shape: a wave instruction fetches 16 key rows x 64 B (half cache lines, 512 B apart), 8
// instructions per 16-key tile -- 4.25 TB/s of the 8 TB/s a stream of this size can get (the LayerNorm glue, 16 contiguous bytes per
// lane, reaches 6.2).  A query row's E[n] is ONE contiguous block of m x 512 B: here a 16-key tile (8 KiB) arrives as 8 pieces of 1 KiB
// = whole contiguous cache lines per instruction, written straight into LDS (no staging registers), and the B fragments are read back
// with ds_read_b128.  LDS image of a tile: key r at r * 512 B, its 16-byte chunk c at position c ^ r (low 4 bits): the XOR is applied to
// the SOURCE address of each lane (a permutation inside the key's own 512 B: the instruction still covers whole lines) and makes the
// 16 lanes of every ds_read_b128 group hit 16 different bank quads.  A wave streams alone (2 tile buffers = 16 KiB per wave, nobody else
// reads them): no barriers, only its own counted vmcnt -- the tile after next is issued as soon as the current one has been read, so
// one to two tiles (8 - 16 KiB per wave, 64 - 128 KiB per CU) are in flight all the time.  Same MFMA order as token_attn_kernel<true>:
// bit-identical results.
#ifndef TA_EPOL
#define TA_EPOL 1  // cache policy of the embedding stream's LDS-DMA loads (gemm_dma16<POL>: 0 default, 1 nt, 2 sc1, 3 sc0 sc1 nt).  The stream (1.27 GB per launch, read once) no longer pushes K / V^T -- re-read by every wave of a cloud -- out of L2: 291.5 -> 272.5 us at 64 x 197 x 197 x 256 (nt), 291 (sc1), 272 (sc0 sc1 nt); round 6
#endif
constexpr int TA_RING = 2 * 8192;  // per wave
constexpr int TA_ROWS = 4;         // query rows per wave, processed in MFMA tiles of 4.  Measured at 64 x 197 x 197: 4 rows 290 us, 7 rows (4 + 3:
                                   // every wave task in ONE round of the 2048 wave slots) 305 us -- the kernel is bound by bytes in flight x latency, and
                                   // fewer, longer waves mean fewer bytes in flight; round 4's fragment-load kernel 307 us

__global__ __launch_bounds__(256, 2) void token_attn_rpe_dma_kernel(const u16 *__restrict__ q, int ldq, const u16 *__restrict__ k, int ldk,
                                                                     const u16 *__restrict__ vt, const u16 *__restrict__ qp, int ldqp,
                                                                     const u16 *__restrict__ E, int n, int m, float scale, int B,
                                                                     u16 *__restrict__ out) {
  constexpr int RW = 4;
  __shared__ __attribute__((aligned(1024))) char ring[4][TA_RING];   // per wave: two tile buffers; the P staging (7 KiB) reuses them afterwards
  const int b = blockIdx.y, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, kg = lane >> 4;
  const int a_nl = li >> 2, a_h = li & 3;
  // A wave owns TA_ROWS consecutive query rows, processed in tiles of 4 rows (see TA_ROWS for the measurement behind its value)
  const int nbase = (blockIdx.x * 4 + wave) * TA_ROWS;
#pragma unroll 1
  for (int n0 = nbase; n0 < min(n, nbase + TA_ROWS); n0 += RW) {
  const int rw_here = min(RW, min(n, nbase + TA_ROWS) - n0);
  const bool a_valid = a_nl < rw_here;
  const u16 *Q = q + ((size_t)b * n + n0 + a_nl) * ldq;
  const u16 *K = k + (size_t)b * m * ldk;

  f32x4 acc[TA_NT];
#pragma unroll
  for (int t = 0; t < TA_NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  // ---- the stream is started first: it runs under the q k^T part
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)&ring[0][0] + (uint32_t)wave * TA_RING;
  const int nt_valid = (m + 15) >> 4;
  const int rows_w = rw_here;                    // query rows of this tile
  const int total = rows_w * nt_valid;            // tiles of the wave's stream
  // whole-tensor descriptor: the last tile of the last query row reads past m keys -- into the next row's block, or (very last row) past
  // the end: zeros.  Keys >= m are masked in the softmax.
  const __amdgpu_buffer_rsrc_t e_rs = __builtin_amdgcn_make_buffer_rsrc((void *)E, 0, (int)min((size_t)B * n * m * 512, (size_t)0xFFFFFFFFu), 0x00020000);
  // piece j of a tile = keys 2 j + (lane >> 5): source byte = key * 512 + (((lane & 31) ^ key) << 4)  (key < 16: XOR of the low 4 chunk bits)
  uint32_t voff[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int key = 2 * j + (lane >> 5);
    voff[j] = (uint32_t)(key * 512 + (((lane & 31) ^ key) << 4));
  }
  const uint32_t row0 = (uint32_t)(((size_t)b * n + n0) * (size_t)m * 512);   // (the tensor is below 4 GiB: checked by the caller)
  auto issue_tile = [&](int s) {   // tile s of the stream: query row s / nt_valid, key tile s % nt_valid
    const int r = s / nt_valid, t = s - r * nt_valid;
    const uint32_t so = row0 + (uint32_t)r * (uint32_t)m * 512u + (uint32_t)t * 8192u;
    const uint32_t dst = lds0 + (uint32_t)(s & 1) * 8192u;
#pragma unroll
    for (int j = 0; j < 8; ++j) gemm_dma16<TA_EPOL>(dst + j * 1024, voff[j], e_rs, (int)so);
  };
  issue_tile(0);
  if (total > 1) issue_tile(1);
  // this lane's folded (q W_p) fragments: row (a_nl, a_h) of the MFMA tile, loaded once
  bf16x8 myq[8];
  {
    const u16 *QP = qp + ((size_t)b * n + n0 + a_nl) * ldqp + a_h * 256;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      myq[ks] = zero8();
      if (a_valid) myq[ks] = *reinterpret_cast<const bf16x8 *>(QP + ks * 32 + kg * 8);
    }
  }
  // ---- q k^T (as token_attn_kernel)
  {
    bf16x8 qa[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int kk = ks * 32 + kg * 8;
      qa[ks] = zero8();
      if (a_valid && (kk >> 6) == a_h) qa[ks] = *reinterpret_cast<const bf16x8 *>(Q + kk);
    }
    bf16x8 kb[2][TA_NT];
    auto load_k = [&](int ks, bf16x8 (&dst)[TA_NT]) {
#pragma unroll
      for (int t = 0; t < TA_NT; ++t) {
        const int mm = min(t * 16 + li, m - 1);
        dst[t] = *reinterpret_cast<const bf16x8 *>(K + (size_t)mm * ldk + ks * 32 + kg * 8);
      }
    };
    load_k(0, kb[0]);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if (ks + 1 < 8) load_k(ks + 1, kb[(ks + 1) & 1]);
#pragma unroll
      for (int t = 0; t < TA_NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[ks], kb[ks & 1][t], acc[t], 0, 0, 0);
    }
  }
  // (the compiler's own waits for the plain loads above have drained the first two tiles as well: from here on the counted waits rule)
  // ---- RPE term
  const char *ringw = &ring[0][0] + wave * TA_RING;
  const uint32_t frag_off = (uint32_t)(li * 512);
  int s = 0;
  for (int nl = 0; nl < rows_w; ++nl) {
    bf16x8 av[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) av[ks] = a_nl == nl ? myq[ks] : zero8();
#pragma unroll
    for (int t = 0; t < TA_NT; ++t) {
      if (t >= nt_valid) continue;  // uniform
      // tile s has landed when at most the 8 pieces of tile s + 1 are outstanding
      if (s + 1 < total)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      const char *tb = ringw + (s & 1) * 8192;
      bf16x8 eb[8];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) eb[ks] = *reinterpret_cast<const bf16x8 *>(tb + frag_off + ((((ks << 2) | kg) ^ li) << 4));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the buffer has been read: the tile after next may overwrite it
      __builtin_amdgcn_sched_barrier(0);
      if (s + 2 < total) issue_tile(s + 2);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[ks], eb[ks], acc[t], 0, 0, 0);
      ++s;
    }
  }
  // ---- softmax over keys: C/D layout row = kg*4 + reg = (query row kg, head reg), column = key li
  float mx[4] = {-3e38f, -3e38f, -3e38f, -3e38f};
#pragma unroll
  for (int t = 0; t < TA_NT; ++t) {
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
    mx[h] = row16_max(mx[h]);
    sm[h] = 0.f;
  }
#pragma unroll
  for (int t = 0; t < TA_NT; ++t) {
    const bool ok = t * 16 + li < m;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const float p = ok ? __expf(acc[t][h] - mx[h]) : 0.f;
      acc[t][h] = p;
      sm[h] += p;
    }
  }
#pragma unroll
  for (int h = 0; h < 4; ++h) sm[h] = 1.f / row16_sum(sm[h]);
  // ---- P (bf16) -> LDS in A-operand order (the wave's own ring space: the stream has ended, every tile has been read)
  u16(*Pl)[TA_MP] = reinterpret_cast<u16(*)[TA_MP]>(const_cast<char *>(ringw));
#pragma unroll
  for (int t = 0; t < TA_NT; ++t)
#pragma unroll
    for (int h = 0; h < 4; ++h) Pl[kg * 4 + h][t * 16 + li] = f2bf_rn(acc[t][h] * sm[h]);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  bf16x8 pa[TA_MP / 32];
#pragma unroll
  for (int ks = 0; ks < TA_MP / 32; ++ks) pa[ks] = *reinterpret_cast<const bf16x8 *>(&Pl[li][ks * 32 + kg * 8]);
  const u16 *VT = vt + (size_t)b * 256 * TA_MP;
  constexpr int PK = TA_MP / 32;
  bf16x8 vb[2][PK];
  auto load_v = [&](int nt, bf16x8 (&dst)[PK]) {
    const u16 *Vr = VT + (size_t)(nt * 16 + li) * TA_MP + kg * 8;
#pragma unroll
    for (int ks = 0; ks < PK; ++ks) dst[ks] = *reinterpret_cast<const bf16x8 *>(Vr + ks * 32);
  };
  load_v(0, vb[0]);
#pragma unroll
  for (int nt = 0; nt < 16; ++nt) {
    if (nt + 1 < 16) load_v(nt + 1, vb[(nt + 1) & 1]);
    f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < PK; ++ks) o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa[ks], vb[nt & 1][ks], o, 0, 0, 0);
    if (kg < rw_here) out[((size_t)b * n + n0 + kg) * 256 + nt * 16 + li] = f2bf_rn(o[nt >> 2]);
  }
  // (the P staging above reused the ring: every lane's reads of it are done before the next tile's stream is started)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }  // row tiles of the wave
}

}  // namespace unopose

using namespace unopose;

extern "C" {

int unopose_token_attention(const void *q, int ldq, const void *k, int ldk, const void *vt, const void *qp, int ldqp,
                            const void *E, int B, int n, int m, float scale, void *out, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(q && k && vt && out, "token_attention: null pointer");
  UNOPOSE_REQUIRE((qp == nullptr) == (E == nullptr), "token_attention: qp and E go together");
  UNOPOSE_REQUIRE(B >= 0 && n >= 1 && m >= 1 && m <= TA_MP && B <= 65535,
                  "token_attention: m=%d exceeds the %d-key tile", m, TA_MP);
  UNOPOSE_REQUIRE(ldq >= 256 && ldk >= 256 && ldq % 8 == 0 && ldk % 8 == 0 && (!E || (ldqp >= 1024 && ldqp % 8 == 0)),
                  "token_attention: row strides must be multiples of 8 elements (16-byte loads)");
  if (B == 0) return UNOPOSE_OK;
  hipStream_t s = (hipStream_t)stream;
  bool dma = E && (size_t)B * n * m * 512 < (1UL << 32);   // (32-bit LDS-DMA offsets; larger embeddings: the fragment-load kernel)
#ifdef UNOPOSE_PROBE_BUILD
  static const bool ta_old = getenv("UNOPOSE_TA_OLD") != nullptr;   // A/B: round 4's fragment-load kernel
  dma = dma && !ta_old;
#endif
  if (dma) {
    static bool opt[64];
    if (lds_optin(opt, (const void *)token_attn_rpe_dma_kernel, 4 * TA_RING, "token_attention") != UNOPOSE_OK) return UNOPOSE_ELAUNCH;
    hipLaunchKernelGGL(token_attn_rpe_dma_kernel, dim3(cdiv(n, 4 * TA_ROWS), B), dim3(256), 0, s, (const u16 *)q, ldq, (const u16 *)k, ldk, (const u16 *)vt,
                       (const u16 *)qp, ldqp, (const u16 *)E, n, m, scale, B, (u16 *)out);
  } else if (E)
    hipLaunchKernelGGL((token_attn_kernel<true, 4>), dim3(cdiv(n, 16), B), dim3(256), 0, s, (const u16 *)q,
                       ldq, (const u16 *)k, ldk, (const u16 *)vt, (const u16 *)qp, ldqp, (const u16 *)E, n, m, scale,
                       (u16 *)out);
  else
    hipLaunchKernelGGL((token_attn_kernel<false, 4>), dim3(cdiv(n, 16), B), dim3(256), 0, s, (const u16 *)q,
                       ldq, (const u16 *)k, ldk, (const u16 *)vt, (const u16 *)nullptr, 0, (const u16 *)nullptr, n, m,
                       scale, (u16 *)out);
  return check_launch("token_attention");
}

int unopose_token_attention_key_pad(void) { return TA_MP; }

}  // extern "C"
