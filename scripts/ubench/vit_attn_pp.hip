// EXPERIMENT (not part of libunopose_hip.so; built and timed by scripts/ubench/pp_abl.py; result in DESIGN.md section 7):
// DINOv2 ViT patch attention, "ping-pong" structure for gfx950 -- a long-sequence form of
// csrc/vit_attn.hip (same math, same LDS images for the fragment reads, same swapped K Q^T / P-from-registers tricks).
//
// Why another structure: in vit_attn.hip the two wavefronts that share a SIMD run the same code in phase, so both want
// the matrix pipe (K Q^T, P V) at the same time and then both want the VALU (max / exp / sum / convert) at the same
// time; by ablation every piece of the tile loop adds to the run time.  Here the 8 waves of a workgroup form two
// groups (waves 0-3 and 4-7: wave w and w+4 sit on the same SIMD) that are held HALF A TILE APART by barriers:
//
//     leader   :  M(0) | V(0) | M(1) | V(1) | ...          M(t) = P V of tile t-1, then K Q^T of tile t   (16 MFMAs)
//     follower :   --  | M(0) | V(0) | M(1) | ...          V(t) = online softmax of tile t                (~140 VALU)
//
// so at any time one wave of a SIMD feeds the matrix pipe while its partner runs the softmax VALU stream.  The two
// phases are the same length by construction (16 x 32 MFMA cycles vs ~140 VALU incl. 32 exp).
//   * K / V chunks (128 keys) arrive by LDS-DMA (buffer_load ... lds) into a 3-deep ring, issued ~6 tile-times ahead;
//     K rows are stored unpadded with the 16-byte chunk XOR-swizzled on the SOURCE address (conflict-free
//     ds_read_b128), V in the 16-channel sub-tile image that ds_read_b64_tr_b16 transposes on the way out;
//     rows past T come back as zeros from the buffer descriptor;
//   * textbook online softmax: the (rare, deferred) move of a query's reference point rescales O in the softmax phase,
//     where O is complete (P V of every earlier tile has been issued in an earlier matrix phase);
//   * row sums are kept per half-wave and combined once at the end.
#include "../../unopose_amd/csrc/common.h"
#include <cstdlib>

namespace unopose {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int PP_CHUNK = 128;                      // keys per LDS chunk
constexpr int PP_KBYTES = PP_CHUNK * 128;          // K image: [key][64 ch] bf16, 16-byte chunks swizzled by key
constexpr int PP_VSUB = PP_CHUNK * 32 + 128;       // bytes per V sub-tile [key][16 ch] + bank skew
constexpr int PP_BUF = PP_KBYTES + 4 * PP_VSUB;    // 33 280 B per chunk buffer
constexpr int PP_NBUF = 3;
constexpr float PP_DEFER = 8.f;                    // log2 of the largest P the deferred reference point lets through

__device__ __forceinline__ u16 pp_f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (u16)(u >> 16);
}

__device__ __forceinline__ void pp_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's LDS traffic is done; LDS-DMA (vmcnt) stays in flight
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// qkv: (B, T, 3, H, 64) bf16; out: (B, T, H*64) bf16.  One workgroup = 8 waves x 64 queries of one (image, head).
__global__ __launch_bounds__(512, 1) __attribute__((amdgpu_waves_per_eu(2, 2))) void vit_attn_pp_kernel(
    const u16 *__restrict__ qkv, int T, int H, int BH, int nq, float scale_log2e, u16 *__restrict__ out) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int bh = (slot / nq) * 8 + xcd, qblk = slot % nq;
  if (bh >= BH) return;
  const int b = bh / H, h = bh % H;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool follower = wave >= 4;
  const int q0 = (qblk * 8 + wave) * 64;
  const bool active = q0 < T;  // inactive waves still take part in the DMA and in every barrier
  const int col = lane & 31, hb = lane >> 5;
  const int C3 = 3 * H * 64;
  const u16 *base = qkv + (size_t)b * T * C3;

  bf16x8 qf[2][4];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int tq = min(q0 + qb * 32 + col, T - 1);
    const u16 *qp = base + (size_t)tq * C3 + h * 64 + hb * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[qb][ks] = *reinterpret_cast<const bf16x8 *>(qp + ks * 16);
  }
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qf[qb][ks]));  // Q loads complete here, not inside the loop

  // ---- LDS-DMA of one chunk: 16 K pieces (8 keys x 128 B) + 16 V pieces (32 keys x 32 B of one sub-tile); 4 per wave
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)((size_t)T * C3 * 2), 0x00020000);
  uint32_t koff[2], voff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = wave * 2 + i;
    const int key = p * 8 + (lane >> 3), c = (lane & 7) ^ ((key >> 1) & 7);
    koff[i] = (uint32_t)((key * C3 + H * 64 + h * 64 + c * 8) * 2);
    const int st = p >> 2, vkey = (p & 3) * 32 + (lane >> 1);
    voff[i] = (uint32_t)((vkey * C3 + 2 * H * 64 + h * 64 + st * 16 + (lane & 1) * 8) * 2);
  }
  // The DMA is issued from inline asm: hipcc's wait-count pass cannot prove that a ds_read does not alias an LDS-DMA in
  // flight (one dynamic LDS array, run-time ring slot) and would drain the ring with s_waitcnt vmcnt(0) in front of every
  // matrix phase.  The waits of the protocol below are placed by hand (counted vmcnt), M0 is saved and restored.
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
  auto dma16 = [&](uint32_t lds_byte, uint32_t vo, int so) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_byte), "v"(vo), "s"(rs), "s"(so)
                 : "memory");
  };
  auto dma_chunk = [&](int c) {
    const uint32_t buf = lds0 + (uint32_t)((c % PP_NBUF) * PP_BUF);
    const int soff = c * PP_CHUNK * C3 * 2;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int p = wave * 2 + i;
      dma16(buf + p * 1024, koff[i], soff);
      dma16(buf + PP_KBYTES + (p >> 2) * PP_VSUB + (p & 3) * 1024, voff[i], soff);
    }
  };

  // ---- per-lane fragment addresses inside a chunk buffer
  const int kx = (col >> 1) & 7;
  uint32_t kfo[4];  // K fragment of k-step ks: key row (kt + col), 16-byte chunk (2 ks + hb) ^ kx
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) kfo[ks] = (uint32_t)(col * 128 + ((((ks << 1) | hb) ^ kx) << 4));
  // V transpose read: sub-tile (lane >> 4) & 1 (+2 for the upper 32 channels), key row 4 hb + ((lane & 15) >> 2), 8-byte piece lane & 3
  const uint32_t vlo = (uint32_t)(PP_KBYTES + ((lane >> 4) & 1) * PP_VSUB + (4 * hb + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8);

  f32x16 o[2][2], s[2];
  float m_run[2], l_run[2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    m_run[qb] = -3e38f;
    l_run[qb] = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[qb][t][r] = 0.f;
  }
  union PF { bf16x8 v; uint32_t w[4]; } pf[2][2];

  const int ntiles = (T + 31) >> 5, nchunks = (T + PP_CHUNK - 1) / PP_CHUNK;
  auto tile_buf = [&](int t) { return smem + ((t >> 2) % PP_NBUF) * PP_BUF; };

  // fragment registers: filled in the softmax phase (LDS latency under the VALU stream), consumed by the next matrix phase
  bf16x8 kfr[4];
  union VF { bf16x8 v; s16x4 h4[2]; } vfr[2][2];
  auto load_k = [&](int t) {
    const char *kb = tile_buf(t) + (t & 3) * (32 * 128);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kfr[ks] = *reinterpret_cast<const bf16x8 *>(kb + kfo[ks]);
  };
  auto load_v = [&](int t) {
    const char *vb = tile_buf(t) + vlo + (t & 3) * (32 * 32);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const char *vp = vb + tt * 2 * PP_VSUB + s2 * (16 * 32);
        vfr[s2][tt].h4[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(vp));
        vfr[s2][tt].h4[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(vp + 8 * 32));
      }
  };
  // S^T = K Q^T for both query blocks (8 MFMAs) from the prefetched K fragments
  auto qk = [&]() {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 kf = kfr[ks];
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        if (ks == 0) {
          f32x16 z;
#pragma unroll
          for (int r = 0; r < 16; ++r) z[r] = 0.f;
          s[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[qb][ks], z, 0, 0, 0);
        } else {
          s[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[qb][ks], s[qb], 0, 0, 0);
        }
      }
    }
  };
  // O^T += V^T P^T of tile t (8 MFMAs); P in the permuted key order the V fragment reads follow
  auto pv = [&]() {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) o[qb][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[s2][tt].v, pf[qb][s2].v, o[qb][tt], 0, 0, 0);
  };
  // online softmax of tile t: scores -> packed P, running reference point / sum
  auto softmax = [&](int t) {
    if (t == ntiles - 1 && (T & 31)) {  // the one partial tile: keys >= T are masked out
      const int k0 = t * 32;
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (k0 + (r & 3) + 8 * (r >> 2) + 4 * hb >= T) s[qb][r] = -3e38f;
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      float mx = s[qb][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[qb][r]);
      const auto sm = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      mx = fmaxf(__uint_as_float(sm[0]), __uint_as_float(sm[1])) * scale_log2e;
      const bool grow = mx > m_run[qb] + PP_DEFER;
      if (__builtin_expect(__any(grow), 0)) {  // first tile, then rare: move the reference point of the queries that grew
        const float m_new = grow ? mx : m_run[qb];
        const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_new);
        m_run[qb] = m_new;
        l_run[qb] *= alpha;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[qb][tt][r] *= alpha;
      }
      const float m_use = m_run[qb];
      float ls = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[qb][r] = __builtin_amdgcn_exp2f(fmaf(s[qb][r], scale_log2e, -m_use));
        ls += s[qb][r];
      }
      l_run[qb] += ls;  // per half-wave partial sum (the halves hold different keys); combined once at the end
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int e = 0; e < 4; ++e) pf[qb][s2].w[e] = cvt_pk_bf16_f32(s[qb][s2 * 8 + 2 * e], s[qb][s2 * 8 + 2 * e + 1]);
    }
  };

  // ---- prologue: chunks 0 and 1 on chip
  dma_chunk(0);
  if (nchunks > 1) dma_chunk(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  pp_barrier();

  // ---- the half-tile ping-pong (every wave executes 2 * ntiles + 1 barriers)
  if (follower) pp_barrier();
  if (active) {
    load_k(0);
    qk();
  }
  for (int t = 0; t < ntiles; ++t) {
    pp_barrier();
    if (active) {
      load_v(t);  // fragments of the NEXT matrix phase: their LDS latency hides under this phase's VALU stream
      if (t + 1 < ntiles) load_k(t + 1);
      if (PP_ABL != 2) softmax(t);
    }
    // DMA protocol (h = half-steps; leader runs V(t) at h = 2t+1, M(t+1) at 2t+2, the follower one later):
    //   issue  chunk c+2 at the start of the own M(4c+1): its ring slot held chunk c-1, last read by P V(4c-1) in M(4c)
    //          -- leader h = 8c, follower h = 8c+1 -- and this point is h >= 8c+2, behind the barrier that ends 8c+1;
    //   wait   for chunk c+2 at the end of the own V(4c+6) (h = 8c+13 / 8c+14), ~6 tile-times after the issue; the 4
    //          pieces of chunk c+3 issued meanwhile may stay in flight (counted vmcnt);
    //   read   first in K Q^T(4c+8) at h = 8c+16, two barriers after the later of the waits.
    if ((t & 3) == 2) {
      if ((t >> 2) + 2 < nchunks) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    pp_barrier();
    if ((t & 3) == 0 && (t >> 2) + 2 < nchunks) dma_chunk((t >> 2) + 2);
#ifndef PP_ABL
#define PP_ABL 0  // scripts/ubench: 1 = no MFMAs, 2 = no softmax VALU
#endif
    if (active && PP_ABL != 1) {
      pv();
      if (t + 1 < ntiles) qk();
    }
  }
  if (!follower) pp_barrier();
  pp_barrier();  // every wave is done with the K / V ring: it becomes the output staging buffer

  if (!active) return;
  // ---- combine the half-wave sums, normalise, transpose through LDS, store token rows
  u16 (*Ot)[32][72] = reinterpret_cast<u16 (*)[32][72]>(smem);
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run[qb]), __float_as_uint(l_run[qb]), false, false);
    const float inv = 1.f / (__uint_as_float(sw[0]) + __uint_as_float(sw[1]));
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hb;
        Ot[wave][col][c] = pp_f2bf(o[qb][t][r] * inv);
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

// entry point used by scripts/ubench/pp_abl.py
int unopose_vit_attention_pp(const void *qkv, int B, int T, int H, void *out, hipStream_t stream) {
  static bool attr_set = false;
  const size_t lds = (size_t)PP_NBUF * PP_BUF;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&vit_attn_pp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess) {
      set_error("vit_attention: cannot reserve %zu bytes of LDS", lds);
      return UNOPOSE_ELAUNCH;
    }
    attr_set = true;
  }
  const int BH = B * H, nq = cdiv(T, 512);
  const float scale_log2e = 0.125f * 1.4426950408889634f;
  const long blocks = (long)cdiv(BH, 8) * nq * 8;
  hipLaunchKernelGGL(vit_attn_pp_kernel, dim3((unsigned)blocks), dim3(512), lds, stream, (const u16 *)qkv, T, H, BH, nq, scale_log2e,
                     (u16 *)out);
  return check_launch("vit_attention");
}
