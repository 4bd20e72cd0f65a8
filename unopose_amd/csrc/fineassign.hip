// Fine-stage soft assignment WITHOUT the (B, n1+1, n2+1) similarity matrix (gfx950).
//
// Replaces, for the bf16 (autocast) path, the chain  compute_feature_similarity -> compute_fine_Rt_overlap up to the
// weighted Procrustes (core/unopose/utils/model_utils.py:260-282, 527-553): the reference -- and posehead.hip's streaming
// passes -- write the fp32 similarity (16.8 MB per pair, 537 MB at 32 pairs) and read it back five times.  The matrix is
// only ever reduced along rows or columns, and recomputing a 32 x 32 tile of it from the 256-wide normalised features
// costs 16 MFMAs, less than reading it from HBM, so nothing of size n1 x n2 is stored here:
//
//   x_ij = f1_i . f2_j   (f1 carries the 1/temp factor; both bf16, fp32 accumulation: the same product the bf16 bmm made)
//   e_ij = exp(x_ij - shift)            shift = a bound of |x| (1/temp): softmax is shift-invariant, so no running maxima
//   rsum_i = sum_j e_ij,  csum_j = sum_i e_ij                                                  (pass 0, both directions)
//   a_ij = (e_ij / rsum_i) (e_ij / csum_j) s1_i s2_j                                           (s*_0 = 1: background)
//   w2_j = [max_{i>=1} a_ij > a_0j]                                                            (pass 1, columns own)
//   w1_i = [max_{j>=1} a_ij > a_i0];  weight_i = w1_i sum_{j>=1} a_ij w2_j;  pred_i = sum a_ij w2_j q_j / (weight_i + 1e-6)
//                                                                                              (pass 2, rows own)
//
// One kernel, three modes.  A workgroup OWNS 256 indices (>= 1) of one side: 8 wavefronts x 32, each lane of a wavefront
// one index (both half-waves the same 32: the MFMA output puts the owned index in the lane and 16 indices of the swept side
// in the registers, so every reduction along the sweep is a per-lane running scalar -- no cross-lane work until the end).
// The owned rows' 256-wide features stay in registers as MFMA B operands (64 VGPRs); the swept side streams through LDS in
// tiles of 32 rows x 256 (16 KiB, double-buffered, LDS-DMA, the GEMM's XOR-swizzled image), read as A operands.  Index 0 of
// the swept side (the background token) is the first element of tile 0; index 0 of the owning side belongs to nobody: its
// statistics are the per-workgroup partial sums the other direction's pass 0 leaves behind plus the corner x_00, summed in
// a fixed order by whoever needs them (deterministic: no atomics).
#include <stdlib.h>

#include "common.h"

namespace unopose {

typedef __bf16 fa_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

constexpr int FA_D = 256;                 // feature width (out_proj)
constexpr int FA_OWN = 256;               // owned indices per workgroup
constexpr int FA_TILE = 32 * FA_D * 2;    // one swept tile: 16 KiB
constexpr float FA_L2E = 1.4426950408889634f;

struct FAParams {
  const u16 *f[2];         // [0]: (B, n[0], 256) rows side (scaled by 1/temp), [1]: (B, n[1], 256) columns side
  int n[2];                // n1 + 1, n2 + 1
  int nblk[2];             // workgroups owning side d
  float shift_l2e;         // shift * log2(e)
  float *rs[2];            // (B, n[d]) reciprocal sums of side d (entry 0 unused)
  float *part[2];          // (B, nblk[d]) partial sums of the SWEPT side's index 0 seen by the owners of side d
  const float *score[2];   // (B, n[d] - 1)
  float *w[2];             // (B, n[d] - 1) labels
  const float *pts2;       // (B, n[1] - 1, 3)
  float *weight, *pred;    // (B, n[0] - 1), (B, n[0] - 1, 3)
  int old_grid;            // A/B switch (UNOPOSE_FA_OLD_GRID=1): block index fastest, as first written
};

__device__ __forceinline__ float fa_half_sum(float v) { return v + __shfl_xor(v, 32); }
__device__ __forceinline__ float fa_half_max(float v) { return fmaxf(v, __shfl_xor(v, 32)); }

// MODE 0: sums (direction = blockIdx.z); MODE 1: labels of the columns side (columns own); MODE 2: labels of the rows
// side + correspondences (rows own).
template <int MODE>
__global__ __launch_bounds__(512) void fine_assign_kernel(const FAParams p) {
  extern __shared__ __attribute__((aligned(1024))) char fa_smem[];
  const int dir = MODE == 0 ? (int)blockIdx.z : (MODE == 1 ? 1 : 0);
  // grid.x = pair, grid.y = block: workgroup ids of one pair are B apart, i.e. (B % 8 == 0) on ONE XCD -- the 8 owners of a pair
  // re-read the same swept side (1 MB) and share it in that XCD's L2 instead of fetching it 8 times (PMC: 601 -> MB per launch)
  const int blk = p.old_grid ? blockIdx.x : blockIdx.y, b = p.old_grid ? blockIdx.y : blockIdx.x;
  if (blk >= p.nblk[dir]) return;
  const int NO = p.n[dir], NS = p.n[1 - dir];
  const u16 *own = p.f[dir] + (size_t)b * NO * FA_D;
  const u16 *swp = p.f[1 - dir] + (size_t)b * NS * FA_D;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int nt = (NS + 31) >> 5, NSP = nt * 32;
  float *arr = reinterpret_cast<float *>(fa_smem + 2 * FA_TILE);  // MODE 1: [rs | sc]; MODE 2: [rs | sc | g | qx | qy | qz], NSP each
  __shared__ float red[8];

  // ---- swept tiles by LDS-DMA (issued from inline asm: see gemm.hip).  Tile = 4 K-slices of 64 (4 KiB each) x 4 groups
  //      of 8 rows (1 KiB = one DMA); wave w moves slice w >> 1, groups 2 (w & 1) + {0, 1}.
  const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc((void *)swp, 0, NS * FA_D * 2, 0x00020000);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)fa_smem;
  uint32_t voff[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int row = ((wave & 1) * 2 + k) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    voff[k] = (uint32_t)((row * FA_D + (wave >> 1) * 64 + c * 8) * 2);
  }
  auto stage = [&](int t) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const uint32_t dst = lds0 + (uint32_t)((t & 1) * FA_TILE + (wave >> 1) * 4096 + ((wave & 1) * 2 + k) * 1024);
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "s"(dst), "v"(voff[k]), "s"(srs), "s"(t * (32 * FA_D * 2))
                   : "memory");
    }
  };
  stage(0);

  // ---- owned rows: B operands in registers
  const int i = 1 + blk * FA_OWN + wave * 32 + l31;
  const bool valid = i < NO;
  const int ic = valid ? i : NO - 1;
  fa_bf16x8 bfr[16];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) bfr[ks] = *reinterpret_cast<const fa_bf16x8 *>(own + (size_t)ic * FA_D + ks * 16 + hi * 8);

  // ---- per-index data of the swept side in LDS (modes 1, 2)
  float o_rs = 0.f, o_sc = 0.f;
  if (MODE != 0) {
    // x_00 and the background entry of the swept side's reciprocal sums
    const uint2 ua = *reinterpret_cast<const uint2 *>(own + lane * 4), ub = *reinterpret_cast<const uint2 *>(swp + lane * 4);
    float d = 0.f;
    d = fmaf(__uint_as_float(ua.x << 16), __uint_as_float(ub.x << 16), d);
    d = fmaf(__uint_as_float(ua.x & 0xffff0000u), __uint_as_float(ub.x & 0xffff0000u), d);
    d = fmaf(__uint_as_float(ua.y << 16), __uint_as_float(ub.y << 16), d);
    d = fmaf(__uint_as_float(ua.y & 0xffff0000u), __uint_as_float(ub.y & 0xffff0000u), d);
    d = wave_sum_f32(d);
    float bg = __builtin_amdgcn_exp2f(fmaf(d, FA_L2E, -p.shift_l2e));
    const float *pp = p.part[dir] + (size_t)b * p.nblk[dir];
    for (int k = 0; k < p.nblk[dir]; ++k) bg += pp[k];
    const float bg_rs = 1.f / bg;
    const float *s_rs = p.rs[1 - dir] + (size_t)b * NS, *s_sc = p.score[1 - dir] + (size_t)b * (NS - 1);
    for (int j = tid; j < NSP; j += 512) {
      const float r = j == 0 ? bg_rs : (j < NS ? s_rs[j] : 0.f);
      const float s = j == 0 ? 1.f : (j < NS ? s_sc[j - 1] : 0.f);
      arr[j] = r;
      arr[NSP + j] = s;
      if (MODE == 2) {
        const bool fg = j >= 1 && j < NS;
        const float w2 = fg ? p.w[1][(size_t)b * (NS - 1) + j - 1] : 0.f;
        arr[2 * NSP + j] = r * s * w2;  // column 0 takes no part in the correspondences (w2 = 0 there)
        const float *q = p.pts2 + ((size_t)b * (NS - 1) + (fg ? j - 1 : 0)) * 3;
        arr[3 * NSP + j] = fg ? q[0] : 0.f;
        arr[4 * NSP + j] = fg ? q[1] : 0.f;
        arr[5 * NSP + j] = fg ? q[2] : 0.f;
      }
    }
    o_rs = p.rs[dir][(size_t)b * NO + ic];
    o_sc = p.score[dir][(size_t)b * (NO - 1) + ic - 1];
  }

  // ---- fragment addresses (gemm.hip's image: row r, 16-byte chunk c of a 128-byte K-slice row at
  //      (r >> 3) * 1024 + (r & 7) * 128 + ((c ^ ((r >> 1) & 7)) << 4))
  uint32_t fr[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) fr[k] = (uint32_t)((l31 >> 3) * 1024 + (l31 & 7) * 128 + ((((k << 1) | hi) ^ ((l31 >> 1) & 7)) << 4));

  float sum = 0.f, bgv = 0.f;           // MODE 0
  float best = -1.f, a0 = 0.f;          // MODE 1, 2
  float sw = 0.f, px = 0.f, py = 0.f, pz = 0.f;  // MODE 2

  for (int t = 0; t < nt; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // tile t has landed; everyone is done with tile t - 1
    if (t + 1 < nt) stage(t + 1);
    const char *tb = fa_smem + (t & 1) * FA_TILE;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const fa_bf16x8 a = *reinterpret_cast<const fa_bf16x8 *>(tb + (ks >> 2) * 4096 + fr[ks & 3]);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bfr[ks], acc, 0, 0, 0);
    }
    // acc[4 q + e] = x(owned index of this lane, swept index 32 t + 8 q + 4 hi + e)
    const int jb = t * 32 + 4 * hi;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float ev[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) ev[e] = __builtin_amdgcn_exp2f(fmaf(acc[4 * q + e], FA_L2E, -p.shift_l2e));
      const int j0 = jb + 8 * q;
      if (MODE == 0) {
        if (t == nt - 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) ev[e] = j0 + e < NS ? ev[e] : 0.f;
        }
        if (t == 0 && q == 0) bgv = ev[0];  // meaningful in the low half-wave only (j = 0)
        sum += (ev[0] + ev[1]) + (ev[2] + ev[3]);
      } else {
        const float4 r4 = *reinterpret_cast<const float4 *>(arr + j0), s4 = *reinterpret_cast<const float4 *>(arr + NSP + j0);
        const float rr[4] = {r4.x, r4.y, r4.z, r4.w}, ss[4] = {s4.x, s4.y, s4.z, s4.w};
        float av[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // ((softmax_row * softmax_col) * s1) * s2, the reference's multiplication order (model_utils.py:538-541)
          if (MODE == 1) av[e] = (((ev[e] * rr[e]) * (ev[e] * o_rs)) * ss[e]) * o_sc;  // swept = rows
          else av[e] = (((ev[e] * o_rs) * (ev[e] * rr[e])) * o_sc) * ss[e];            // swept = columns
        }
        if (t == 0 && q == 0) {
          if (hi == 0) {
            a0 = av[0];
            av[0] = -1.f;
          }
        }
        best = fmaxf(best, fmaxf(fmaxf(av[0], av[1]), fmaxf(av[2], av[3])));
        if (MODE == 2) {
          const float4 g4 = *reinterpret_cast<const float4 *>(arr + 2 * NSP + j0);
          const float4 x4 = *reinterpret_cast<const float4 *>(arr + 3 * NSP + j0);
          const float4 y4 = *reinterpret_cast<const float4 *>(arr + 4 * NSP + j0);
          const float4 z4 = *reinterpret_cast<const float4 *>(arr + 5 * NSP + j0);
          const float gg[4] = {g4.x, g4.y, g4.z, g4.w}, xx[4] = {x4.x, x4.y, x4.z, x4.w};
          const float yy[4] = {y4.x, y4.y, y4.z, y4.w}, zz[4] = {z4.x, z4.y, z4.z, z4.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float a = (ev[e] * ev[e]) * gg[e];
            sw += a;
            px = fmaf(a, xx[e], px);
            py = fmaf(a, yy[e], py);
            pz = fmaf(a, zz[e], pz);
          }
        }
      }
    }
  }

  if (MODE == 0) {
    const float s = fa_half_sum(sum);
    if (valid && hi == 0) p.rs[dir][(size_t)b * NO + i] = 1.f / s;  // consumers multiply by the reciprocal
    const float v = wave_sum_f32(valid && hi == 0 ? bgv : 0.f);
    if (lane == 0) red[wave] = v;
    __syncthreads();
    if (tid == 0) p.part[dir][(size_t)b * p.nblk[dir] + blk] = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
  } else {
    best = fa_half_max(best);
    const float label = best > a0 ? 1.f : 0.f;  // first-index argmax: the background wins ties
    if (MODE == 1) {
      if (valid && hi == 0) p.w[1][(size_t)b * (NO - 1) + i - 1] = label;
    } else {
      const float rfac = (o_rs * o_sc) * label;
      const float s = fa_half_sum(sw) * rfac, ax = fa_half_sum(px) * rfac, ay = fa_half_sum(py) * rfac, az = fa_half_sum(pz) * rfac;
      if (valid && hi == 0) {
        const size_t o = (size_t)b * (NO - 1) + i - 1;
        p.w[0][o] = label;
        p.weight[o] = s;
        const float inv = 1.f / (s + 1e-6f);
        p.pred[o * 3] = ax * inv;
        p.pred[o * 3 + 1] = ay * inv;
        p.pred[o * 3 + 2] = az * inv;
      }
    }
  }
}

}  // namespace unopose

using namespace unopose;

extern "C" {

int unopose_fine_assign(const void *f1, const void *f2, int B, int R, int C, int D, float shift, const float *score1,
                        const float *score2, const float *pts2, float *ws, float *w1, float *w2, float *weight, float *pred,
                        void *stream) {
  UNOPOSE_REQUIRE(f1 && f2 && score1 && score2 && pts2 && ws && w1 && w2 && weight && pred, "fine_assign: null pointer");
  UNOPOSE_REQUIRE(D == FA_D, "fine_assign: feature width must be %d, got %d", FA_D, D);
  UNOPOSE_REQUIRE(B >= 0 && R >= 2 && C >= 2 && R <= (1 << 20) && C <= (1 << 20), "fine_assign: bad sizes");
  UNOPOSE_REQUIRE(shift >= 0.f && shift <= 40.f, "fine_assign: shift (1/temp) must lie in [0, 40] for exp(2 (x - shift)) to stay in fp32 range");
  if (B == 0) return UNOPOSE_OK;
  hipStream_t s = (hipStream_t)stream;
  FAParams p;
  p.f[0] = (const u16 *)f1;
  p.f[1] = (const u16 *)f2;
  p.n[0] = R;
  p.n[1] = C;
  p.nblk[0] = cdiv(R - 1, FA_OWN);
  p.nblk[1] = cdiv(C - 1, FA_OWN);
  p.shift_l2e = shift * FA_L2E;
  p.rs[0] = ws;
  p.rs[1] = ws + (size_t)B * R;
  p.part[0] = p.rs[1] + (size_t)B * C;
  p.part[1] = p.part[0] + (size_t)B * p.nblk[0];
  p.score[0] = score1;
  p.score[1] = score2;
  p.w[0] = w1;
  p.w[1] = w2;
  p.pts2 = pts2;
  p.weight = weight;
  p.pred = pred;
  const int nsp[2] = {cdiv(R, 32) * 32, cdiv(C, 32) * 32};
  const size_t lds0 = 2 * FA_TILE, lds1 = 2 * FA_TILE + (size_t)2 * nsp[0] * 4, lds2 = 2 * FA_TILE + (size_t)6 * nsp[1] * 4;
  UNOPOSE_REQUIRE(lds1 <= 150 * 1024 && lds2 <= 150 * 1024, "fine_assign: %d x %d does not fit the LDS-resident per-index tables", R, C);
  static bool opt1[64], opt2[64];
  if (lds_optin(opt1, (const void *)fine_assign_kernel<1>, 150 * 1024, "fine_assign") != UNOPOSE_OK ||
      lds_optin(opt2, (const void *)fine_assign_kernel<2>, 150 * 1024, "fine_assign") != UNOPOSE_OK)
    return UNOPOSE_ELAUNCH;
  const int nb = p.nblk[0] > p.nblk[1] ? p.nblk[0] : p.nblk[1];
#ifdef UNOPOSE_PROBE_BUILD  // (the grid order before the XCD-aware one: A/B in probe builds only)
  static const int old_grid = getenv("UNOPOSE_FA_OLD_GRID") ? atoi(getenv("UNOPOSE_FA_OLD_GRID")) : 0;
#else
  const int old_grid = 0;
#endif
  p.old_grid = old_grid;
  if (old_grid) {
    hipLaunchKernelGGL(fine_assign_kernel<0>, dim3(nb, B, 2), dim3(512), lds0, s, p);
    hipLaunchKernelGGL(fine_assign_kernel<1>, dim3(p.nblk[1], B), dim3(512), lds1, s, p);
    hipLaunchKernelGGL(fine_assign_kernel<2>, dim3(p.nblk[0], B), dim3(512), lds2, s, p);
    return check_launch("fine_assign");
  }
  hipLaunchKernelGGL(fine_assign_kernel<0>, dim3(B, nb, 2), dim3(512), lds0, s, p);
  hipLaunchKernelGGL(fine_assign_kernel<1>, dim3(B, p.nblk[1]), dim3(512), lds1, s, p);
  hipLaunchKernelGGL(fine_assign_kernel<2>, dim3(B, p.nblk[0]), dim3(512), lds2, s, p);
  return check_launch("fine_assign");
}

}  // extern "C"
