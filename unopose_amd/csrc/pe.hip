// Fused positional-encoding branch of the fine matcher for gfx950 (C ABI part 2).
//
// Replaces, per scale, QueryAndLRFGroup -> SharedMLP[6,32,64,128] -> max over neighbours
// (core/unopose/model/oneref_predator_fine_point_matching.py:167-174,
//  core/unopose/model/pointnet2/pointnet2_utils.py:429-584, pointnet2/pytorch_utils.py:25-132).
// The reference materialises (B,6,N,S), (B,32,N,S), (B,64,N,S) and (B,128,N,S) fp32 tensors in HBM
// (268 MB for the last one alone at S=256, B=1).  Here ONE wavefront owns one centre: ball-query
// compaction into an LDS neighbour list, the per-point local reference frame (register Jacobi), and the
// three 1x1-conv layers (BatchNorm folded) chained on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32, exact fp32) with NO data movement between layers: the layers are computed
// transposed, D[out_ch][neighbour] = W[out_ch][k] * X[k][neighbour], so a layer's accumulator registers
// are directly the next layer's B operand (the k order inside the contraction is permuted to match the
// C/D register layout, and the weights are staged in LDS in that permuted order).  Only the (B,N,128)
// max-pooled result reaches HBM.
#include "common.h"
#include "jacobi3.h"

#ifndef PE_SCAN_STEPS
#define PE_SCAN_STEPS 4  // 64-candidate steps of the ball query per loop trip (A/B: scripts/build_variant.py -DPE_SCAN_STEPS=1)
#endif
#ifndef PE_GRID
#define PE_GRID 1  // uniform grid for the bf16x3 kernel's ball query (A/B: -DPE_GRID=0 = the index-order scan over the whole cloud)
#endif
#ifndef PE_ABL
#define PE_ABL 0  // timing probes (scripts/ubench/pe_ab.py; wrong results): 1 no MLP tiles, 2 no frame (eigen-solver, sign vote, x axis), 3 ball query over 64 points only, 4 no eigen-solver
#endif

namespace unopose {

typedef unsigned short u16;

// channel held by accumulator register r of half-wave h in the 32x32 C/D layout
__device__ __forceinline__ int cd_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// Ball query (pointnet2 ball_query_gpu.cu:14-49 semantics) into the wave's LDS neighbour list and the
// local reference frame of LRF_batch (pointnet2_utils.py:436-481) for one centre; wave-collective.
// `cand` / `ncand`: optional list of candidate indices IN INDEX ORDER that is known to contain every point
// within the radius (the neighbour list a larger-radius pass of the same cloud wrote, see cand_out below);
// ncand < 0 = scan the whole cloud.  Returns the number of points inside the radius if the list holds them
// all, a value > S otherwise.
// ---- Round 5: a uniform grid for the ball query of the bf16x3 kernel.  The scan tests all N points of the cloud per centre although
// ~75 of 2048 lie inside the radius; cells of edge >= 1.001 radius (at most PE_GDIM per axis: larger clouds get larger cells) leave the 27
// cells around the centre's, and cell ids run along x, so those are NINE contiguous runs of the cell-sorted point list: ~9 steps of 64
// candidates instead of 32.  The reference's order (the first S hits BY INDEX) is kept exactly: hits set bits of an N-bit map in LDS
// (ds_or), and the list is read off the map in index order (popcount prefix over the lanes' words) -- same hit test on the same
// coordinates, same list, bit-identical outputs.
constexpr int PE_GDIM = 8, PE_GCELLS = PE_GDIM * PE_GDIM * PE_GDIM;
struct PeGrid {
  float ox, oy, oz, ihx, ihy, ihz;  // origin, inverse cell edges
  int nx, ny, nz;
  const u16 *start;  // [PE_GCELLS + 1]: first slot of a cell in `order`
  const u16 *order;  // [N] point ids sorted by cell
  uint32_t *bits;    // this wave's map, ceil(N / 32) words
};
__device__ __forceinline__ int pe_cell(float v, float o, float ih, int n) { return max(0, min(n - 1, (int)((v - o) * ih))); }

template <typename NT>
__device__ __forceinline__ int pe_centre_frame(const float *sx, const float *sy, const float *sz, int N, int S,
                                               float radius, float r2, int lane, float cx, float cy, float cz,
                                               NT *nbr, Vec3 &xp, Vec3 &yp, Vec3 &zp, const int *cand = nullptr,
                                               int ncand = -1, const PeGrid *grid = nullptr) {
  // ---- ball query (pointnet2 ball_query_gpu.cu:14-49 semantics)
  int cnt = 0, first = 0;
  if (grid && ncand < 0 && PE_ABL != 3) {
    const int W = (N + 31) >> 5;
    for (int w = lane; w < W; w += 64) grid->bits[w] = 0u;
    // the nine runs: lane i < 9 looks up run (dy, dz) = (i % 3 - 1, i / 3 - 1)
    const int icx = pe_cell(cx, grid->ox, grid->ihx, grid->nx), icy = pe_cell(cy, grid->oy, grid->ihy, grid->ny),
              icz = pe_cell(cz, grid->oz, grid->ihz, grid->nz);
    int rs = 0, re = 0;
    if (lane < 9) {
      const int y = icy + lane % 3 - 1, z = icz + lane / 3 - 1;
      if (y >= 0 && y < grid->ny && z >= 0 && z < grid->nz) {
        const int row = (z * grid->ny + y) * grid->nx;
        rs = grid->start[row + max(icx - 1, 0)];
        re = grid->start[row + min(icx + 1, grid->nx - 1) + 1];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 1
    for (int i = 0; i < 9; ++i) {
      const int s0 = __builtin_amdgcn_readlane(rs, i), e0 = __builtin_amdgcn_readlane(re, i);
      for (int q = s0 + lane; q < e0; q += 64) {
        const int k = grid->order[q];
        const float x = sx[k], y = sy[k], z = sz[k];
        const float d2 = (cx - x) * (cx - x) + (cy - y) * (cy - y) + (cz - z) * (cz - z);
        if (d2 < r2) atomicOr(&grid->bits[k >> 5], 1u << (k & 31));
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // the list in index order: lane-owned words, exclusive prefix of their popcounts
    for (int w0 = 0; w0 < W; w0 += 64) {
      const int w = w0 + lane;
      uint32_t bits = w < W ? grid->bits[w] : 0u;
      const int pc = __builtin_popcount(bits);
      int incl = pc;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
      }
      int pos = cnt + incl - pc;
      while (bits) {
        const int bpos = __builtin_ctz(bits);
        bits &= bits - 1u;
        if (pos < S) nbr[pos] = (NT)(32 * w + bpos);
        ++pos;
      }
      cnt += __builtin_amdgcn_readlane(incl, 63);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (cnt > 0) first = nbr[0];
  } else {
  const int nscan = PE_ABL == 3 ? 64 : ncand >= 0 ? ncand : N;
  int k0 = 0;
  // PE_SCAN_STEPS (four) 64-candidate steps per trip: their LDS reads, distance tests and ballots are independent, only the list positions chain
  // through cnt (one step at a time the loop was a chain of LDS -> VALU -> ballot -> scalar latencies: a third of a launch)
  constexpr int U = PE_SCAN_STEPS;
  for (; k0 < nscan && cnt < S; k0 += 64 * U) {
    int kk[U];
    unsigned long long mask[U];
    bool hit[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int k = k0 + 64 * u + lane;
      hit[u] = false;
      if (k < nscan) {
        if (ncand >= 0) k = cand[k];
        const float x = sx[k], y = sy[k], z = sz[k];
        const float d2 = (cx - x) * (cx - x) + (cy - y) * (cy - y) + (cz - z) * (cz - z);
        hit[u] = d2 < r2;
      }
      kk[u] = k;
      mask[u] = __ballot(hit[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (mask[u]) {
        const int pre =
            (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask[u] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask[u], 0u));
        const int pos = cnt + pre;
        if (hit[u] && pos < S) nbr[pos] = (NT)kk[u];
        if (cnt == 0) {
          const int fl = __builtin_ctzll(mask[u]);  // lane of the first hit
          first = ncand >= 0 ? __builtin_amdgcn_readlane(kk[u], fl) : k0 + 64 * u + fl;
        }
        cnt += __builtin_popcountll(mask[u]);
      }
    }
  }
  if (k0 < nscan) cnt = S + 1;  // stopped early at a full list: the rest of the cloud was not looked at
  }
  for (int l = min(cnt, S) + lane; l < S; l += 64) nbr[l] = (NT)first;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  if (PE_ABL == 2) {
    xp = v3(1.f, 0.f, 0.f), yp = v3(0.f, 1.f, 0.f), zp = v3(0.f, 0.f, 1.f);
    return cnt;
  }
  // ---- local reference frame (LRF_batch, pointnet2_utils.py:436-481)
  // (measured and not kept, round 5: the three passes with the padding entries' terms as per-pass constants -- bit-identical, no faster:
  //  the frame's time is the eigen-solver and the wave reductions, not these reads)
  float a00 = 0, a01 = 0, a02 = 0, a11 = 0, a12 = 0, a22 = 0;
  for (int l = lane; l < S; l += 64) {
    const int k = nbr[l];
    const float x = cx - sx[k], y = cy - sy[k], z = cz - sz[k];
    a00 += x * x; a01 += x * y; a02 += x * z; a11 += y * y; a12 += y * z; a22 += z * z;
  }
  const float inv_s = 1.f / (float)S;
  a00 = wave_sum_f32(a00) * inv_s; a01 = wave_sum_f32(a01) * inv_s; a02 = wave_sum_f32(a02) * inv_s;
  a11 = wave_sum_f32(a11) * inv_s; a12 = wave_sum_f32(a12) * inv_s; a22 = wave_sum_f32(a22) * inv_s;
  Vec3 e0, e1, z0;
  float l0, l1, l2;
  if (PE_ABL == 4)  // (timing probe: no eigen-solver)
    z0 = v3(a00 + 1e-3f, a01, a02), z0 = scale(z0, __builtin_amdgcn_rsqf(dot(z0, z0)));
  else
    eig_sym3(a00, a01, a02, a11, a12, a22, e0, e1, z0, l0, l1, l2);
  int vote = 0;
  for (int l0i = 0; l0i < S; l0i += 64) {
    const int l = l0i + lane;
    float pr = 0.f;
    if (l < S) {
      const int k = nbr[l];
      pr = z0.x * (cx - sx[k]) + z0.y * (cy - sy[k]) + z0.z * (cz - sz[k]);
    }
    vote += __builtin_popcountll(__ballot(pr > 1e-3f)) - __builtin_popcountll(__ballot(pr < -1e-3f));
  }
  zp = vote < 0 ? scale(z0, -1.f) : z0;
  float vx = 0, vy = 0, vz = 0;
  for (int l = lane; l < S; l += 64) {
    const int k = nbr[l];
    const Vec3 xn = v3(sx[k] - cx, sy[k] - cy, sz[k] - cz);
    const float nrm = dot(zp, xn);
    const Vec3 vi = sub(xn, scale(zp, nrm));
    float alpha = radius - sqrtf(dot(xn, xn));
    alpha *= alpha;
    const float ab = alpha * (nrm * nrm);
    vx += ab * vi.x; vy += ab * vi.y; vz += ab * vi.z;
  }
  vx = wave_sum_f32(vx); vy = wave_sum_f32(vy); vz = wave_sum_f32(vz);
  const float nacc = sqrtf(vx * vx + vy * vy + vz * vz) + 1e-10f;
  xp = v3(vx / nacc, vy / nacc, vz / nacc);
  yp = cross(xp, zp);
  return cnt;
}

struct PeLds {
  // weights transposed to [k][out] so that lanes (l & 31) read consecutive floats
  float w1[6 * 32];
  float w2[32 * 64];
  float w3[64 * 128];
  float b1[32], b2[64], b3[128];
};

__global__ __launch_bounds__(256) void pe_group_mlp_max_kernel(
    const float *__restrict__ xyz, int N, float radius, int S, int cpw, const float *__restrict__ w1,
    const float *__restrict__ b1, const float *__restrict__ w2, const float *__restrict__ b2,
    const float *__restrict__ w3, const float *__restrict__ b3, float *__restrict__ out) {
  extern __shared__ float4 smem4[];
  PeLds *L = reinterpret_cast<PeLds *>(smem4);
  float *sx = reinterpret_cast<float *>(L + 1);
  float *sy = sx + N, *sz = sy + N;
  int *nbr_all = reinterpret_cast<int *>(sz + N);
  float *stage_all = reinterpret_cast<float *>(nbr_all + 4 * S);  // [4 waves][128]
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, col = lane & 31;
  int *nbr = nbr_all + wave * S;
  float *stage = stage_all + wave * 128;
  const float *P = xyz + (size_t)b * N * 3;

  for (int e = tid; e < N * 3; e += 256) {
    const float v = P[e];
    const int p = e / 3, comp = e - p * 3;
    (comp == 0 ? sx : comp == 1 ? sy : sz)[p] = v;
  }
  // weights arrive row-major [out][in] (BN already folded): transpose to [in][out]
  for (int e = tid; e < 32 * 6; e += 256) L->w1[(e % 6) * 32 + e / 6] = w1[e];
  for (int e = tid; e < 64 * 32; e += 256) L->w2[(e % 32) * 64 + e / 32] = w2[e];
  for (int e = tid; e < 128 * 64; e += 256) L->w3[(e % 64) * 128 + e / 64] = w3[e];
  if (tid < 32) L->b1[tid] = b1[tid];
  if (tid < 64) L->b2[tid] = b2[tid];
  if (tid < 128) L->b3[tid] = b3[tid];
  __syncthreads();
  const float r2 = radius * radius;

  for (int ci = 0; ci < cpw; ++ci) {
    const int j = (blockIdx.x * 4 + wave) * cpw + ci;
    if (j >= N) break;  // wave-uniform
    const float cx = sx[j], cy = sy[j], cz = sz[j];
    Vec3 xp, yp, zp;
    const int cnt_nb = pe_centre_frame(sx, sy, sz, N, S, radius, r2, lane, cx, cy, cz, nbr, xp, yp, zp);
    // Neighbour-list entries past the `cnt` points inside the radius are copies of the FIRST neighbour (ball_query_gpu.cu:14-49): their
    // MLP rows equal row 0's and cannot change the maximum -- tiles that hold nothing but such copies are skipped (bit-identical result).
    // The frame above is computed over all S entries, copies included, as the reference does.
    const int S_eff = min(S, (max(min(cnt_nb, S), 1) + 31) & ~31);  // (no point inside the radius -- a NaN centre: the list is S copies of point 0, run one tile like the reference)

    // ---- MLP over tiles of 32 neighbours, running max over tiles
    f32x16 rmax[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) rmax[t][r] = 0.f;  // post-ReLU values are >= 0
    for (int t0 = 0; t0 < S_eff; t0 += 32) {
      const int k = nbr[t0 + col];
      const float dx = sx[k] - cx, dy = sy[k] - cy, dz = sz[k] - cz;
      const Vec3 q = v3(dx / radius, dy / radius, dz / radius);
      const float lx = dot(xp, q), ly = dot(yp, q), lz = dot(zp, q);
      // layer 1: K = 6 -> 3 MFMA steps; step s contracts features (2s | 2s+1) on (lower | upper) half
      f32x16 h1;
#pragma unroll
      for (int r = 0; r < 16; ++r) h1[r] = 0.f;
      h1 = __builtin_amdgcn_mfma_f32_32x32x2f32(L->w1[(0 + half) * 32 + col], half ? dy : dx, h1, 0, 0, 0);
      h1 = __builtin_amdgcn_mfma_f32_32x32x2f32(L->w1[(2 + half) * 32 + col], half ? lx : dz, h1, 0, 0, 0);
      h1 = __builtin_amdgcn_mfma_f32_32x32x2f32(L->w1[(4 + half) * 32 + col], half ? lz : ly, h1, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) h1[r] = fmaxf(h1[r] + L->b1[cd_row(r, half)], 0.f);
      // layer 2: 32 -> 64 (2 output tiles); step r contracts channels cd_row(r,0) | cd_row(r,1)
      f32x16 h2[2];
#pragma unroll
      for (int ot = 0; ot < 2; ++ot) {
#pragma unroll
        for (int r = 0; r < 16; ++r) h2[ot][r] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          h2[ot] = __builtin_amdgcn_mfma_f32_32x32x2f32(L->w2[cd_row(r, half) * 64 + ot * 32 + col], h1[r], h2[ot], 0,
                                                       0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) h2[ot][r] = fmaxf(h2[ot][r] + L->b2[ot * 32 + cd_row(r, half)], 0.f);
      }
      // layer 3: 64 -> 128 (4 output tiles, 2 input tiles)
#pragma unroll
      for (int ot = 0; ot < 4; ++ot) {
        f32x16 h3;
#pragma unroll
        for (int r = 0; r < 16; ++r) h3[r] = 0.f;
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            h3 = __builtin_amdgcn_mfma_f32_32x32x2f32(L->w3[(it * 32 + cd_row(r, half)) * 128 + ot * 32 + col],
                                                      h2[it][r], h3, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r)
          rmax[ot][r] = fmaxf(rmax[ot][r], fmaxf(h3[r] + L->b3[ot * 32 + cd_row(r, half)], 0.f));
      }
    }
    // ---- max over the 32 neighbour lanes of each half-wave; lanes 31 / 63 hold the result
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = rmax[t][r];
        v = fmaxf(v, dpp_f32<0x111, 0xF>(v, 0.f));
        v = fmaxf(v, dpp_f32<0x112, 0xF>(v, 0.f));
        v = fmaxf(v, dpp_f32<0x114, 0xF>(v, 0.f));
        v = fmaxf(v, dpp_f32<0x118, 0xF>(v, 0.f));
        v = fmaxf(v, dpp_f32<0x142, 0xA>(v, 0.f));  // row_bcast:15 into rows 1 and 3
        if (col == 31) stage[t * 32 + cd_row(r, half)] = v;
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float *O = out + ((size_t)b * N + j) * 128;
    O[lane] = stage[lane];
    O[lane + 64] = stage[lane + 64];
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------------------------------------
// bf16 hi/lo-split variant: the same three layers on v_mfma_f32_32x32x16_bf16 with both operands split
// into hi + lo bf16 parts (a*b ~ a_hi*b_hi + a_hi*b_lo + a_lo*b_hi, ~2^-16 relative error, fp32
// accumulation): 3 bf16 MFMAs replace 8 fp32 MFMAs.  Layer chaining without data movement as above:
// accumulator registers 8s..8s+7 of a 32x32 C/D tile hold, per half-wave hb, the channels
// 16s + (e&3) + 8(e>>2) + 4hb (e = 0..7) -- exactly one 16-wide k-step of the next layer -- so the
// weights are stored with their input channels permuted to k' = 16s + 8hb + e.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ u16 pe_f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (u16)(u >> 16);
}
__device__ __forceinline__ float pe_bf2f(u16 h) { return __uint_as_float(((uint32_t)h) << 16); }

__device__ __forceinline__ int pe_kperm(int c) {  // input channel -> position in the permuted k order
  const int cc = c & 15;
  return (c & ~15) + ((cc >> 2) & 1) * 8 + (cc & 3) + 4 * (cc >> 3);
}

struct PeLdsB {
  u16 w1h[32][24], w1l[32][24];   // K padded 6 -> 16 (+8 pad)
  u16 w2h[64][40], w2l[64][40];   // K = 32 (+8 pad)
  u16 w3h[128][64], w3l[128][64]; // K = 64, 16-byte slots XOR-swizzled by (row & 7)
  float b1[32], b2[64], b3[128];
};

// gfx950 packed fp32 -> bf16 conversion (RNE): low half = cvt(a), high half = cvt(b)
__device__ __forceinline__ uint32_t pe_cvt_pk(float a, float b) {
  return cvt_pk_bf16_f32(a, b);
}

// split 8 fp32 values into packed bf16 hi / lo fragments: v ~ hi + lo
__device__ __forceinline__ void pe_split8(const float *v, bf16x8 &hi, bf16x8 &lo) {
  union { bf16x8 v; uint32_t w[4]; } H, Lo;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const uint32_t h = pe_cvt_pk(v[2 * e], v[2 * e + 1]);
    H.w[e] = h;
    const float r0 = v[2 * e] - __uint_as_float(h << 16);
    const float r1 = v[2 * e + 1] - __uint_as_float(h & 0xFFFF0000u);
    Lo.w[e] = pe_cvt_pk(r0, r1);
  }
  hi = H.v;
  lo = Lo.v;
}

#define PE_MFMA3(acc, ah, al, bh, bl)                                       \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);      \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);      \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0)

// One-time packing of the folded fp32 weights into the LDS image of the bf16x3 kernel.
__global__ __launch_bounds__(256) void pe_pack_weights_kernel(const float *__restrict__ w1, const float *__restrict__ b1,
                                                              const float *__restrict__ w2, const float *__restrict__ b2,
                                                              const float *__restrict__ w3, const float *__restrict__ b3,
                                                              PeLdsB *__restrict__ L) {
  const int tid = threadIdx.x;
  for (int e = tid; e < 32 * 16; e += 256) {
    const int o = e >> 4, kk = e & 15;
    const float v = kk < 6 ? w1[o * 6 + kk] : 0.f;
    const u16 h = pe_f2bf(v);
    L->w1h[o][kk] = h;
    L->w1l[o][kk] = pe_f2bf(v - pe_bf2f(h));
  }
  for (int e = tid; e < 64 * 32; e += 256) {
    const int o = e >> 5, c = e & 31;
    const float v = w2[e];
    const u16 h = pe_f2bf(v);
    L->w2h[o][pe_kperm(c)] = h;
    L->w2l[o][pe_kperm(c)] = pe_f2bf(v - pe_bf2f(h));
  }
  for (int e = tid; e < 128 * 64; e += 256) {
    const int o = e >> 6, c = e & 63;
    const float v = w3[e];
    const u16 h = pe_f2bf(v);
    const int kp = pe_kperm(c) ^ ((o & 7) << 3);  // swizzle the 8-element (16-byte) slot
    L->w3h[o][kp] = h;
    L->w3l[o][kp] = pe_f2bf(v - pe_bf2f(h));
  }
  if (tid < 32) L->b1[tid] = b1[tid];
  if (tid < 64) L->b2[tid] = b2[tid];
  if (tid < 128) L->b3[tid] = b3[tid];
}

#ifndef PE_WPE
#define PE_WPE 0
#endif
#ifndef PE_PREFETCH
#define PE_PREFETCH 1
#endif
#ifndef PE_NW
#define PE_NW 4  // waves per workgroup of the bf16x3 kernel: two 4-wave workgroups per CU; 8 = one per CU (measured 4 % slower: -DPE_NW=8)
#endif
__global__ __launch_bounds__(PE_NW * 64, 8 / PE_NW) void pe_group_mlp_max_bf16x3_kernel(
    const float *__restrict__ xyz, int N, float radius, int S, int cpw, const uint4 *__restrict__ image,
    const int *__restrict__ cand_in, const int *__restrict__ cand_cnt_in, int cand_stride, int *__restrict__ cand_out,
    int *__restrict__ cand_cnt_out, float *__restrict__ out, int out_ld, int out_split, int use_grid) {
  extern __shared__ float4 smem4[];
  PeLdsB *L = reinterpret_cast<PeLdsB *>(smem4);
  float *sx = reinterpret_cast<float *>(L + 1);
  float *sy = sx + N, *sz = sy + N;
  float *stage_all = sz + N;                                           // [PE_NW][128] floats; the grid build's counters and each wave's bit map live here too
  u16 *nbr_all = reinterpret_cast<u16 *>(stage_all + PE_NW * 128);    // [PE_NW][S] neighbour lists (N < 65536)
  u16 *gstart = nbr_all + PE_NW * S;                                   // grid (use_grid): [PE_GCELLS + 2] cell starts, then [N] point ids by cell
  u16 *gorder = gstart + PE_GCELLS + 2;
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, col = lane & 31;
  u16 *nbr = nbr_all + wave * S;
  float *stage = stage_all + wave * 128;
  const float *P = xyz + (size_t)b * N * 3;

  for (int e = tid; e < N * 3; e += PE_NW * 64) {
    const float v = P[e];
    const int p = e / 3, comp = e - p * 3;
    (comp == 0 ? sx : comp == 1 ? sy : sz)[p] = v;
  }
  // the LDS weight image (hi/lo bf16, permuted, swizzled; built once by pe_pack_weights_kernel) is copied
  // verbatim with coalesced 16-byte loads
  for (int e = tid; e < (int)(sizeof(PeLdsB) / 16); e += PE_NW * 64) smem4[e] = *reinterpret_cast<const float4 *>(image + e);
  __syncthreads();
  const float r2 = radius * radius;
  PeGrid grid;
  if (use_grid) {
    // ---- the cloud's grid, built by every workgroup for itself (2 k points: a few microseconds against ~150 of centres)
    static_assert(PE_NW * 128 >= PE_GCELLS, "the cell counters borrow the staging area");
    uint32_t *cnt32 = reinterpret_cast<uint32_t *>(stage_all);
    float *red = reinterpret_cast<float *>(nbr_all);  // 6 x PE_NW partial extrema (the lists are not in use yet)
    float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
    for (int p = tid; p < N; p += PE_NW * 64) {
      lo[0] = fminf(lo[0], sx[p]), hi[0] = fmaxf(hi[0], sx[p]);
      lo[1] = fminf(lo[1], sy[p]), hi[1] = fmaxf(hi[1], sy[p]);
      lo[2] = fminf(lo[2], sz[p]), hi[2] = fmaxf(hi[2], sz[p]);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lo[a] = -wave_max_f32(-lo[a]);
      hi[a] = wave_max_f32(hi[a]);
      if (lane == 0) red[wave * 6 + a] = lo[a], red[wave * 6 + 3 + a] = hi[a];
    }
    for (int c = tid; c < PE_GCELLS; c += PE_NW * 64) cnt32[c] = 0u;
    __syncthreads();
    float org[3], ih[3];
    int nd[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      float l = red[a], h = red[3 + a];
      for (int w = 1; w < PE_NW; ++w) l = fminf(l, red[w * 6 + a]), h = fmaxf(h, red[w * 6 + 3 + a]);
      const float ext = h - l, h0 = radius * 1.001f;  // (cells a little wider than the radius: a neighbour is at most ONE cell away under fp rounding)
      int n = (int)(ext / h0) + 1;
      float edge = h0;
      if (!(n <= PE_GDIM)) n = PE_GDIM, edge = fmaxf(h0, ext / (float)PE_GDIM * 1.001f);
      // (the same in every lane: kept in scalar registers)
      org[a] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(l)));
      ih[a] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(1.f / edge)));
      nd[a] = __builtin_amdgcn_readfirstlane(max(n, 1));
    }
    grid.ox = org[0], grid.oy = org[1], grid.oz = org[2], grid.ihx = ih[0], grid.ihy = ih[1], grid.ihz = ih[2];
    grid.nx = nd[0], grid.ny = nd[1], grid.nz = nd[2];
    grid.start = gstart, grid.order = gorder, grid.bits = reinterpret_cast<uint32_t *>(stage);
    auto cell_of = [&](int p) {
      return (pe_cell(sz[p], grid.oz, grid.ihz, grid.nz) * grid.ny + pe_cell(sy[p], grid.oy, grid.ihy, grid.ny)) * grid.nx +
             pe_cell(sx[p], grid.ox, grid.ihx, grid.nx);
    };
    for (int p = tid; p < N; p += PE_NW * 64) atomicAdd(&cnt32[cell_of(p)], 1u);
    __syncthreads();
    {  // exclusive scan of the PE_GCELLS counters: PE_GCELLS / (PE_NW * 64) consecutive cells per thread, wave scan, wave totals through LDS
      constexpr int CPT = PE_GCELLS / (PE_NW * 64);
      static_assert(CPT * PE_NW * 64 == PE_GCELLS, "cells divide among the threads");
      uint32_t c[CPT], sum = 0;
#pragma unroll
      for (int e = 0; e < CPT; ++e) c[e] = cnt32[tid * CPT + e], sum += c[e];
      uint32_t incl = sum;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
      }
      uint32_t *wtot = reinterpret_cast<uint32_t *>(red) + 32;
      if (lane == 63) wtot[wave] = incl;
      __syncthreads();
      uint32_t base = incl - sum;
      for (int w = 0; w < wave; ++w) base += wtot[w];
#pragma unroll
      for (int e = 0; e < CPT; ++e) {
        gstart[tid * CPT + e] = (u16)base;
        cnt32[tid * CPT + e] = base;  // the fill pass's cursor
        base += c[e];
      }
      if (tid == PE_NW * 64 - 1) gstart[PE_GCELLS] = (u16)base;
    }
    __syncthreads();
    for (int p = tid; p < N; p += PE_NW * 64) gorder[atomicAdd(&cnt32[cell_of(p)], 1u)] = (u16)p;
    __syncthreads();
  }

  for (int ci = 0; ci < cpw; ++ci) {
    const int j = (blockIdx.x * PE_NW + wave) * cpw + ci;
    if (j >= N) break;  // wave-uniform
    const float cx = sx[j], cy = sy[j], cz = sz[j];
    Vec3 xp, yp, zp;
    // The larger-radius pass of a cloud hands its neighbour list to the smaller-radius pass: that one then
    // tests <= S_large candidates instead of scanning all N points (the full scan is ~1/3 of a small-S launch).
    const int *cand = nullptr;
    int ncand = -1;
    if (cand_in) {
      ncand = cand_cnt_in[(size_t)b * N + j];  // -1: the producer's list overflowed, scan everything
      cand = cand_in + ((size_t)b * N + j) * cand_stride;
    }
    const int cnt = pe_centre_frame(sx, sy, sz, N, S, radius, r2, lane, cx, cy, cz, nbr, xp, yp, zp, cand, ncand, use_grid ? &grid : nullptr);
    if (cand_out) {
      int *co = cand_out + ((size_t)b * N + j) * S;
      for (int l = lane; l < S; l += 64) co[l] = nbr[l];
      if (lane == 0) cand_cnt_out[(size_t)b * N + j] = cnt <= S ? cnt : -1;
    }

    f32x16 rmax[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) rmax[t][r] = 0.f;  // post-ReLU values are >= 0
    // (tiles of nothing but copies of the first neighbour -- the padding of a list with fewer than S points inside the radius -- are
    //  skipped: their rows equal row 0's, the maximum cannot change; round 5)
    const int S_eff = PE_ABL == 1 ? 0 : min(S, (max(min(cnt, S), 1) + 31) & ~31);  // (cnt == 0, a NaN centre: one tile of the all-point-0 list, like the reference)
    for (int t0 = 0; t0 < S_eff; t0 += 32) {
      // keep the weight fragments in LDS (re-read per tile) instead of letting the compiler hoist ~170
      // registers of loop-invariant operands: leaves room for 2 waves / SIMD so one wave's VALU phases
      // (ball query, frame, hi/lo splits) overlap the other's MFMAs
      asm volatile("" ::: "memory");
      const int k = nbr[t0 + col];
      const float dx = sx[k] - cx, dy = sy[k] - cy, dz = sz[k] - cz;
      const Vec3 q = v3(dx / radius, dy / radius, dz / radius);
      float f[8] = {dx, dy, dz, dot(xp, q), dot(yp, q), dot(zp, q), 0.f, 0.f};
      if (half) {
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = 0.f;  // k' 8..15 of the padded first layer
      }
      bf16x8 xh, xl;
      pe_split8(f, xh, xl);
#if PE_PREFETCH
      // The weight fragments and bias values of the NEXT group of MFMAs are read from LDS (into the other of two register sets) before
      // the current group is issued: left to the compiler every k-step's two fragments were read right in front of their three MFMAs
      // -- ~20 exposed LDS round trips per tile (round 4: the tile loop was latency-bound, the matrix pipe 36 % busy).
      struct Grp {
        bf16x8 h[4], l[4];
        float4 b[4];
      } ga, gb;
      auto load_l1 = [&](Grp &g) {
        g.h[0] = *reinterpret_cast<const bf16x8 *>(&L->w1h[col][half * 8]);
        g.l[0] = *reinterpret_cast<const bf16x8 *>(&L->w1l[col][half * 8]);
#pragma unroll
        for (int q = 0; q < 4; ++q) g.b[q] = *reinterpret_cast<const float4 *>(&L->b1[8 * q + 4 * half]);
      };
      auto load_l2 = [&](Grp &g, int ot) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          g.h[ks] = *reinterpret_cast<const bf16x8 *>(&L->w2h[ot * 32 + col][ks * 16 + half * 8]);
          g.l[ks] = *reinterpret_cast<const bf16x8 *>(&L->w2l[ot * 32 + col][ks * 16 + half * 8]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) g.b[q] = *reinterpret_cast<const float4 *>(&L->b2[ot * 32 + 8 * q + 4 * half]);
      };
      auto load_l3 = [&](Grp &g, int ot) {
        const int row = ot * 32 + col;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int kp = (ks * 16 + half * 8) ^ ((row & 7) << 3);
          g.h[ks] = *reinterpret_cast<const bf16x8 *>(&L->w3h[row][kp]);
          g.l[ks] = *reinterpret_cast<const bf16x8 *>(&L->w3l[row][kp]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) g.b[q] = *reinterpret_cast<const float4 *>(&L->b3[ot * 32 + 8 * q + 4 * half]);
      };
      auto init = [&](f32x16 &h, const Grp &g) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          h[4 * q + 0] = g.b[q].x;
          h[4 * q + 1] = g.b[q].y;
          h[4 * q + 2] = g.b[q].z;
          h[4 * q + 3] = g.b[q].w;
        }
      };
      bf16x8 a1h[2], a1l[2], a2h[4], a2l[4];
      auto relu_split = [&](const f32x16 &h, bf16x8 *oh, bf16x8 *ol) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(h[s2 * 8 + e], 0.f);
          pe_split8(v, oh[s2], ol[s2]);
        }
      };
      load_l1(ga);
      load_l2(gb, 0);
      __builtin_amdgcn_sched_barrier(0);
      {
        f32x16 h1;
        init(h1, ga);
        PE_MFMA3(h1, ga.h[0], ga.l[0], xh, xl);
        load_l2(ga, 1);
        __builtin_amdgcn_sched_barrier(0);
        relu_split(h1, a1h, a1l);
      }
      {
        f32x16 h2;
        init(h2, gb);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) { PE_MFMA3(h2, gb.h[ks], gb.l[ks], a1h[ks], a1l[ks]); }
        load_l3(gb, 0);
        __builtin_amdgcn_sched_barrier(0);
        relu_split(h2, a2h, a2l);
        init(h2, ga);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) { PE_MFMA3(h2, ga.h[ks], ga.l[ks], a1h[ks], a1l[ks]); }
        load_l3(ga, 1);
        __builtin_amdgcn_sched_barrier(0);
        relu_split(h2, a2h + 2, a2l + 2);
      }
#pragma unroll
      for (int ot = 0; ot < 4; ++ot) {
        Grp &g = (ot & 1) ? ga : gb;
        // (measured and not kept: b3 + ReLU once per centre after the loop -- max_k relu(h_k + b) = relu(max_k h_k + b) -- with the
        // accumulator starting from the inline constant 0: 2021 -> 2101 us at S = 256, 1036 -> 1139 us at S = 64)
        f32x16 h3;
        init(h3, g);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { PE_MFMA3(h3, g.h[ks], g.l[ks], a2h[ks], a2l[ks]); }
        if (ot + 2 < 4) {
          load_l3(g, ot + 2);  // (into the set the MFMAs above have just read: the hardware keeps the order)
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) rmax[ot][r] = fmaxf(rmax[ot][r], h3[r]);
      }
#else
      // layer 1 (one k-step)
      // (biases are folded into the accumulator initialisation: one move instead of move + add)
      f32x16 h1;
#pragma unroll
      for (int r = 0; r < 16; ++r) h1[r] = L->b1[cd_row(r, half)];
      {
        const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(&L->w1h[col][half * 8]);
        const bf16x8 al = *reinterpret_cast<const bf16x8 *>(&L->w1l[col][half * 8]);
        PE_MFMA3(h1, ah, al, xh, xl);
      }
      bf16x8 a1h[2], a1l[2];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(h1[s2 * 8 + e], 0.f);
        pe_split8(v, a1h[s2], a1l[s2]);
      }
      // layer 2: 32 -> 64
      bf16x8 a2h[4], a2l[4];
#pragma unroll
      for (int ot = 0; ot < 2; ++ot) {
        f32x16 h2;
#pragma unroll
        for (int r = 0; r < 16; ++r) h2[r] = L->b2[ot * 32 + cd_row(r, half)];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(&L->w2h[ot * 32 + col][ks * 16 + half * 8]);
          const bf16x8 al = *reinterpret_cast<const bf16x8 *>(&L->w2l[ot * 32 + col][ks * 16 + half * 8]);
          PE_MFMA3(h2, ah, al, a1h[ks], a1l[ks]);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e)
            v[e] = fmaxf(h2[s2 * 8 + e], 0.f);
          pe_split8(v, a2h[ot * 2 + s2], a2l[ot * 2 + s2]);
        }
      }
      // layer 3: 64 -> 128, running max
#pragma unroll
      for (int ot = 0; ot < 4; ++ot) {
        f32x16 h3;
#pragma unroll
        for (int r = 0; r < 16; ++r) h3[r] = L->b3[ot * 32 + cd_row(r, half)];
        const int row = ot * 32 + col;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int kp = (ks * 16 + half * 8) ^ ((row & 7) << 3);
          const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(&L->w3h[row][kp]);
          const bf16x8 al = *reinterpret_cast<const bf16x8 *>(&L->w3l[row][kp]);
          PE_MFMA3(h3, ah, al, a2h[ks], a2l[ks]);
        }
        // running max starts at 0, so max(rmax, h3) already includes the ReLU
#pragma unroll
        for (int r = 0; r < 16; ++r) rmax[ot][r] = fmaxf(rmax[ot][r], h3[r]);
      }
#endif
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = rmax[t][r];
        v = fmaxf(v, dpp_f32<0x111, 0xF>(v, 0.f));
        v = fmaxf(v, dpp_f32<0x112, 0xF>(v, 0.f));
        v = fmaxf(v, dpp_f32<0x114, 0xF>(v, 0.f));
        v = fmaxf(v, dpp_f32<0x118, 0xF>(v, 0.f));
        v = fmaxf(v, dpp_f32<0x142, 0xA>(v, 0.f));
        if (col == 31) stage[t * 32 + cd_row(r, half)] = v;
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float *O = out + ((size_t)b * N + j) * out_ld;
    if (!out_split) {
      O[lane] = stage[lane];
      O[lane + 64] = stage[lane + 64];
    } else {
      // split layout of csrc/gemm_f32.hip: 32-channel blocks of 128 bytes [hi (32 x bf16) | lo (32 x bf16)]; this scale's 128
      // channels are 4 blocks = 128 dwords, two per lane
      uint32_t *O32 = reinterpret_cast<uint32_t *>(O);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int d = lane + 64 * q, blk = d >> 5, w = d & 31, k = blk * 32 + (w & 15) * 2;
        const float v0 = stage[k], v1 = stage[k + 1];
        const uint32_t h = cvt_pk_bf16_f32(v0, v1);
        O32[d] = (w & 16) ? cvt_pk_bf16_f32(v0 - __uint_as_float(h << 16), v1 - __uint_as_float(h & 0xffff0000u)) : h;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}
#undef PE_MFMA3

}  // namespace unopose

using namespace unopose;

extern "C" {

int unopose_pe_image_bytes(void) { return (int)sizeof(PeLdsB); }

int unopose_pe_pack_weights(const float *w1, const float *b1, const float *w2, const float *b2, const float *w3,
                            const float *b3, void *image, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(w1 && b1 && w2 && b2 && w3 && b3 && image, "pe_pack_weights: null pointer");
  hipLaunchKernelGGL(pe_pack_weights_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, w1, b1, w2, b2, w3, b3,
                     (PeLdsB *)image);
  return check_launch("pe_pack_weights");
}

int unopose_pe_group_mlp_max_packed(const float *xyz, int B, int N, float radius, int nsample, const void *image,
                                    float *out, unopose_stream_t stream) {
  return unopose_pe_group_mlp_max_packed_cand(xyz, B, N, radius, nsample, image, nullptr, nullptr, 0, nullptr, nullptr,
                                              out, stream);
}

int unopose_pe_group_mlp_max_packed_cand(const float *xyz, int B, int N, float radius, int nsample, const void *image,
                                         const int *cand_in, const int *cand_cnt_in, int cand_stride, int *cand_out,
                                         int *cand_cnt_out, float *out, unopose_stream_t stream) {
  return unopose_pe_group_mlp_max_packed_out(xyz, B, N, radius, nsample, image, cand_in, cand_cnt_in, cand_stride, cand_out,
                                             cand_cnt_out, out, 128, 0, stream);
}

int unopose_pe_group_mlp_max_packed_out(const float *xyz, int B, int N, float radius, int nsample, const void *image,
                                        const int *cand_in, const int *cand_cnt_in, int cand_stride, int *cand_out,
                                        int *cand_cnt_out, void *out, int out_ld, int out_split, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(xyz && image && out, "pe_group_mlp_max_packed: null pointer");
  UNOPOSE_REQUIRE(out_ld >= 128 && (out_split == 0 || out_split == 1), "pe_group_mlp_max_packed: bad output stride / mode");
  UNOPOSE_REQUIRE((cand_in == nullptr) == (cand_cnt_in == nullptr) && (cand_out == nullptr) == (cand_cnt_out == nullptr) &&
                      (!cand_in || cand_stride >= 1),
                  "pe_group_mlp_max_packed: candidate list and its counts go together");
  UNOPOSE_REQUIRE(B >= 0 && N >= 1 && nsample >= 32 && nsample % 32 == 0 && B <= 65535,
                  "pe_group_mlp_max_packed: nsample must be a positive multiple of 32 (got %d)", nsample);
  if (B == 0) return UNOPOSE_OK;
  // cloud (3 N floats) + staging + 16-bit neighbour lists; the ball query's grid (cell starts + the cell-sorted point ids) when the
  // cloud's bit map fits a wave's staging row and the whole still leaves room for two workgroups per CU or the cloud is large anyway
  size_t lds = sizeof(PeLdsB) + ((size_t)3 * N + PE_NW * 128) * 4 + (size_t)PE_NW * nsample * 2;
  UNOPOSE_REQUIRE(N < 65536 && lds <= 160 * 1024, "pe_group_mlp_max_packed: N=%d nsample=%d exceed the LDS tile", N, nsample);
  const size_t grid_bytes = ((size_t)PE_GCELLS + 2 + (size_t)N) * 2;
  const int use_grid = PE_GRID && radius > 0.f && N <= 4096 && N >= 256 && lds + grid_bytes <= 160 * 1024 && (lds + grid_bytes <= 80 * 1024 || lds > 80 * 1024);
  if (use_grid) lds += grid_bytes;
  lds = (lds + 15) & ~(size_t)15;
  static bool opt[64];
  if (lds_optin(opt, (const void *)pe_group_mlp_max_bf16x3_kernel, 160 * 1024, "pe_group_mlp_max_packed") != UNOPOSE_OK) return UNOPOSE_ELAUNCH;
  const long centres = (long)B * N;
  const int cpw = centres >= 65536 ? 16 : centres >= 32768 ? 8 : centres >= 8192 ? 4 : centres >= 2048 ? 2 : 1;
  dim3 grid(cdiv(N, PE_NW * cpw), B);
  hipLaunchKernelGGL(pe_group_mlp_max_bf16x3_kernel, grid, dim3(PE_NW * 64), lds, (hipStream_t)stream, xyz, N, radius, nsample,
                     cpw, (const uint4 *)image, cand_in, cand_cnt_in, cand_stride, cand_out, cand_cnt_out, (float *)out, out_ld, out_split, use_grid);
  return check_launch("pe_group_mlp_max_packed");
}

int unopose_pe_group_mlp_max(const float *xyz, int B, int N, float radius, int nsample, const float *w1,
                             const float *b1, const float *w2, const float *b2, const float *w3, const float *b3,
                             int bf16x3, float *out, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(xyz && w1 && b1 && w2 && b2 && w3 && b3 && out, "pe_group_mlp_max: null pointer");
  UNOPOSE_REQUIRE(B >= 0 && N >= 1 && nsample >= 32 && nsample % 32 == 0 && B <= 65535,
                  "pe_group_mlp_max: nsample must be a positive multiple of 32 (got %d)", nsample);
  if (B == 0) return UNOPOSE_OK;
  const size_t lds = sizeof(PeLds) + ((size_t)3 * N + 4 * (size_t)nsample + 4 * 128) * 4;
  UNOPOSE_REQUIRE(lds <= 160 * 1024, "pe_group_mlp_max: N=%d nsample=%d exceed the LDS tile", N, nsample);
  static bool opt[64];
  if (lds_optin(opt, (const void *)pe_group_mlp_max_kernel, 160 * 1024, "pe_group_mlp_max") != UNOPOSE_OK) return UNOPOSE_ELAUNCH;
  const long centres = (long)B * N;
  const int cpw = centres >= 32768 ? 8 : centres >= 8192 ? 4 : centres >= 2048 ? 2 : 1;
  dim3 grid(cdiv(N, 4 * cpw), B);
  UNOPOSE_REQUIRE(!bf16x3, "pe_group_mlp_max: the bf16x3 form takes a packed weight image "
                           "(unopose_pe_pack_weights + unopose_pe_group_mlp_max_packed)");
  hipLaunchKernelGGL(pe_group_mlp_max_kernel, grid, dim3(256), lds, (hipStream_t)stream, xyz, N, radius, nsample, cpw,
                     w1, b1, w2, b2, w3, b3, out);
  return check_launch("pe_group_mlp_max");
}

}  // extern "C"
