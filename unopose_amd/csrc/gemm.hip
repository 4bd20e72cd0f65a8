// bf16 "linear" GEMM with fused epilogues for gfx950 (C ABI part 2):
//     C[M][N] = act( A[M][K] . W[N][K]^T + bias[N] ),   A / W / C bf16 row-major, bias fp32, fp32 accumulation.
// This is nn.Linear as timm's ViT blocks call it (qkv, proj, fc1 + GELU, fc2; the up-projection of
// oneref_feature_extraction.py:221) and as the matcher's transformer layers call it (transformer.py:151-193).
//
// Structure (one 512-thread workgroup per 256 x 256 output tile, K step 64, one persistent workgroup per CU):
//   * both operands are K-contiguous, so A and W tiles are the same kind of image: [256 rows][64 k] bf16.
//     Each 1-KiB piece (8 rows x 128 B, whole cache lines) is moved L2 -> LDS by ONE LDS-DMA wave instruction.  The bank
//     swizzle lives on the per-lane SOURCE address: LDS slot (row, p) holds the row's 16-byte chunk
//     c = p ^ ((row >> 1) & 7)  and the fragment reads apply the same XOR -- every ds_read_b128 lane group then covers
//     all 16 slots of the 256-byte bank row (conflict-free), and every DMA instruction still fetches full 128-byte lines;
//   * 8 waves as 2 (M) x 4 (N), 128 x 64 outputs per wave, v_mfma_f32_32x32x16_bf16 with the operands SWAPPED
//     (rows of the MFMA result = output columns n): a lane then owns 4 consecutive n of one output row, which
//     packs to 8-byte LDS writes in the epilogue;
//   * ROUND 4 -- the K loop is a HALF-TILE STREAM with COUNTED waits (never vmcnt(0) inside the stream).  A K-tile is four
//     16-KiB half-tiles in stream order  A0 (rows 0-63 of each wave row-block), B0 (columns 0-31 of each wave column-block),
//     B1, A1;  the ring is 2 K-tiles x 4 half-tile slots = 128 KiB.  A K-tile is computed in two PHASES of two 64 x 32 output
//     quadrants over the whole K-step each (16 MFMAs):  X = (A0 x B0, A0 x B1),  Y = (A1 x B1, A1 x B0),  and every phase is
//         [fragment reads of the phase | LDS-DMA pieces (X: one half-tile, Y: three) | s_waitcnt vmcnt(8) | lgkmcnt(0)]
//         s_barrier   [16 MFMAs]   s_barrier
//     Four half-tiles (8 loads per wave) stay in flight ACROSS the barriers; a half-tile is read one phase after the wait
//     that covers it and its slot is refilled one phase after its last read (reads retired before the barrier).
//     The two wave groups (wm = 0 / 1: one wave of each per SIMD) run ONE BARRIER apart, so one group's MFMA segment sits
//     beside the other group's read / DMA segment on every SIMD (ping-pong).  The guide's 4-phase form (8 MFMAs per phase, 8
//     barriers per K-tile) was built first and measured 12-16 % slower on every shape: scripts/ubench/gemm_r04_variants.hip;
//   * the stream does not stop at a tile boundary: while the last two K-tiles of a tile are computed the first six
//     half-tiles of the NEXT tile (and its bias slice, also by LDS-DMA) are issued, the epilogue stages C through the two slots
//     of the ring the stream refills last (A1 / B1 of the last K-tile's buffer, 4 KiB per wave), and the next tile's
//     phases find their operands landed.  Only a workgroup's last tile drains (counted 4 / 2 / 0);
//   * every tile starts its K walk at a tile-dependent K-tile (the sum is order-independent): concurrently running
//     tiles then touch different 128-byte columns of their panels at any instant;
//   * epilogue on the fp32 accumulators: + bias, optional GELU / ReLU / residual + LayerNorm, bf16, staged through LDS
//     (XOR-swizzled) and written as whole 128-byte row segments with NON-TEMPORAL stores when the output exceeds the L2s.
#include <stdlib.h>

#include <mutex>

#include "gemm_common.h"

namespace unopose {

#ifndef GEMM_ABL
#define GEMM_ABL 0  // scripts/ubench/gemm_var.py: 1 = no LDS-DMA in the K loop, 2 = no MFMAs, 3 = no fragment reads
#endif
#ifndef GEMM_EABL
#define GEMM_EABL 0  // epilogue ablations: 1 = no global stores, 2 = no epilogue at all (accumulators kept live), 3 = no bias / activation math
#endif
#ifndef GEMM_SAME
#define GEMM_SAME 0  // probe: every tile streams the operands of tile (0, 0) -- an all-hit L2 stream under the full K loop
#endif
#ifndef GEMM_PRIO
#define GEMM_PRIO 0  // 1 = s_setprio 1 around the MFMA segment (measured: -1..2 % with 16-MFMA segments; scripts/ubench/gemm_r04_variants.hip)
#endif
#define GEMM_BK 64
constexpr bool kMfma = GEMM_ABL != 2, kFrag = GEMM_ABL != 3, kDma = GEMM_ABL != 1;

#ifndef GEMM_ROTX
#define GEMM_ROTX 5  // K-tile rotation between XCDs (-1: spread evenly, xcd * nk / 8) and between steps
#endif
#ifndef GEMM_ROTS
#define GEMM_ROTS 3
#endif

#define GEMM_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define GEMM_WAIT_VM_(n) GEMM_WAIT_VM(n)

// LDS map (ONE __shared__ object): [0, 128 Ki) the ring: buffer b at b * 64 Ki = A image (32 Ki) | W image (32 Ki);
// then 2 x 1 Ki bias slices (tile parity); EPI 3 only: LayerNorm weight / bias (2 x 1 Ki) and the row-statistics exchange (8 Ki).
#define GEMM_LDS_BIAS (2 * GEMM_BUFBYTES)
#define GEMM_LDS_MBOX (GEMM_LDS_BIAS + 2048)  // 16 bytes: the next tile's ticket (dynamic tile scheduling)
#define GEMM_LDS_LNW (GEMM_LDS_MBOX + 16)
#define GEMM_LDS_LNB (GEMM_LDS_LNW + 1024)
#define GEMM_LDS_LNPART (GEMM_LDS_LNB + 1024)

// GATHER (grouped, row-gathered form; unopose_linear_bf16_gather): output row r of tile t is A row row_list[256 t + r]
// times the 256-row weight block of the group tile t belongs to (tile_info[1 + g] = first tile of group g, g = 0..N/256;
// tile_info[0] = number of tiles, read on the device: the host never learns it); C is (tiles * 256, 256).
// EPI 3 (N == 256 only: a row is one tile wide): C = LayerNorm(A W^T + bias + resid) * ln_w + ln_b, the post-LN glue of the
// matcher's transformer layers (transformer.py:151-193) -- the residual add and the LayerNorm run on the fp32 accumulators.
// `sched` (optional): DYNAMIC tile scheduling.  A persistent workgroup that is dispatched late -- its CU was held by a kernel of another
// stream: the 5000 -> 2048 FPS keeps 32 CUs for 1.9 ms under the ViT -- would still own its full static share of the tiles and
// double the launch's makespan (12 ViT GEMMs: 3.58 ms alone, 4.74 ms beside that FPS; scripts/ubench/gemm_beside_fps.py).  With
// `sched`, tiles are TICKETS drawn from one counter per XCD chunk (sched[0..7]; sched[8] counts finished workgroups, the last one
// zeroes the slot for its next use).  The ticket of the NEXT tile is drawn at the start of the current one by lane 0 of wave 0 with a
// returning global atomic whose result lands in v255 -- a register the compiler never allocates (amdgpu_num_vgpr(127) below: on
// gfx90a+ the attribute counts in units of 2 registers, so the compiler keeps to v0 .. v253) -- so that NO wait sits between issue and use: returning atomics retire in order
// with the wave's LDS-DMA loads, and the counted waits of the next two K-tiles retire it.  Wave 0 then posts it in LDS; all waves read
// it two K-tiles before the tile ends, when the stream needs the next tile's addresses.
template <int EPI, bool GATHER = false>  // EPI 0: bias; 1: bias + GELU; 2: bias + ReLU; 3: bias + residual + LayerNorm
__global__ __launch_bounds__(512, 1) __attribute__((amdgpu_num_vgpr(127))) void gemm_bf16_kernel(const u16 *__restrict__ A, const u16 *__restrict__ W,
                                                           const float *__restrict__ bias, u16 *__restrict__ C, int M,
                                                           int N, int K, int tiles_n, int tiles_arg, int nt_store,
                                                           const int *__restrict__ row_list = nullptr,
                                                           const int *__restrict__ tile_info = nullptr,
                                                           const u16 *__restrict__ resid = nullptr, const float *__restrict__ ln_w = nullptr,
                                                           const float *__restrict__ ln_b = nullptr, float ln_eps = 0.f, int lda = 0,
                                                           int ldw = 0, int ldc = 0, int *__restrict__ sched = nullptr) {
  // row strides in elements (0 = dense; unopose_linear_bf16_ld)
  const int LDA = lda ? lda : K, LDW = ldw ? ldw : K, LDC = ldc ? ldc : N;
  const int tiles = GATHER ? __builtin_amdgcn_readfirstlane(tile_info[0]) : tiles_arg;
  __shared__ __attribute__((aligned(1024))) char smem[GEMM_LDS_BIAS + 2048 + 16 + (EPI == 3 ? 2048 + 8192 : 0)];
  float *const lnw_lds = reinterpret_cast<float *>(smem + GEMM_LDS_LNW), *const lnb_lds = reinterpret_cast<float *>(smem + GEMM_LDS_LNB);
  float2 *const ln_part = reinterpret_cast<float2 *>(smem + GEMM_LDS_LNPART);  // [wm][mb][row][wn]: (sum, sum of squares) of 64 columns
  if (EPI == 3 && threadIdx.x < GEMM_BN) {  // visible after the first barrier of the tile loop
    lnw_lds[threadIdx.x] = ln_w[threadIdx.x];
    lnb_lds[threadIdx.x] = ln_b[threadIdx.x];
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int l31 = lane & 31, hi = lane >> 5;
  // ---- persistent, lock-stepped tile walk.  The grid is ONE workgroup per CU (gridDim.x <= 256, a multiple of 8;
  // 130 KiB of LDS admits one per CU).  Workgroup b sits on XCD b % 8 (observed dispatch rule: a SPEED assumption
  // only) and is slot b / 8 of that XCD; XCD x owns one contiguous range of the tile sequence and its slots take
  // tiles slot, slot + nslots, ... of it.  All workgroups start together and every tile costs the same, so the ~32
  // tiles an XCD has in flight are 32 CONSECUTIVE tiles walking K in lock step: a (GEMM_GM x 32/GEMM_GM) patch of the
  // output that shares GEMM_GM A panels and 32/GEMM_GM W panels K-slice by K-slice in that XCD's L2.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int cq = tiles >> 3, cr = tiles & 7;
  const int chunk_base = xcd < cr ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq, chunk_len = cq + (xcd < cr ? 1 : 0);
  const int tiles_m = tiles / tiles_n, per_group = GEMM_GM * tiles_n;
  const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc((void *)A, 0, (int)((size_t)M * LDA * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc((void *)W, 0, (int)((size_t)N * LDW * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t c_rs = __builtin_amdgcn_make_buffer_rsrc((void *)C, 0, GATHER ? 0x7fffffff : (int)((size_t)M * LDC * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc((void *)bias, 0, N * 4, 0x00020000);
  const int nk = K / GEMM_BK;

  // ---- fragment read addresses: tile row r = base + l31 (base a multiple of 32), chunk c = 2 ks + hi:
  //      byte = (r >> 3) * 1024 + (r & 7) * 128 + ((c ^ ((r >> 1) & 7)) << 4)
  const int fx = (l31 >> 1) & 7;
  uint32_t fr_off[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) fr_off[ks] = (uint32_t)((l31 >> 3) * 1024 + (l31 & 7) * 128 + ((((ks << 1) | hi) ^ fx) << 4));
  const uint32_t a_base = (uint32_t)(wm * 16384);                // A rows wm*128 ..  (+ ah * 8192 + mbl * 4096)
  const uint32_t w_base = (uint32_t)(GEMM_OPBYTES + wn * 8192);  // W rows wn*64 ..   (+ bh * 4096)
  // ---- LDS-DMA destinations of this wave's two pieces of a half-tile (within a buffer); piece i at + i * 1024
  //      A half ah: rows (wave >> 2) * 128 + ah * 64 + (wave & 3) * 16 + 8 i ..;   B half bh: rows (wave >> 1) * 64 + bh * 32 + (wave & 1) * 16 + 8 i ..
  const uint32_t a_dst = (uint32_t)((wave >> 2) * 16384 + (wave & 3) * 2048);
  const uint32_t w_dst = (uint32_t)(GEMM_OPBYTES + (wave >> 1) * 8192 + (wave & 1) * 2048);

  // per-tile DMA parameters: tile origin, K rotation, per-lane source offsets of the wave's pieces
  struct TileP {
    int m0, n0, rot;
    uint32_t a_off[4], w_off[2];  // a_off[2 ah + i]; the B half enters through the scalar offset (32 rows further)
  };
  auto tile_params = [&](int ti, int step, TileP &p) {
    const int t = chunk_base + ti;
    int tn, tm;
    if (GATHER) {
      tm = t;
      tn = 0;
      const int ng = N / GEMM_BN;
      for (int g = 1; g < ng; ++g) tn += t >= tile_info[1 + g] ? 1 : 0;  // the group of tile t (uniform scalar loads)
      tn = __builtin_amdgcn_readfirstlane(tn);
    } else {
      // tile order: groups of GEMM_GM row panels, column tiles fastest across the group
      const int mg = t / per_group, rr = t - mg * per_group;
      const int gm = min(GEMM_GM, tiles_m - mg * GEMM_GM);
      tn = rr / gm;
      tm = mg * GEMM_GM + (rr - tn * gm);
    }
    p.m0 = __builtin_amdgcn_readfirstlane(tm * GEMM_BM);
    p.n0 = __builtin_amdgcn_readfirstlane(tn * GEMM_BN);
    // per-lane byte offset in the VGPR, K-tile offset in an SGPR; rows past M (ragged last tile) fall outside the
    // descriptor -> zeros
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = (wave >> 2) * 128 + (j >> 1) * 64 + (wave & 3) * 16 + (j & 1) * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((row >> 1) & 7);
      int arow = (GEMM_SAME ? 0 : p.m0) + row;
      if (GATHER) arow = max(row_list[p.m0 + row], 0);  // padding rows of a group (-1) compute on row 0; nobody reads them
      p.a_off[j] = (uint32_t)(((size_t)arow * LDA + c * 8) * 2);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (wave >> 1) * 64 + (wave & 1) * 16 + i * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((row >> 1) & 7);
      p.w_off[i] = (uint32_t)(((size_t)((GEMM_SAME ? 0 : p.n0) + row) * LDW + c * 8) * 2);
    }
    // K-tile rotation, uniform over the tiles an XCD runs together (they must stay on the same K-slice to share it) and
    // different between XCDs / steps: the chip as a whole touches different 128-byte columns at any instant.
    // GEMM_SKEW: tiles sharing a panel start 0..SKEW-1 K-tiles apart, so a K-slice one of them has fetched is RESIDENT in L2
    // when the others ask for it
    const int skew = ((tm & 3) + tn) % (GEMM_SKEW > 1 ? GEMM_SKEW : 1);
    p.rot = __builtin_amdgcn_readfirstlane(((GEMM_ROTX < 0 ? xcd * nk / 8 : xcd * GEMM_ROTX) + step * GEMM_ROTS + skew) % nk);
  };
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
  enum { H_A0 = 0, H_B0 = 1, H_B1 = 2, H_A1 = 3 };  // a K-tile's half-tiles in stream (= consumption) order
  // this wave's 2 pieces of half-tile `kind` of K-tile kt (rotation applied here) of tile p into the buffer at byte `bufoff`
  auto stage_half = [&](const TileP &p, int kind, int kt, uint32_t bufoff) {
    if (!kDma) return;
    kt += __builtin_amdgcn_readfirstlane(p.rot);
    if (kt >= nk) kt -= nk;
    const int so = kt * (GEMM_BK * 2);
    if (kind == H_A0 || kind == H_A1) {
      const int ah = kind == H_A1 ? 1 : 0;
      const uint32_t la = lds0 + bufoff + a_dst + ah * 8192;
      gemm_dma16(la, p.a_off[2 * ah], a_rs, so);
      gemm_dma16(la + 1024, p.a_off[2 * ah + 1], a_rs, so);
    } else {
      const int bh = kind == H_B1 ? 1 : 0;
      const uint32_t lw = lds0 + bufoff + w_dst + bh * 4096;
      const int sob = so + bh * (32 * 2) * LDW;
      gemm_dma16(lw, p.w_off[0], w_rs, sob);
      gemm_dma16(lw + 1024, p.w_off[1], w_rs, sob);
    }
  };
  // the tile's 256 bias values (1 KiB) by LDS-DMA as well: no ordinary load sits between the stream's counted waits.  Every wave
  // issues the same piece (same bytes, same place), so each wave's own vmcnt covers the copy it reads and the counts stay uniform.
  auto stage_bias = [&](const TileP &p, int bsel) { gemm_dma16(lds0 + GEMM_LDS_BIAS + bsel * 1024, (uint32_t)(lane * 16), b_rs, __builtin_amdgcn_readfirstlane(p.n0 * 4)); };

  // The stream continues across tiles when the tile has >= 2 K-tiles and the epilogue leaves the registers for the next
  // tile's offsets (EPI 3, the LayerNorm epilogue, does not: every tile then starts from an empty pipeline).
  const bool can_stream = EPI != 3 && nk >= 2;
  const bool dyn = sched != nullptr && !GATHER && can_stream && nk >= 5;  // (the ticket needs two K-tiles of loads behind it: see above)
  int *const mbox = reinterpret_cast<int *>(smem + GEMM_LDS_MBOX);
  TileP cur;
  bool have = false;
  uint32_t par = 0;  // byte offset of the buffer of the current K-tile (0 / GEMM_BUFBYTES), toggles per K-tile ACROSS tiles
  int bsel = 0;      // bias slice of the current tile
  int ti = slot;
  if (dyn) {  // first ticket: nothing is in flight yet, an ordinary atomic and a barrier
    if (tid == 0) mbox[0] = atomicAdd(sched + xcd, 1);
    __syncthreads();
    ti = __builtin_amdgcn_readfirstlane(mbox[0]);
  }
  for (; ti < chunk_len;) {
    const int step = dyn ? ti / nslots : (ti - slot) / nslots;  // tiles an XCD runs together share the K rotation
    if (!have) {
      // empty pipeline: bias + half-tiles 0..5 of the stream (K-tile 0 complete, A0 / B0 of K-tile 1)
      tile_params(ti, step, cur);
      stage_bias(cur, bsel);
      stage_half(cur, H_A0, 0, par);
      stage_half(cur, H_B0, 0, par);
      stage_half(cur, H_B1, 0, par);
      stage_half(cur, H_A1, 0, par);
      if (nk >= 2) {
        stage_half(cur, H_A0, 1, par ^ GEMM_BUFBYTES);
        stage_half(cur, H_B0, 1, par ^ GEMM_BUFBYTES);
        GEMM_WAIT_VM(6);  // A0, B0, B1 of K-tile 0 have landed
      } else {
        GEMM_WAIT_VM(2);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (EPI 3: the LayerNorm parameters written above)
      __builtin_amdgcn_s_barrier();
    }
    const int m0 = __builtin_amdgcn_readfirstlane(cur.m0), n0 = __builtin_amdgcn_readfirstlane(cur.n0);
    cur.rot = __builtin_amdgcn_readfirstlane(cur.rot);
    bool more = can_stream && ti + nslots < chunk_len;  // (dynamic scheduling: decided two K-tiles before the end, from the ticket)
    int ti_next = ti + nslots;
    TileP nxt;

    // the accumulators start at the bias (EPI 3 adds it in its LayerNorm epilogue): the bias slice landed with an earlier wait of
    // the stream (it is the OLDEST load of a fresh pipeline; in a continuing stream it was issued 6 phases before this point)
    const float *bias_lds = reinterpret_cast<const float *>(smem + GEMM_LDS_BIAS + bsel * 1024);
    f32x16 acc[2][4];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 bv = EPI == 3 || GEMM_EABL == 3 ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4 *>(bias_lds + wn * 64 + nb * 32 + 8 * g + 4 * hi);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
          acc[nb][mb][4 * g + 0] = bv.x;
          acc[nb][mb][4 * g + 1] = bv.y;
          acc[nb][mb][4 * g + 2] = bv.z;
          acc[nb][mb][4 * g + 3] = bv.w;
        }
      }

    bf16x8 af[2][4], wf0[4], wf1[4];  // A half (2 row blocks x 4 k-substeps), B0, B1
    auto read_a = [&](const char *lb, int ah) {
      if (!kFrag) return;
#pragma unroll
      for (int mbl = 0; mbl < 2; ++mbl)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) af[mbl][ks] = *reinterpret_cast<const bf16x8 *>(lb + a_base + ah * 8192 + mbl * 4096 + fr_off[ks]);
    };
    auto read_b = [&](const char *lb, int bh, bf16x8(&wf)[4]) {
      if (!kFrag) return;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) wf[ks] = *reinterpret_cast<const bf16x8 *>(lb + w_base + bh * 4096 + fr_off[ks]);
    };
    // one quadrant: 2 row blocks (ah) x 1 column block (bh) x 4 k-substeps, the two accumulators alternating
    auto mfma_q = [&](int ah, int bh, const bf16x8(&wf)[4]) {
      if (!kMfma) {
        asm volatile("" ::"v"(wf[0]), "v"(wf[1]), "v"(wf[2]), "v"(wf[3]), "v"(af[0][0]), "v"(af[0][1]), "v"(af[0][2]), "v"(af[0][3]), "v"(af[1][0]),
                     "v"(af[1][1]), "v"(af[1][2]), "v"(af[1][3]));
        return;
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int mbl = 0; mbl < 2; ++mbl)
          acc[bh][2 * ah + mbl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], af[mbl][ks], acc[bh][2 * ah + mbl], 0, 0, 0);
    };
    if (wm == 1) __builtin_amdgcn_s_barrier();  // group 1 runs one barrier behind group 0
    // ---- K loop: two phases per K-tile, 16 MFMAs each.  X(t) = (A0 x B0, A0 x B1) reads A0, B0, B1 of K-tile t and issues A1 of
    // K-tile t + 1 (the first X of a tile also B1 of K-tile 1, which the epilogue before it kept out of its staging slot);
    // Y(t) = (A1 x B1, A1 x B0) reads A1 and issues A0, B0, B1 of K-tile t + 2.  Each phase is
    //     [reads | DMA pieces | vmcnt(8) | lgkmcnt(0)]  s_barrier  [16 MFMAs]  s_barrier
    // A slot is refilled ONE phase after its last read, which is why the reads are retired BEFORE the phase's first barrier;
    // a half-tile is read one phase after the wait that covers it.  Every wait leaves 4 half-tiles (8 loads) in flight; near the
    // end of a tile the stream either continues with the NEXT tile's half-tiles (`more`) or ends, the waits counting down 2 / 0.
    // One loop body (uniform scalar branches around DMA issue and waits only), so the 32 MFMAs accumulate in place.
    auto phase = [&](auto compute) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (GEMM_PRIO) __builtin_amdgcn_s_setprio(1);
      compute();
      if (GEMM_PRIO) __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    };
    for (int t = 0; t < nk; ++t) {
      const bool last = t + 1 == nk, last2 = t + 2 >= nk;  // K-tile t + 1 / t + 2 belongs to the next tile (or to nobody)
      if (dyn) {
        if (t == 2 && wave == 0) {  // the ticket drawn in Y(0) has landed (14 younger loads, the waits of Y(0), X(1), Y(1) behind it)
          int tk;
          asm volatile("v_readfirstlane_b32 %0, v255" : "=s"(tk)::"memory");
          if (lane == 0) mbox[0] = tk;
        }
        if (t + 2 == nk) {  // (posted >= 2 barriers ago)
          ti_next = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile int *>(mbox));
          more = ti_next < chunk_len;
        }
      }
      if (more && t + 2 == nk) tile_params(ti_next, dyn ? ti_next / nslots : step + 1, nxt);
      const char *lb = smem + par;
      const uint32_t bnext = par ^ GEMM_BUFBYTES;  // buffer of K-tile t + 1; K-tile t + 2 goes where K-tile t is
      // X
      read_b(lb, 0, wf0);
      read_b(lb, 1, wf1);
      read_a(lb, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (t == 0 && !last) stage_half(cur, H_B1, 1, bnext);
      if (!last)
        stage_half(cur, H_A1, t + 1, bnext);
      else if (more)
        stage_half(nxt, H_A1, 0, bnext);
      if (last && !more)
        GEMM_WAIT_VM(0);  // the stream ends: A1 of this K-tile is its last half-tile
      else
        GEMM_WAIT_VM(8);  // A1 of this K-tile has landed
      phase([&] {
        mfma_q(0, 0, wf0);
        mfma_q(0, 1, wf1);
      });
      // Y
      read_a(lb, 1);
      __builtin_amdgcn_sched_barrier(0);
      if (dyn && t == 0 && wave == 0) {  // next tile's ticket: lane 0 only, result into the reserved v255, no wait
        unsigned long long keep;
        const uint32_t zero = 0, one = 1;
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\ts_nop 1\n\tglobal_atomic_add v255, %2, %3, %1 sc0\n\ts_mov_b64 exec, %0\n\ts_nop 1"
                     : "=&s"(keep)
                     : "s"(sched + xcd), "v"(zero), "v"(one)
                     : "memory", "v255");
      }
      if (!last2) {
        stage_half(cur, H_A0, t + 2, par);
        stage_half(cur, H_B0, t + 2, par);
        stage_half(cur, H_B1, t + 2, par);
        GEMM_WAIT_VM(8);  // A0, B0, B1 of K-tile t + 1 have landed
      } else if (more) {
        if (!last) {
          stage_bias(nxt, bsel ^ 1);
          stage_half(nxt, H_A0, 0, par);
          stage_half(nxt, H_B0, 0, par);
          stage_half(nxt, H_B1, 0, par);
          GEMM_WAIT_VM(8);
        } else {
          stage_half(nxt, H_A0, 1, par);
          stage_half(nxt, H_B0, 1, par);
          GEMM_WAIT_VM(6);  // A0, B0, B1 of the next tile's K-tile 0 (B1 of its K-tile 1 follows after the epilogue)
        }
      } else if (!last) {
        GEMM_WAIT_VM(2);  // only A1 of the last K-tile is still in flight
      }
      phase([&] {
        mfma_q(1, 1, wf1);
        mfma_q(1, 0, wf0);
      });
      par ^= GEMM_BUFBYTES;
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();  // both groups have executed the same number of barriers again
    par ^= GEMM_BUFBYTES;                       // back to the LAST K-tile's buffer: its A1 / B1 slots stage C (restored below)
    if (EPI == 3) {
      // v = acc + bias + residual; row statistics across the 4 column waves through LDS; normalise in place
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        const int m = min(m0 + wm * 128 + mb * 32 + l31, M - 1);
        const u16 *rp = resid + (size_t)m * GEMM_BN + wn * 64 + 4 * hi;
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int nl = nb * 32 + 8 * g + 4 * hi;
            const float4 bv = *reinterpret_cast<const float4 *>(bias_lds + wn * 64 + nl);
            const uint2 r = *reinterpret_cast<const uint2 *>(rp + nb * 32 + 8 * g);
            const float v0 = acc[nb][mb][4 * g + 0] + bv.x + __uint_as_float(r.x << 16);
            const float v1 = acc[nb][mb][4 * g + 1] + bv.y + __uint_as_float(r.x & 0xffff0000u);
            const float v2 = acc[nb][mb][4 * g + 2] + bv.z + __uint_as_float(r.y << 16);
            const float v3 = acc[nb][mb][4 * g + 3] + bv.w + __uint_as_float(r.y & 0xffff0000u);
            acc[nb][mb][4 * g + 0] = v0;
            acc[nb][mb][4 * g + 1] = v1;
            acc[nb][mb][4 * g + 2] = v2;
            acc[nb][mb][4 * g + 3] = v3;
            a1 += (v0 + v1) + (v2 + v3);
            a2 += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
          }
        a1 += __shfl_xor(a1, 32);  // lanes l31 and l31 + 32 hold complementary columns of the same row
        a2 += __shfl_xor(a2, 32);
        if (hi == 0) ln_part[((wm * 4 + mb) * 32 + l31) * 4 + wn] = make_float2(a1, a2);
      }
      __syncthreads();
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        const float2 *pp = ln_part + ((wm * 4 + mb) * 32 + l31) * 4;
        const float t1 = (pp[0].x + pp[1].x) + (pp[2].x + pp[3].x), t2 = (pp[0].y + pp[1].y) + (pp[2].y + pp[3].y);
        const float mean = t1 * (1.f / GEMM_BN);
        const float rstd = rsqrtf(fmaxf(t2 * (1.f / GEMM_BN) - mean * mean, 0.f) + ln_eps);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int nl = wn * 64 + nb * 32 + 8 * g + 4 * hi;
            const float4 gw = *reinterpret_cast<const float4 *>(lnw_lds + nl), gb = *reinterpret_cast<const float4 *>(lnb_lds + nl);
            acc[nb][mb][4 * g + 0] = (acc[nb][mb][4 * g + 0] - mean) * rstd * gw.x + gb.x;
            acc[nb][mb][4 * g + 1] = (acc[nb][mb][4 * g + 1] - mean) * rstd * gw.y + gb.y;
            acc[nb][mb][4 * g + 2] = (acc[nb][mb][4 * g + 2] - mean) * rstd * gw.z + gb.z;
            acc[nb][mb][4 * g + 3] = (acc[nb][mb][4 * g + 3] - mean) * rstd * gw.w + gb.w;
          }
      }
    }
    // ---- epilogue: acc[nb][mb][4g + e] = C[m = wm*128 + mb*32 + l31][n = wn*64 + nb*32 + 8g + 4hi + e]
    //      four passes of 32 rows per wave through a 4-KiB slot of the last K-tile's buffer (16-byte slots XOR-swizzled by
    //      row): waves 0-3 use the A1 half-tile slots, waves 4-7 the B1 slots -- the two the stream refills after the epilogue
    char *cw = smem + par + (wave < 4 ? 8192 + (wave & 1) * 4096 + (wave >> 1) * 16384 : GEMM_OPBYTES + 4096 + (wave - 4) * 8192);
    if (GEMM_EABL == 2) {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) asm volatile("" ::"v"(acc[nb][mb]));
    }
    // stores go through a buffer descriptor: rows past M (ragged last tile) fall outside it and are dropped -- no branches
    const uint32_t c_v0 = GATHER ? (uint32_t)((((size_t)m0 + wm * 128 + (lane >> 3)) * GEMM_BN + wn * 64 + (lane & 7) * 8) * 2)
                                 : (uint32_t)((((size_t)m0 + wm * 128 + (lane >> 3)) * LDC + n0 + wn * 64 + (lane & 7) * 8) * 2);
    const uint32_t c_rowb = (uint32_t)((GATHER ? GEMM_BN : LDC) * 2);
#pragma unroll
    for (int mb = 0; mb < (GEMM_EABL == 2 ? 0 : 4); ++mb) {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nl = nb * 32 + 8 * g + 4 * hi;  // local column of the 4 values
          float v0 = acc[nb][mb][4 * g + 0], v1 = acc[nb][mb][4 * g + 1], v2 = acc[nb][mb][4 * g + 2], v3 = acc[nb][mb][4 * g + 3];
          if (EPI == 1 && GEMM_EABL != 3) {
            v0 = gelu_bf16_class(v0);
            v1 = gelu_bf16_class(v1);
            v2 = gelu_bf16_class(v2);
            v3 = gelu_bf16_class(v3);
          }
          if (EPI == 2) {
            v0 = fmaxf(v0, 0.f);
            v1 = fmaxf(v1, 0.f);
            v2 = fmaxf(v2, 0.f);
            v3 = fmaxf(v3, 0.f);
          }
          const int slot16 = (nl >> 3) ^ (l31 & 7);
          *reinterpret_cast<uint2 *>(cw + l31 * 128 + slot16 * 16 + (nl & 4) * 2) = make_uint2(cvt_pk_bf16_f32(v0, v1), cvt_pk_bf16_f32(v2, v3));
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      u32x4 cv[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + (lane >> 3), q = lane & 7;
        cv[it] = *reinterpret_cast<const u32x4 *>(cw + row * 128 + ((q ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const uint32_t off = c_v0 + (uint32_t)(mb * 32 + it * 8) * c_rowb;
        if (GEMM_EABL == 1)
          asm volatile("" ::"v"(cv[it]));
        else if (nt_store)
          __builtin_amdgcn_raw_buffer_store_b128(cv[it], c_rs, off, 0, 2);  // aux 2 = nt
        else
          __builtin_amdgcn_raw_buffer_store_b128(cv[it], c_rs, off, 0, 0);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    par ^= GEMM_BUFBYTES;
    // every wave has read its staged outputs: the next tile's first phases may refill the slots
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    have = more;
    if (more) {
      cur = nxt;
      bsel ^= 1;
    }
    ti = ti_next;
  }  // tile loop
  if (dyn && tid == 0) {  // the last workgroup to finish zeroes the slot (tickets drawn past the end included) for its next launch
    if (atomicAdd(sched + 8, 1) == (int)gridDim.x - 1) {
#pragma unroll
      for (int i = 0; i < 9; ++i) __hip_atomic_store(sched + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace unopose

using namespace unopose;

// Outputs larger than the chip's L2 (8 x 4 MiB) are written with non-temporal stores: they cannot stay cached until the
// next launch reads them, and a round of tiles would otherwise push the operand panels out of L2.
static inline int use_nt_store(long M, int N) { return (size_t)M * N * 2 > (32u << 20) ? 1 : 0; }

// Shape policy: when the 256 x 256 tiles cannot give every CU one (`UNOPOSE_GEMM_SMALL_TILES` overrides the limit for A/Bs:
// scripts/gemm_policy_ab.sh), the GEMM runs on gemm_small.hip's 128 x 128 (64 x 256 with the LayerNorm epilogue) tiles.
// (Measured and not kept: giving the 256-tile kernel only whole rounds of tiles and the last row panels to the small kernel --
// fc2 at M = 87 936 is 1032 tiles on 256 CUs -- gains 0 - 3 % per shape, 0.07 ms per forward: the few tiles of a last round
// already run faster than those of a full one.)
static int small_tiles_limit() {
  static const int v = [] {
    const char *e = getenv("UNOPOSE_GEMM_SMALL_TILES");
    return e && *e ? atoi(e) : -1;
  }();
  return v >= 0 ? v : gemm_cu_count();
}

// Ticket slots of the dynamic tile scheduling: a ring of 1024 slots of 16 ints per device (zeroed once; every launch's last workgroup
// re-zeroes its slot).  Consecutive launches take consecutive slots, so launches of different streams that run at the same time never
// share one (a slot comes round again after 1024 launches: 20 forwards later).  `UNOPOSE_GEMM_DYN=0`: static tile lists (A/B).
static int *sched_slot() {
  static const bool on = [] {
    const char *e = getenv("UNOPOSE_GEMM_DYN");
    return !(e && *e == '0');
  }();
  if (!on) return nullptr;
  static std::mutex mu;
  static int *ring[64] = {nullptr};
  static unsigned next_slot[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  if (!ring[dev]) {
    int *p = nullptr;
    if (hipMalloc(&p, 1024 * 16 * sizeof(int)) != hipSuccess || hipMemset(p, 0, 1024 * 16 * sizeof(int)) != hipSuccess) return nullptr;
    ring[dev] = p;
  }
  return ring[dev] + (size_t)(next_slot[dev]++ & 1023u) * 16;
}

static int linear_bf16_dispatch(const void *A, int lda, const void *W, int ldw, const float *bias, void *C, int ldc, long M, int N, int K,
                                int epilogue, hipStream_t s, const char *what) {
  const int tiles_n = N / GEMM_BN;
  const int tiles = cdiv(M, GEMM_BM) * tiles_n;
  if (tiles < small_tiles_limit()) return gemm_small_linear(A, W, bias, C, M, N, K, lda, ldw, ldc, epilogue, s);
  const int n_cu = gemm_cu_count();
  const int grid = tiles >= n_cu ? n_cu : ((tiles + 7) & ~7);
  const int nt = use_nt_store(M, N);
#define UNOPOSE_LD_LAUNCH(E)                                                                                                                \
  hipLaunchKernelGGL(gemm_bf16_kernel<E>, dim3(grid), dim3(512), 0, s, (const u16 *)A, (const u16 *)W, bias, (u16 *)C, (int)M, N, K, tiles_n, \
                     tiles, nt, (const int *)nullptr, (const int *)nullptr, (const u16 *)nullptr, (const float *)nullptr,                     \
                     (const float *)nullptr, 0.f, lda, ldw, ldc, sched)
  int *const sched = tiles > grid ? sched_slot() : nullptr;  // (one tile per workgroup: nothing to schedule)
  if (epilogue == 1)
    UNOPOSE_LD_LAUNCH(1);
  else if (epilogue == 2)
    UNOPOSE_LD_LAUNCH(2);
  else
    UNOPOSE_LD_LAUNCH(0);
#undef UNOPOSE_LD_LAUNCH
  return check_launch(what);
}

extern "C" {

int unopose_gemm_bf16_tile(void) { return GEMM_BM; }

int unopose_linear_bf16(const void *A, const void *W, const float *bias, void *C, long M, int N, int K, int epilogue,
                        unopose_stream_t stream) {
  UNOPOSE_REQUIRE(A && W && bias && C, "linear_bf16: null pointer");
  UNOPOSE_REQUIRE(M >= 1 && M < (1L << 31) && N >= GEMM_BN && N % GEMM_BN == 0 && K >= GEMM_BK && K % GEMM_BK == 0,
                  "linear_bf16: needs N %% 256 == 0 and K %% 64 == 0 (got M=%ld N=%d K=%d)", M, N, K);
  UNOPOSE_REQUIRE((size_t)M * K * 2 < (1UL << 32) && (size_t)N * K * 2 < (1UL << 32) && (size_t)M * N * 2 < (1UL << 32), "linear_bf16: operand larger than 4 GiB");
  UNOPOSE_REQUIRE(epilogue >= 0 && epilogue <= 2, "linear_bf16: epilogue must be 0 (bias), 1 (bias + GELU) or 2 (bias + ReLU)");
  return linear_bf16_dispatch(A, K, W, K, bias, C, N, M, N, K, epilogue, (hipStream_t)stream, "linear_bf16");
}

int unopose_linear_bf16_ld(const void *A, int lda, const void *W, int ldw, const float *bias, void *C, int ldc, long M, int N, int K,
                           int epilogue, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(A && W && bias && C, "linear_bf16_ld: null pointer");
  UNOPOSE_REQUIRE(M >= 1 && M < (1L << 31) && N >= GEMM_BN && N % GEMM_BN == 0 && K >= GEMM_BK && K % GEMM_BK == 0,
                  "linear_bf16_ld: needs N %% 256 == 0 and K %% 64 == 0 (got M=%ld N=%d K=%d)", M, N, K);
  UNOPOSE_REQUIRE(lda >= K && ldw >= K && ldc >= N && lda % 8 == 0 && ldw % 8 == 0 && ldc % 8 == 0,
                  "linear_bf16_ld: row strides must cover the rows and be multiples of 8 elements (lda=%d ldw=%d ldc=%d)", lda, ldw, ldc);
  UNOPOSE_REQUIRE((size_t)M * lda * 2 < (1UL << 32) && (size_t)N * ldw * 2 < (1UL << 32) && (size_t)M * ldc * 2 < (1UL << 32), "linear_bf16_ld: operand larger than 4 GiB");
  UNOPOSE_REQUIRE(epilogue >= 0 && epilogue <= 2, "linear_bf16_ld: epilogue must be 0 (bias), 1 (bias + GELU) or 2 (bias + ReLU)");
  return linear_bf16_dispatch(A, lda, W, ldw, bias, C, ldc, M, N, K, epilogue, (hipStream_t)stream, "linear_bf16_ld");
}

int unopose_linear_add_layernorm_bf16(const void *A, const void *W, const float *bias, const void *resid, const float *ln_w,
                                      const float *ln_b, float eps, void *C, long M, int K, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(A && W && bias && resid && ln_w && ln_b && C, "linear_add_layernorm_bf16: null pointer");
  UNOPOSE_REQUIRE(M >= 1 && M < (1L << 31) && K >= GEMM_BK && K % GEMM_BK == 0, "linear_add_layernorm_bf16: needs K %% 64 == 0 (got M=%ld K=%d)", M, K);
  UNOPOSE_REQUIRE((size_t)M * K * 2 < (1UL << 32), "linear_add_layernorm_bf16: operand larger than 4 GiB");
  const int tiles = cdiv(M, GEMM_BM);
  if (tiles < small_tiles_limit()) return gemm_small_linear_ln(A, W, bias, resid, ln_w, ln_b, eps, C, M, K, (hipStream_t)stream);
  const int n_cu = gemm_cu_count();
  const int grid = tiles >= n_cu ? n_cu : ((tiles + 7) & ~7);
  hipLaunchKernelGGL((gemm_bf16_kernel<3, false>), dim3(grid), dim3(512), 0, (hipStream_t)stream, (const u16 *)A, (const u16 *)W, bias,
                     (u16 *)C, (int)M, GEMM_BN, K, 1, tiles, use_nt_store(M, GEMM_BN), (const int *)nullptr, (const int *)nullptr,
                     (const u16 *)resid, ln_w, ln_b, eps);
  return check_launch("linear_add_layernorm_bf16");
}

int unopose_linear_bf16_gather(const void *A, long M, int K, const void *W, int N, const float *bias, const int *row_list,
                               const int *tile_info, int max_tiles, void *C, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(A && W && bias && C && row_list && tile_info, "linear_bf16_gather: null pointer");
  UNOPOSE_REQUIRE(M >= 1 && M < (1L << 31) && N >= GEMM_BN && N % GEMM_BN == 0 && N / GEMM_BN <= 64 && K >= GEMM_BK && K % GEMM_BK == 0,
                  "linear_bf16_gather: needs N %% 256 == 0 and K %% 64 == 0 (got M=%ld N=%d K=%d)", M, N, K);
  UNOPOSE_REQUIRE((size_t)M * K * 2 < (1UL << 32) && (size_t)N * K * 2 < (1UL << 32), "linear_bf16_gather: operand larger than 4 GiB");
  UNOPOSE_REQUIRE(max_tiles >= 0, "linear_bf16_gather: bad tile capacity");
  if (max_tiles == 0) return UNOPOSE_OK;
  const int n_cu = gemm_cu_count();
  const int grid = max_tiles >= n_cu ? n_cu : ((max_tiles + 7) & ~7);
  hipLaunchKernelGGL((gemm_bf16_kernel<0, true>), dim3(grid), dim3(512), 0, (hipStream_t)stream, (const u16 *)A, (const u16 *)W, bias,
                     (u16 *)C, (int)M, N, K, 1, 0, 0, row_list, tile_info);
  return check_launch("linear_bf16_gather");
}

}  // extern "C"
