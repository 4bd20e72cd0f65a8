// The 256 x 256-tile persistent GEMM kernel shared by gemm.hip (bf16 operands) and gemm_f32.hip (fp32-class: hi / lo-split operands,
// three MFMAs per product).  See gemm.hip for the design notes.
#pragma once
#include "gemm_common.h"

namespace unopose {

#ifndef GEMM_ABL
#define GEMM_ABL 0  // scripts/ubench/gemm_var.py: 1 = no LDS-DMA in the K loop, 2 = no MFMAs, 3 = no fragment reads
#endif
#ifndef GEMM_EABL
#define GEMM_EABL 0  // epilogue ablations: 1 = no global stores, 2 = no epilogue at all (accumulators kept live), 3 = no bias / activation math, 4 = EPI 5 without the loads of the old x
#endif
#ifndef GEMM_SAME
#define GEMM_SAME 0  // probe: every tile streams the operands of tile (0, 0) -- an all-hit L2 stream under the full K loop
#endif
#ifndef GEMM_POLA
#define GEMM_POLA 0  // cache policy of the A / W operand streams (gemm_dma16<POL>)
#endif
#ifndef GEMM_POLW
#define GEMM_POLW 0
#endif
#ifndef GEMM_CW
#define GEMM_CW -1  // column-blocked tile walk: column tiles per block (0: never; -1: chosen from the size of W, see the kernel)
#endif
#ifndef GEMM_PRIO
#define GEMM_PRIO 0  // 1 = s_setprio 1 around the MFMA segment (measured: -1..2 % with 16-MFMA segments; scripts/ubench/gemm_r04_variants.hip)
#endif
#ifndef GEMM_LNF_ABL
#define GEMM_LNF_ABL 0  // EPI 6 / 7 timing ablations (wrong results): 1 = no LayerNorm math in the epilogue, 2 = no c / row-partial DMA, 3 = both
#endif
#ifndef GEMM_XPOL
#define GEMM_XPOL 2  // cache policy of EPI 5's accesses to the fp32 residual stream (aux bits; 2 = nt): a stream of 540 MB per launch that is next read a whole block later -- default policy: proj 209 us at M = 87 936, nt 178, nt + sc0 183, nt + sc1 184, sc1 195-207; fc2 439-442 with every policy (round 6)
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
#define GEMM_LDS_STATS GEMM_LDS_LNPART     // EPI 6 / 7: the tile's 256 x nparts row partials (sum, sum of squares), 2 parities x 8 KiB (nparts <= 4)
#define GEMM_LNF_MAXPARTS 4

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
// F32 = the fp32-class form (gemm_f32.hip): operands in the "split layout" (per row and 32-k block one 128-byte line [hi(32 bf16) |
// lo(32 bf16)], 4 bytes per k), a K-tile is 32 k (the same 128-byte rows, so the SAME stream, slots and phases), three MFMAs per
// fragment pair (hi.hi + hi.lo + lo.hi), epilogues: EPI 0 / 1 / 2 -> C fp32 and / or C2 in the split layout (bias / exact-erf GELU /
// ReLU); EPI 4 -> C = bf16(resid + bf16(x W^T + b)).  Its epilogue stages C through a WHOLE buffer (8 KiB per wave), so the next
// tile's K-tile 1 is put in flight after the epilogue instead of before it.
// ROUND 6 -- the ViT's residual + LayerNorm passes folded into the GEMMs around them (timm Block: x + ls(f(norm(x))),
// oneref_feature_extraction.py:24-42; bf16 form only):
//   EPI 5 (producer: proj / fc2 with LayerScale folded into W and b): x = xres + acc on the FP32 residual stream, in place (`C2v`),
//         C = bf16(x) (the un-normalised rows the next GEMM reads) and per (row, column tile) the partial sums (sum x, sum x^2) of the
//         tile's 256 columns into `stats[row * tiles_n + tn]` -- the four column waves combined through LDS, no atomics;
//   EPI 6 / 7 (consumer: qkv / fc1 + GELU on W' = g (.) W): acc = x_bf16 W'^T starts at zero and the epilogue applies LayerNorm
//         algebraically, out = rstd_r (acc - mean_r c_n) + d_n with c_n = sum_k W'[n][k] (`aux_vec`), d_n = sum_k beta_k W[n][k] + b_n
//         (`bias`), mean_r / rstd_r from the `nparts` partials of row r in `stats` (ln_eps; the LayerNorm width is K).
template <int EPI, bool GATHER = false, bool F32 = false>  // EPI 0: bias; 1: bias + GELU; 2: bias + ReLU; 3: bias + residual + LayerNorm; 5 / 6 / 7: above
__global__ __launch_bounds__(512, 1) __attribute__((amdgpu_num_vgpr(127))) void gemm256_kernel(const void *__restrict__ Av, const void *__restrict__ Wv,
                                                           const float *__restrict__ bias, void *__restrict__ Cv, int M,
                                                           int N, int K, int tiles_n, int tiles_arg, int nt_store,
                                                           const int *__restrict__ row_list = nullptr,
                                                           const int *__restrict__ tile_info = nullptr,
                                                           const u16 *__restrict__ resid = nullptr, const float *__restrict__ ln_w = nullptr,
                                                           const float *__restrict__ ln_b = nullptr, float ln_eps = 0.f, int lda = 0,
                                                           int ldw = 0, int ldc = 0, int *__restrict__ sched = nullptr,
                                                           void *__restrict__ C2v = nullptr, const float *__restrict__ aux_vec = nullptr,
                                                           float2 *__restrict__ stats = nullptr, int nparts = 0, int stagger = 0) {
  static_assert(!(F32 && (EPI == 3 || GATHER)), "the fp32-class form has no LayerNorm epilogue / gathered rows");
  static_assert(!((F32 || GATHER) && EPI >= 5), "the residual / LayerNorm-fold epilogues belong to the dense bf16 form");
  constexpr bool LNF = EPI == 6 || EPI == 7;  // consumer of the folded LayerNorm
  static_assert(F32 || EPI != 4, "EPI 4 (bf16 + residual out of fp32-class operands) is an F32 epilogue");
  constexpr int ESZ = F32 ? 4 : 2;                        // bytes per k of an operand row
  constexpr int KT = F32 ? 32 : 64;                       // k per K-tile (128-byte rows either way)
  const char *A = reinterpret_cast<const char *>(Av), *W = reinterpret_cast<const char *>(Wv);
  // row strides in elements (0 = dense; unopose_linear_bf16_ld)
  const int LDA = lda ? lda : K, LDW = ldw ? ldw : K, LDC = ldc ? ldc : N;
  const int tiles = GATHER ? __builtin_amdgcn_readfirstlane(tile_info[0]) : tiles_arg;
  __shared__ __attribute__((aligned(1024))) char smem[GEMM_LDS_BIAS + 2048 + 16 + (EPI == 3 || EPI == 5 ? 2048 + 8192 : LNF ? 2048 + 2 * 8192 : 0)];
  float *const lnw_lds = reinterpret_cast<float *>(smem + GEMM_LDS_LNW), *const lnb_lds = reinterpret_cast<float *>(smem + GEMM_LDS_LNB);
  float2 *const ln_part = reinterpret_cast<float2 *>(smem + GEMM_LDS_LNPART);  // [wm][mb][row][wn]: (sum, sum of squares) of 64 columns
  if (EPI == 3 && threadIdx.x < GEMM_BN) {  // visible after the first barrier of the tile loop
    lnw_lds[threadIdx.x] = ln_w[threadIdx.x];
    lnb_lds[threadIdx.x] = ln_b[threadIdx.x];
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int l31 = lane & 31, hi = lane >> 5;
  const int l31_ = l31, hi_ = hi, lane_ = lane;  // (names for the fp32-class epilogue's laundered copies)
  // ---- persistent, lock-stepped tile walk.  The grid is ONE workgroup per CU (gridDim.x <= 256, a multiple of 8;
  // 130 KiB of LDS admits one per CU).  Workgroup b sits on XCD b % 8 (observed dispatch rule: a SPEED assumption
  // only) and is slot b / 8 of that XCD; XCD x owns one contiguous range of the tile sequence and its slots take
  // tiles slot, slot + nslots, ... of it.  All workgroups start together and every tile costs the same, so the ~32
  // tiles an XCD has in flight are 32 CONSECUTIVE tiles walking K in lock step.  The order is row-major (GEMM_GM = 1): the tiles
  // that share an A panel are neighbours in the sequence, hence closest in time, and start on the same K-tile (GEMM_SKEW = 1) --
  // round 4's counters: fabric reads of the four ViT linears 2691 -> 1668 MB per layer against (GM 4, SKEW 4), same time.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int tiles_m = tiles / tiles_n, per_group = GEMM_GM * tiles_n;
  // Column-blocked walk: XCD x owns a contiguous range of ROW PANELS and walks it once per block of `cw` column tiles, so the
  // block's W panels stay in that XCD's 4 MiB L2 while its share of A streams through.  Used when W as a whole does not fit
  // (fc1: 4.7 MB; measured fabric reads 646 -> 417 MB per launch) and the row panels split evenly enough over the 8 XCDs.
  int cw = GEMM_CW;
  if (GEMM_CW < 0) {
    cw = (size_t)N * K * ESZ > (size_t)(4u << 20) ? (2400 * 1024) / (GEMM_BN * K * ESZ) : 0;
    if (cw < 3) cw = 0;
  }
  const bool colwalk = cw > 0 && !GATHER && tiles_n > cw && ((tiles_m & 7) == 0 || tiles_m >= 512);
  const int rq = tiles_m >> 3, rr = tiles_m & 7;
  const int rows_x = rq + (xcd < rr ? 1 : 0), row_base = xcd < rr ? xcd * (rq + 1) : rr * (rq + 1) + (xcd - rr) * rq;
  const int cq = tiles >> 3, cr = tiles & 7;
  const int chunk_base = xcd < cr ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq;
  const int chunk_len = colwalk ? rows_x * tiles_n : cq + (xcd < cr ? 1 : 0);
  const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc((void *)A, 0, (int)((size_t)M * LDA * ESZ), 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc((void *)W, 0, (int)((size_t)N * LDW * ESZ), 0x00020000);
  const __amdgpu_buffer_rsrc_t c_rs = __builtin_amdgcn_make_buffer_rsrc(Cv, 0, GATHER ? 0x7fffffff : (int)((size_t)M * LDC * 2), 0x00020000);  // (bf16 form)
  const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc((void *)bias, 0, N * 4, 0x00020000);
  const int nk = K / KT;

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
    } else if (colwalk) {
      const int per_block = rows_x * cw;
      const int cb = ti / per_block, v = ti - cb * per_block;
      const int cwb = min(cw, tiles_n - cb * cw);  // (the last block may be narrower)
      const int g = v / (GEMM_GM * cwb), w = v - g * GEMM_GM * cwb;
      const int gm = min(GEMM_GM, rows_x - g * GEMM_GM);
      const int tl = w / gm;
      tn = cb * cw + tl;
      tm = row_base + g * GEMM_GM + (w - tl * gm);
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
      p.a_off[j] = (uint32_t)((size_t)arow * LDA * ESZ + c * 16);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (wave >> 1) * 64 + (wave & 1) * 16 + i * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((row >> 1) & 7);
      p.w_off[i] = (uint32_t)((size_t)((GEMM_SAME ? 0 : p.n0) + row) * LDW * ESZ + c * 16);
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
    const int so = kt * GEMM_ROWB;
    if (kind == H_A0 || kind == H_A1) {
      const int ah = kind == H_A1 ? 1 : 0;
      const uint32_t la = lds0 + bufoff + a_dst + ah * 8192;
      gemm_dma16<GEMM_POLA>(la, p.a_off[2 * ah], a_rs, so);
      gemm_dma16<GEMM_POLA>(la + 1024, p.a_off[2 * ah + 1], a_rs, so);
    } else {
      const int bh = kind == H_B1 ? 1 : 0;
      const uint32_t lw = lds0 + bufoff + w_dst + bh * 4096;
      const int sob = so + bh * (32 * ESZ) * LDW;
      gemm_dma16<GEMM_POLW>(lw, p.w_off[0], w_rs, sob);
      gemm_dma16<GEMM_POLW>(lw + 1024, p.w_off[1], w_rs, sob);
    }
  };
  // the tile's 256 bias values (1 KiB) by LDS-DMA as well: no ordinary load sits between the stream's counted waits.  Every wave
  // issues the same piece (same bytes, same place), so each wave's own vmcnt covers the copy it reads and the counts stay uniform.
  // (EPI 6 / 7: the tile's 256 column sums c_n ride with the bias slice, into the slot pair EPI 3 keeps its LayerNorm weights in)
  const __amdgpu_buffer_rsrc_t cv_rs = __builtin_amdgcn_make_buffer_rsrc((void *)(LNF ? aux_vec : bias), 0, N * 4, 0x00020000);
  // and so do the row partials of the tile's 256 rows (256 x nparts x 8 bytes, contiguous: 2 nparts pieces), so that the epilogue finds
  // them in LDS instead of paying an L2 round trip per tile (the stats buffer holds whole tiles of rows: no bounds)
  const __amdgpu_buffer_rsrc_t st_rs = __builtin_amdgcn_make_buffer_rsrc((void *)stats, 0, LNF ? (int)((size_t)((M + GEMM_BM - 1) / GEMM_BM) * GEMM_BM * nparts * 8) : 0, 0x00020000);
  auto stage_bias = [&](const TileP &p, int bsel) {
    if (!LNF) {
      gemm_dma16(lds0 + GEMM_LDS_BIAS + bsel * 1024, (uint32_t)(lane * 16), b_rs, __builtin_amdgcn_readfirstlane(p.n0 * 4));
      return;
    }
    // EPI 6 / 7 read all of this in the EPILOGUE only (the accumulators start at zero), many barriers after the issuing wave's counted waits
    // have retired it: the 2 + 2 nparts pieces (d slice, c slice, row partials) are dealt over the waves, ONE per wave and slot (a wave
    // whose piece index is past the end repeats piece 0, so that every wave issues the same number of loads and the counted waits stay
    // uniform) instead of every wave fetching everything (+7 % on the L2 -> LDS stream of a K = 768 tile)
    const int total = (GEMM_LNF_ABL & 2) ? 1 : 2 + 2 * nparts;
#pragma unroll
    for (int rep = 0; rep < (2 + 2 * GEMM_LNF_MAXPARTS + 7) / 8; ++rep) {
      if (rep * 8 >= total) break;  // (uniform)
      int pc = rep * 8 + wave;
      if (pc >= total) pc = 0;
      if (pc == 0)
        gemm_dma16(lds0 + GEMM_LDS_BIAS + bsel * 1024, (uint32_t)(lane * 16), b_rs, __builtin_amdgcn_readfirstlane(p.n0 * 4));
      else if (pc == 1)
        gemm_dma16(lds0 + GEMM_LDS_LNW + bsel * 1024, (uint32_t)(lane * 16), cv_rs, __builtin_amdgcn_readfirstlane(p.n0 * 4));
      else
        gemm_dma16(lds0 + GEMM_LDS_STATS + bsel * 8192 + (pc - 2) * 1024, (uint32_t)(lane * 16), st_rs,
                   __builtin_amdgcn_readfirstlane(p.m0 * nparts * 8 + (pc - 2) * 1024));
    }
  };
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
  if (EPI == 5 && stagger > 0 && ((slot / tiles_n) & 1)) {
    // STAGGER (see gemm.hip): every second GROUP of tiles_n workgroups of an XCD -- the workgroups that draw the tiles of one row panel, i.e.
    // share an A panel through L2, stay together -- starts `stagger` ticks of the 100 MHz clock late (about half a tile), so that the
    // HBM-sized residual epilogues of one half of the chip run beside the K loops of the other half instead of all 256 at once
    // (proj at M = 87 936: 208 -> 190 us; fc2 459 -> 443 us).  The longer, contended epilogues let the sharers of a panel drift apart in
    // K all the same: fc2's fabric reads are 1.5 GB per launch against 0.74 GB with the plain epilogue (profiles/r06_pmc_summary.json).
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while ((int64_t)(__builtin_amdgcn_s_memrealtime() - t0) < (int64_t)stagger) __builtin_amdgcn_s_sleep(8);
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
        const float4 bv = EPI == 3 || LNF || GEMM_EABL == 3 ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4 *>(bias_lds + wn * 64 + nb * 32 + 8 * g + 4 * hi);
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
      if (F32) {
        // fragments 0 / 1 = the hi parts of the two 16-k blocks, 2 / 3 = their lo parts; small terms first
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int mbl = 0; mbl < 2; ++mbl) {
            f32x16 &a = acc[bh][2 * ah + mbl];
            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[2 + b], af[mbl][b], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[b], af[mbl][2 + b], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[b], af[mbl][b], a, 0, 0, 0);
          }
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
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"  // v255 is reserved (amdgpu_num_vgpr): naming it keeps the kernel's register count at 256
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\ts_nop 1\n\tglobal_atomic_add v255, %2, %3, %1 sc0\n\ts_mov_b64 exec, %0\n\ts_nop 1"
                     : "=&s"(keep)
                     : "s"(sched + xcd), "v"(zero), "v"(one)
                     : "memory", "v255");
#pragma clang diagnostic pop
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
          if (LNF) {  // the column sums and the row partials issued with the bias slice stay in flight too: they have until the next phase's wait
            if (nparts <= 3) GEMM_WAIT_VM(9);   // one piece per wave (8 pieces in all)
            else GEMM_WAIT_VM(10);              // two
          } else
            GEMM_WAIT_VM(8);
        } else if (F32) {
          GEMM_WAIT_VM(2);  // (this epilogue stages C through the whole buffer: the next tile's K-tile 1 follows after it)
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
    if constexpr (LNF && !(GEMM_LNF_ABL & 1)) {
      // ---- LayerNorm applied algebraically on the accumulators: out = rstd_r * acc - (rstd_r * mean_r) * c_n + d_n
      int l31 = l31_, hi = hi_;
      asm volatile("" : "+v"(l31), "+v"(hi));
      const float *c_lds = reinterpret_cast<const float *>(smem + GEMM_LDS_LNW + bsel * 1024);
      const float inv_k = 1.f / (float)K;
      float rs[4], nm[4];
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        const float2 *sp = reinterpret_cast<const float2 *>(smem + GEMM_LDS_STATS + bsel * 8192) + (wm * 128 + mb * 32 + l31) * nparts;
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int i = 0; i < GEMM_LNF_MAXPARTS; ++i)
          if (i < nparts) {
            const float2 q = sp[i];
            t1 += q.x;
            t2 += q.y;
          }
        const float mean = t1 * inv_k;
        rs[mb] = rsqrtf(fmaxf(t2 * inv_k - mean * mean, 0.f) + ln_eps);
        nm[mb] = -mean * rs[mb];
      }
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nl = wn * 64 + nb * 32 + 8 * g + 4 * hi;
          const float4 cv = *reinterpret_cast<const float4 *>(c_lds + nl), dv = *reinterpret_cast<const float4 *>(bias_lds + nl);
#pragma unroll
          for (int mb = 0; mb < 4; ++mb) {
            acc[nb][mb][4 * g + 0] = fmaf(rs[mb], acc[nb][mb][4 * g + 0], fmaf(nm[mb], cv.x, dv.x));
            acc[nb][mb][4 * g + 1] = fmaf(rs[mb], acc[nb][mb][4 * g + 1], fmaf(nm[mb], cv.y, dv.y));
            acc[nb][mb][4 * g + 2] = fmaf(rs[mb], acc[nb][mb][4 * g + 2], fmaf(nm[mb], cv.z, dv.z));
            acc[nb][mb][4 * g + 3] = fmaf(rs[mb], acc[nb][mb][4 * g + 3], fmaf(nm[mb], cv.w, dv.w));
          }
        }
    }
    if constexpr (EPI == 5) {
      // ---- residual epilogue: x = xres + acc on the fp32 residual stream in place, C = bf16(x), row partial sums of x.
      // The fp32 stream must move in WHOLE cache lines (a first form that read and wrote it in the accumulator layout -- 32 bytes per row
      // and instruction -- cost as much as the separate LayerNorm pass it replaces: 128 us per launch).  So the accumulators go through
      // the wave's 4-KiB staging slot 16 rows x 64 columns at a time ([row][256 B], 16-byte slots XOR-swizzled by row) and come back
      // row-major: 16 lanes x 16 bytes = one row's 256 contiguous bytes per DPP row -- the load of the old x, the store of the new x
      // (2 full lines) and the bf16 store (1 full line) are all contiguous, and the row sums are DPP-row reductions.  Three rounds of
      // loads (12 KiB per wave) are kept in flight ahead of the staging.
      int l31 = l31_, hi = hi_, lane = lane_;
      asm volatile("" : "+v"(l31), "+v"(hi), "+v"(lane));
      char *cw = smem + par + (wave < 4 ? 8192 + (wave & 1) * 4096 + (wave >> 1) * 16384 : GEMM_OPBYTES + 4096 + (wave - 4) * 8192);
      const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(C2v, 0, (int)((size_t)M * N * 4), 0x00020000);
      const int r4 = lane >> 4, q = lane & 15;
      const uint32_t x_v0 = (uint32_t)((((size_t)m0 + wm * 128 + r4) * N + n0 + wn * 64 + q * 4) * 4);
      const uint32_t c_v0 = (uint32_t)((((size_t)m0 + wm * 128 + r4) * LDC + n0 + wn * 64 + q * 4) * 2);
      const uint32_t x_rowb = (uint32_t)(N * 4), c_rowb = (uint32_t)(LDC * 2);
      constexpr int PF = 3;  // rounds of loads in flight (4 spills 56 VGPRs)
      f32x4 rv[PF][4];
      auto load_round = [&](int r, f32x4(&dst)[4]) {  // round r = rows r * 16 .. + 15 of the wave's 128
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          if (GEMM_EABL >= 4) {  // timing probes (wrong results): 4 = no loads of the old x (what an epilogue of stores alone costs); 5 = nor the fp32 store; 6 = nor the row sums; 7 = neither
            dst[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            continue;
          }
          dst[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rs, x_v0 + (uint32_t)(r * 16 + it * 4) * x_rowb, 0, GEMM_XPOL));
        }
      };
#pragma unroll
      for (int r = 0; r < PF; ++r) load_round(r, rv[r]);
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int mb = r >> 1, h = r & 1;
        if ((l31 >> 4) == h) {
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int s16 = (nb * 8 + 2 * g + hi) ^ (l31 & 15);
              *reinterpret_cast<float4 *>(cw + (l31 & 15) * 256 + s16 * 16) =
                  make_float4(acc[nb][mb][4 * g + 0], acc[nb][mb][4 * g + 1], acc[nb][mb][4 * g + 2], acc[nb][mb][4 * g + 3]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int row = it * 4 + r4;
          const f32x4 a = *reinterpret_cast<const f32x4 *>(cw + row * 256 + ((q ^ row) << 4));
          const f32x4 o = rv[r % PF][it];
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = a[e] + o[e];
          if (GEMM_EABL != 5 && GEMM_EABL != 7)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), x_rs, x_v0 + (uint32_t)(r * 16 + it * 4) * x_rowb, 0, GEMM_XPOL);
          typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
          const u32x2 pk = {cvt_pk_bf16_f32(v[0], v[1]), cvt_pk_bf16_f32(v[2], v[3])};
          if (nt_store)
            __builtin_amdgcn_raw_buffer_store_b64(pk, c_rs, c_v0 + (uint32_t)(r * 16 + it * 4) * c_rowb, 0, 2);
          else
            __builtin_amdgcn_raw_buffer_store_b64(pk, c_rs, c_v0 + (uint32_t)(r * 16 + it * 4) * c_rowb, 0, 0);
          if (GEMM_EABL == 6 || GEMM_EABL == 7) continue;
          float a1 = (v[0] + v[1]) + (v[2] + v[3]), a2 = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
          a1 += dpp_f32<0x111, 0xF>(a1, 0.f);  // row_shr 1, 2, 4, 8: the 16 lanes of a DPP row hold one matrix row; the sum lands in lane 15
          a2 += dpp_f32<0x111, 0xF>(a2, 0.f);
          a1 += dpp_f32<0x112, 0xF>(a1, 0.f);
          a2 += dpp_f32<0x112, 0xF>(a2, 0.f);
          a1 += dpp_f32<0x114, 0xF>(a1, 0.f);
          a2 += dpp_f32<0x114, 0xF>(a2, 0.f);
          a1 += dpp_f32<0x118, 0xF>(a1, 0.f);
          a2 += dpp_f32<0x118, 0xF>(a2, 0.f);
          if (q == 15) ln_part[((wm * 4 + mb) * 32 + h * 16 + row) * 4 + wn] = make_float2(a1, a2);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (r + PF < 8) load_round(r + PF, rv[r % PF]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // (raw: the next tile's LDS-DMA stream stays in flight)
      if (wn == 0 && hi == 0) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
          const float2 *pp = ln_part + ((wm * 4 + mb) * 32 + l31) * 4;
          const float4 p01 = *reinterpret_cast<const float4 *>(pp), p23 = *reinterpret_cast<const float4 *>(pp + 2);
          // (the stats buffer is padded to whole 256-row tiles: no bounds test)
          stats[(size_t)(m0 + wm * 128 + mb * 32 + l31) * tiles_n + (n0 >> 8)] = make_float2((p01.x + p01.z) + (p23.x + p23.z), (p01.y + p01.w) + (p23.y + p23.w));
        }
      }
    } else if constexpr (!F32) {
    // (lane-derived addresses recomputed per tile from laundered values: see the fp32-class epilogue below)
    int l31 = l31_, hi = hi_, lane = lane_;
    asm volatile("" : "+v"(l31), "+v"(hi), "+v"(lane));
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
          if ((EPI == 1 || EPI == 7) && GEMM_EABL != 3) {
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
    } else {
    // ---- fp32-class epilogue: acc[nb][mb][4g + e] = C[m = wm*128 + mb*32 + l31][n = wn*64 + nb*32 + 8g + 4hi + e] (bias already inside);
    //      four passes (one per mb) of 32 rows x 64 columns per wave through 8 KiB of the last K-tile's buffer
    // (lane-derived LDS addresses below are recomputed per tile from laundered values: left loop-invariant, the compiler hoists ~30
    //  of them out of the TILE loop and they spill across the K loop)
    int l31 = l31_, hi = hi_, lane = lane_;
    asm volatile("" : "+v"(l31), "+v"(hi), "+v"(lane));
    char *cw = smem + par + wave * 8192;
    // stores through buffer descriptors with 32-bit offsets: rows past M fall outside and are dropped (no branches, no 64-bit
    // address registers -- which spilled into the K loop)
    const __amdgpu_buffer_rsrc_t cf_rs = __builtin_amdgcn_make_buffer_rsrc(Cv, 0, (int)((size_t)M * N * (EPI == 4 ? 2 : 4)), 0x00020000);
    const __amdgpu_buffer_rsrc_t cs_rs = __builtin_amdgcn_make_buffer_rsrc(C2v, 0, (int)((size_t)M * N * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_rs = __builtin_amdgcn_make_buffer_rsrc((void *)resid, 0, (int)((size_t)M * N * 2), 0x00020000);
    const bool has_cf = Cv != nullptr, has_cs = C2v != nullptr;
    // fp32 row-major and the split layout put (row m, 16-byte piece q of the wave's 64 columns) at the same byte offset
    const uint32_t c32_v0 = (uint32_t)((((size_t)m0 + wm * 128 + (lane >> 4)) * N + n0 + wn * 64 + (lane & 15) * 4) * 4);
    const uint32_t c16_v0 = (uint32_t)((((size_t)m0 + wm * 128 + (lane >> 3)) * N + n0 + wn * 64 + (lane & 7) * 8) * 2);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      // the 4 values a lane holds of (nb, g); recomputed per output image instead of kept (32 registers) across both
      auto vals = [&](int nb, int g, float (&x)[4]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = acc[nb][mb][4 * g + e];
          if (EPI == 1) t = gelu_erf(t);
          if (EPI == 2) t = fmaxf(t, 0.f);
          x[e] = t;
        }
        __builtin_amdgcn_sched_barrier(0);  // one group at a time: interleaving all 32 of a pass spills into the K loop
      };
      if (EPI == 4) {
        // bf16 image: row = 128 B = 8 slots of 16 B (8 columns), slot nb * 4 + g, XOR-swizzled by the row
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float x[4];
            vals(nb, g, x);
            *reinterpret_cast<uint2 *>(cw + l31 * 128 + (((nb * 4 + g) ^ (l31 & 7)) << 4) + hi * 8) =
                make_uint2(cvt_pk_bf16_f32(x[0], x[1]), cvt_pk_bf16_f32(x[2], x[3]));
          }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int row = it * 8 + (lane >> 3), q = lane & 7;
          u32x4 o = *reinterpret_cast<const u32x4 *>(cw + row * 128 + ((q ^ (row & 7)) << 4));
          const uint32_t off = c16_v0 + (uint32_t)(mb * 32 + it * 8) * (uint32_t)(N * 2);
          if (resid) {
            const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs_rs, off, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
              o[e] = cvt_pk_bf16_f32(__uint_as_float(o[e] << 16) + __uint_as_float(r[e] << 16),
                                     __uint_as_float(o[e] & 0xffff0000u) + __uint_as_float(r[e] & 0xffff0000u));
          }
          __builtin_amdgcn_raw_buffer_store_b128(o, cf_rs, off, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        continue;
      }
      if (has_cf) {
        // fp32 image: row = 256 B = 16 slots of 16 B, slot s = nb * 8 + 2 g + hi, XOR-swizzled by the row
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float x[4];
            vals(nb, g, x);
            const int s = (nb * 8 + 2 * g + hi) ^ (l31 & 15);
            *reinterpret_cast<float4 *>(cw + l31 * 256 + s * 16) = make_float4(x[0], x[1], x[2], x[3]);
          }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int row = it * 4 + (lane >> 4), q = lane & 15;
          const u32x4 o = *reinterpret_cast<const u32x4 *>(cw + row * 256 + ((q ^ (row & 15)) << 4));
          const uint32_t off = c32_v0 + (uint32_t)(mb * 32 + it * 4) * (uint32_t)(N * 4);
          if (nt_store)
            __builtin_amdgcn_raw_buffer_store_b128(o, cf_rs, off, 0, 2);  // aux 2 = nt
          else
            __builtin_amdgcn_raw_buffer_store_b128(o, cf_rs, off, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      if (has_cs) {
        // split image of the wave's 64 columns = 2 k-blocks of 128 B: block nb, element j = 8 g + 4 hi + e: hi part at byte
        // nb * 128 + 2 j, lo part 64 bytes further; 16-byte slots XOR-swizzled by the row
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float x[4];
            vals(nb, g, x);
            const uint32_t h0 = cvt_pk_bf16_f32(x[0], x[1]), h1 = cvt_pk_bf16_f32(x[2], x[3]);
            const float r0 = x[0] - __uint_as_float(h0 << 16), r1 = x[1] - __uint_as_float(h0 & 0xffff0000u);
            const float r2 = x[2] - __uint_as_float(h1 << 16), r3 = x[3] - __uint_as_float(h1 & 0xffff0000u);
            const uint32_t l0 = cvt_pk_bf16_f32(r0, r1), l1 = cvt_pk_bf16_f32(r2, r3);
            const int sh = (nb * 8 + g) ^ (l31 & 15), sl = (nb * 8 + 4 + g) ^ (l31 & 15);  // slot of the hi / lo 16-byte chunk (8 elements)
            *reinterpret_cast<uint2 *>(cw + l31 * 256 + sh * 16 + hi * 8) = make_uint2(h0, h1);
            *reinterpret_cast<uint2 *>(cw + l31 * 256 + sl * 16 + hi * 8) = make_uint2(l0, l1);
          }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int row = it * 4 + (lane >> 4), q = lane & 15;
          const u32x4 o = *reinterpret_cast<const u32x4 *>(cw + row * 256 + ((q ^ (row & 15)) << 4));
          // the row's split bytes of columns n0 + wn * 64 .. + 63 are contiguous: (n / 32) * 128 = (n0 + wn * 64) * 4
          __builtin_amdgcn_raw_buffer_store_b128(o, cs_rs, c32_v0 + (uint32_t)(mb * 32 + it * 4) * (uint32_t)(N * 4), 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
    }
    par ^= GEMM_BUFBYTES;
    // every wave has read its staged outputs: the next tile's first phases may refill the slots
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (F32 && more) {  // (the buffer the epilogue staged C through is free again)
      stage_half(nxt, H_A0, 1, par ^ GEMM_BUFBYTES);
      stage_half(nxt, H_B0, 1, par ^ GEMM_BUFBYTES);
    }
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
