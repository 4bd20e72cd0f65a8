// bf16 "linear" GEMM with fused epilogues for gfx950 (C ABI part 2):
//     C[M][N] = act( A[M][K] . W[N][K]^T + bias[N] ),   A / W / C bf16 row-major, bias fp32, fp32 accumulation.
// This is nn.Linear as timm's ViT blocks call it (qkv, proj, fc1 + exact-erf GELU, fc2; the up-projection of
// oneref_feature_extraction.py:221) -- 41 % of the forward at 518x518 crops when it runs on the library.
//
// Structure (one 512-thread workgroup per 256 x 256 output tile, K step 64, one workgroup per CU):
//   * both operands are K-contiguous, so A and W tiles are the same kind of image: [256 rows][64 k] bf16.
//     Each 1-KiB piece (8 rows x 128 B, whole cache lines) is moved HBM/L2 -> LDS by ONE LDS-DMA wave instruction
//     (global_load_lds_dwordx4: destination = wave-uniform base + lane * 16).  The bank swizzle therefore lives on
//     the per-lane SOURCE address: LDS slot (row, p) holds the row's 16-byte chunk  c = p ^ ((row >> 1) & 7)  and the
//     fragment reads apply the same XOR -- every ds_read_b128 lane group then covers all 16 slots of the 256-byte
//     bank row (conflict-free), and every DMA instruction still fetches full 128-byte lines;
//   * two LDS buffers (2 x 64 KiB): the 8 DMA pieces of K-tile t+1 are spread over the four MFMA groups of tile t,
//     fragment reads run one k-substep ahead of the MFMAs (two register sets), one vmcnt(0) + barrier per K-tile;
//   * every tile starts its K walk at a tile-dependent K-tile (the sum is order-independent): concurrently running
//     tiles then touch different 128-byte columns of their panels at any instant;
//   * 8 waves as 2 (M) x 4 (N), 128 x 64 outputs per wave, v_mfma_f32_32x32x16_bf16 with the operands SWAPPED
//     (rows of the MFMA result = output columns n): a lane then owns 4 consecutive n of one output row, which
//     packs to 8-byte LDS writes in the epilogue;
//   * epilogue: + bias, optional exact GELU (erf by Abramowitz-Stegun 7.1.26, |err| < 1.5e-7, on the fp32
//     accumulators -- no bf16 round trip between the Linear and the activation), bf16, staged through LDS
//     (XOR-swizzled, the K-loop buffers are free by then) and written as whole 128-byte row segments.
#include "common.h"

namespace unopose {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

#ifndef GEMM_ABL
#define GEMM_ABL 0  // scripts/ubench/gemm_abl.py builds variants: 1 = no LDS-DMA in the K loop, 2 = no MFMAs, 3 = no fragment reads
#endif
#ifndef GEMM_ROT
#define GEMM_ROT 1  // K-tile rotation (measured +7 % at K = 768, neutral at K = 3072; scripts/ubench/gemm_abl.py)
#endif
#ifndef GEMM_GM
#define GEMM_GM 4  // row panels per patch of co-resident tiles
#ifndef GEMM_SKEW
#define GEMM_SKEW 4  // tiles sharing a panel start 0..SKEW-1 K-tiles apart
#endif
#endif
#ifndef GEMM_EABL
#define GEMM_EABL 0  // epilogue ablations: 1 = no global stores, 2 = no epilogue at all (accumulators kept live)
#endif
#ifndef GEMM_CNT
#define GEMM_CNT 0  // 1 = non-temporal C stores
#endif
#ifndef GEMM_PHASE
#define GEMM_PHASE 1  // phase groups per XCD (workgroups of group g start g * GEMM_PHASE_TICKS x 10 ns late)
#endif
#ifndef GEMM_PHASE_TICKS
#define GEMM_PHASE_TICKS 1400
#endif
#ifndef GEMM_SAME
#define GEMM_SAME 0  // probe: every tile streams the operands of tile (0, 0) -- an all-hit L2 stream under the full K loop
#endif
#ifndef GEMM_STAMP
#define GEMM_STAMP 0
#endif
#ifndef GEMM_STAMP_BLOCK
#define GEMM_STAMP_BLOCK 16
#endif
#if GEMM_STAMP
__device__ unsigned long long g_stamps[8 * 64 * 4];
extern "C" int unopose_gemm_read_stamps(unsigned long long *host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), sizeof(g_stamps)); }
#endif
#ifndef GEMM_PACE
#define GEMM_PACE 0
#endif
#ifndef GEMM_YPOS
#define GEMM_YPOS 0  // waves 4-7 issue their DMA burst after MFMA group 1 / 2 instead of group 0 (complementary phases on a SIMD)
#endif
#ifndef GEMM_SPEC
#define GEMM_SPEC 0  // 1: waves 4-7 (the later-dispatched wave of every SIMD) issue ALL the LDS-DMA, waves 0-3 none
#endif
#ifndef GEMM_ROTATE
#define GEMM_ROTATE 0
#endif
#ifndef GEMM_PRIO
#define GEMM_PRIO 0  // 1: static s_setprio 1 for waves 4-7 (the later-dispatched half)
#endif
#ifndef GEMM_BURST
#define GEMM_BURST 0  // 1: all 8 DMA pieces of the next K-tile right after the barrier; 2: in the first two MFMA groups
#endif
#define GEMM_BM 256
#define GEMM_BN 256
#define GEMM_BK 64
#define GEMM_OPBYTES (256 * 64 * 2)       // one operand tile: 32 KiB
#define GEMM_BUFBYTES (2 * GEMM_OPBYTES)  // A + W: 64 KiB

__device__ __forceinline__ float gelu_erf(float x) {
  // 0.5 x (1 + erf(x / sqrt 2)); erf(z) = 1 - (a1 t + ... + a5 t^5) exp(-z^2), t = 1 / (1 + p z), z >= 0
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  p *= t;
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);
  const float erfz = fmaf(-p, e, 1.0f);  // erf(|x| / sqrt 2)
  const float hx = 0.5f * x;
  return fmaf(fabsf(hx), erfz, hx);  // 0.5 x + 0.5 |x| erf(|x|/sqrt2) = 0.5 x (1 + sign(x) erf(.))
}

// GATHER (grouped, row-gathered form; unopose_linear_bf16_gather): output row r of tile t is A row row_list[256 t + r]
// times the 256-row weight block of the group tile t belongs to (tile_info[1 + g] = first tile of group g, g = 0..N/256;
// tile_info[0] = number of tiles, read on the device: the host never learns it); C is (tiles * 256, 256).
// EPI 3 (N == 256 only: a row is one tile wide): C = LayerNorm(A W^T + bias + resid) * ln_w + ln_b, the post-LN glue of the
// matcher's transformer layers (transformer.py:151-193) -- the residual add and the LayerNorm run on the fp32 accumulators.
template <int EPI, bool GATHER = false>  // EPI 0: bias; 1: bias + exact GELU; 2: bias + ReLU; 3: bias + residual + LayerNorm
__global__ __launch_bounds__(512, 1) void gemm_bf16_kernel(const u16 *__restrict__ A, const u16 *__restrict__ W,
                                                           const float *__restrict__ bias, u16 *__restrict__ C, int M,
                                                           int N, int K, int tiles_n, int tiles_arg, int cgn,
                                                           const int *__restrict__ row_list = nullptr,
                                                           const int *__restrict__ tile_info = nullptr,
                                                           const u16 *__restrict__ resid = nullptr, const float *__restrict__ ln_w = nullptr,
                                                           const float *__restrict__ ln_b = nullptr, float ln_eps = 0.f) {
  const int tiles = GATHER ? __builtin_amdgcn_readfirstlane(tile_info[0]) : tiles_arg;
  __shared__ __attribute__((aligned(1024))) char smem[2 * GEMM_BUFBYTES];
  __shared__ __attribute__((aligned(16))) float bias_lds[GEMM_BN];  // this tile's bias slice (LDS reads: no vmcnt traffic in the epilogue)
  __shared__ __attribute__((aligned(16))) float lnw_lds[EPI == 3 ? GEMM_BN : 4], lnb_lds[EPI == 3 ? GEMM_BN : 4];
  __shared__ float2 ln_part[EPI == 3 ? 2 * 4 * 32 * 4 : 1];  // [wm][mb][row][wn]: (sum, sum of squares) of 64 columns
  if (EPI == 3 && threadIdx.x < GEMM_BN) {  // visible after the first barrier of the tile loop
    lnw_lds[threadIdx.x] = ln_w[threadIdx.x];
    lnb_lds[threadIdx.x] = ln_b[threadIdx.x];
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  if (GEMM_PRIO == 1 && wave >= 4) __builtin_amdgcn_s_setprio(1);
  const int l31 = lane & 31, hi = lane >> 5;
  // ---- persistent, lock-stepped tile walk.  The grid is ONE workgroup per CU (gridDim.x <= 256, a multiple of 8;
  // 128 KiB of LDS admits one per CU).  Workgroup b sits on XCD b % 8 (observed dispatch rule: a SPEED assumption
  // only) and is slot b / 8 of that XCD; XCD x owns one contiguous range of the tile sequence and its slots take
  // tiles slot, slot + nslots, ... of it.  All workgroups start together and every tile costs the same, so the ~32
  // tiles an XCD has in flight are 32 CONSECUTIVE tiles walking K in lock step: a (GEMM_GM x 32/GEMM_GM) patch of the
  // output that shares GEMM_GM A panels and 32/GEMM_GM W panels K-slice by K-slice in that XCD's L2.  (With one
  // workgroup per tile in dispatch order the resident tiles drift apart in K and the L2 -> LDS stream runs at half
  // the rate: scripts/ubench/gemm_abl.py)
  const int xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3, nslots0 = gridDim.x >> 3;
  const bool phased = GEMM_PHASE > 1 && EPI != 3 && !GATHER && nslots0 % GEMM_PHASE == 0 && tiles >= 2 * (int)gridDim.x;
  const int nph = phased ? GEMM_PHASE : 1;
  const int pg = slot0 % nph, slot = slot0 / nph, nslots = nslots0 / nph;
  if (phased && pg > 0) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)(pg * GEMM_PHASE_TICKS)) __builtin_amdgcn_s_sleep(64);
  }
  const int cq = tiles >> 3, cr = tiles & 7;
  const int tiles_m = tiles / tiles_n, per_group = GEMM_GM * tiles_n;
  // cgn > 0: "resident W" schedule.  XCD x owns a contiguous range of ROW panels and sweeps it once per column group
  // (cgn groups of <= ceil(tiles_n / cgn) column tiles whose W panels together fit the XCD's L2 with room to spare); inside
  // a sweep the tiles in flight form patches of gme row panels x the group's columns, and every row panel of a patch walks
  // K from a different rotation: a W line is then touched gme times per round (reuse distance = W_sub + A_round / gme < L2)
  // and stays resident under LRU, while an A line is used by the row's column tiles within a few K-tiles and dies.
  const int rq = tiles_m >> 3, rr8 = tiles_m & 7;
  const int row_lo = xcd < rr8 ? xcd * (rq + 1) : rr8 * (rq + 1) + (xcd - rr8) * rq, nrows = rq + (xcd < rr8 ? 1 : 0);
  const bool resident = !GATHER && cgn > 0;
  const int chunk_base = xcd < cr ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq;
  const int chunk_len = resident ? nrows * tiles_n : cq + (xcd < cr ? 1 : 0);
  const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc((void *)A, 0, (int)((size_t)M * K * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc((void *)W, 0, (int)((size_t)N * K * 2), 0x00020000);
  const int nk_ = K / GEMM_BK;

  // ---- fragment read addresses: tile row r = base + l31 (base a multiple of 32), chunk c = 2 ks + hi:
  //      byte = (r >> 3) * 1024 + (r & 7) * 128 + ((c ^ ((r >> 1) & 7)) << 4)
  const int fx = (l31 >> 1) & 7;
  uint32_t fr_off[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) fr_off[ks] = (uint32_t)((l31 >> 3) * 1024 + (l31 & 7) * 128 + ((((ks << 1) | hi) ^ fx) << 4));
  const uint32_t a_base = (uint32_t)(wm * 128 * 128);                // A rows wm*128 .. (+ mb * 32 rows = mb * 4096 B)
  const uint32_t w_base = (uint32_t)(GEMM_OPBYTES + wn * 64 * 128);  // W rows wn*64 ..  (+ nb * 4096 B)

  // per-tile DMA parameters: tile origin, K rotation, per-lane source offsets of the wave's 4 + 4 pieces
  struct TileP {
    int m0, n0, rot;
    uint32_t a_off[4], w_off[4];
  };
  auto tile_params = [&](int ti, int step, TileP &p) {
    const int t = chunk_base + ti;
    int tn, tm, rot_res = 0;
    if (GATHER) {
      tm = t;
      tn = 0;
      const int ng = N / GEMM_BN;
      for (int g = 1; g < ng; ++g) tn += t >= tile_info[1 + g] ? 1 : 0;  // the group of tile t (uniform scalar loads)
      tn = __builtin_amdgcn_readfirstlane(tn);
    } else if (resident) {
      const int cgq = tiles_n / cgn, cgr = tiles_n % cgn;
      int rem = ti, c_lo = 0, ncol = 1;
      for (int g = 0; g < cgn; ++g) {  // the column group of sequence position ti (cgn <= 4)
        ncol = cgq + (g < cgr ? 1 : 0);
        if (rem < nrows * ncol || g == cgn - 1) break;
        rem -= nrows * ncol;
        c_lo += ncol;
      }
      const int gme = max(1, (nslots0 + ncol / 2) / ncol);  // rows per patch: patch ~ the tiles the XCD has in flight
      const int pgr = gme * ncol, mg = rem / pgr, rr = rem - mg * pgr;
      const int gm = min(gme, nrows - mg * gme);
      const int tcol = rr / gm, trow = rr - tcol * gm;
      tn = c_lo + tcol;
      tm = row_lo + mg * gme + trow;
      rot_res = (xcd * 5 + (trow * nk_) / gm + tcol % (GEMM_SKEW > 1 ? GEMM_SKEW : 1)) % nk_;
    } else {
      // tile order: groups of GEMM_GM row panels, column tiles fastest across the group
      const int mg = t / per_group, rr = t - mg * per_group;
      const int gm = min(GEMM_GM, tiles_m - mg * GEMM_GM);
      tn = rr / gm;
      tm = mg * GEMM_GM + (rr - tn * gm);
    }
    p.m0 = __builtin_amdgcn_readfirstlane(tm * GEMM_BM);
    p.n0 = __builtin_amdgcn_readfirstlane(tn * GEMM_BN);
    // LDS-DMA (buffer_load_dwordx4 ... lds): piece j = wave * 4 + i covers tile rows 8j .. 8j+7; per-lane byte offset in the
    // VGPR, K-tile offset in an SGPR; rows past M (ragged last tile) fall outside the descriptor -> zeros
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = ((GEMM_SPEC && !GATHER) ? (wave & 3) : wave) * 32 + i * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((row >> 1) & 7);
      int arow = ((GEMM_ABL == 7 || GEMM_SAME) ? 0 : p.m0) + row;  // ABL 7: every tile streams tile (0, 0)
      if (GATHER) arow = max(row_list[p.m0 + row], 0);  // padding rows of a group (-1) compute on row 0; nobody reads them
      p.a_off[i] = (uint32_t)(((size_t)arow * K + c * 8) * 2);
      p.w_off[i] = (uint32_t)(((size_t)(((GEMM_ABL == 7 || GEMM_SAME) ? 0 : p.n0) + row) * K + c * 8) * 2);
    }
    // K-tile rotation, uniform over the tiles an XCD runs together (they must stay on the same K-slice to share it) and
    // different between XCDs / steps: the chip as a whole touches different 128-byte columns at any instant.
    // GEMM_SKEW: tiles sharing a panel start 0..SKEW-1 K-tiles apart, so a K-slice one of them has fetched is RESIDENT in L2
    // when the others ask for it (requests for a line still in flight do not merge into one fetch)
    const int skew = ((tm & 3) + tn) % (GEMM_SKEW > 1 ? GEMM_SKEW : 1);
    p.rot = __builtin_amdgcn_readfirstlane(resident ? rot_res : GEMM_ROT ? (xcd * 5 + step * 3 + skew) % nk_ : 0);
  };
  // The DMA is issued from inline asm: the compiler does not see an LDS write and so keeps its own s_waitcnt vmcnt out of
  // the LDS reads (it would otherwise drain the queue before every fragment read and every epilogue access); the waits
  // on DMA data are the explicit vmcnt + barrier pairs below.
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
#ifndef GEMM_NT
#define GEMM_NT 0  // experiment (scripts/ubench/gemm_abl.py): 1 = activation tiles loaded non-temporal, 2 = weight tiles, 3 = both
#endif
#define GEMM_DMA_ASM(MOD) "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen" MOD " lds\n\ts_mov_b32 m0, %0"
  auto dma16 = [&](uint32_t lds_byte, uint32_t vo, __amdgpu_buffer_rsrc_t rs, int so, bool is_a) {
    unsigned keep;
    if ((GEMM_NT & 1) && is_a)
      asm volatile(GEMM_DMA_ASM(" nt") : "=&s"(keep) : "s"(lds_byte), "v"(vo), "s"(rs), "s"(so) : "memory");
    else if ((GEMM_NT & 2) && !is_a)
      asm volatile(GEMM_DMA_ASM(" nt") : "=&s"(keep) : "s"(lds_byte), "v"(vo), "s"(rs), "s"(so) : "memory");
    else
      asm volatile(GEMM_DMA_ASM("") : "=&s"(keep) : "s"(lds_byte), "v"(vo), "s"(rs), "s"(so) : "memory");
  };
  auto stage_p = [&](const TileP &p, int buf, int kt, int i) {  // pieces i of A and W of K-tile kt (i = 0..3)
    kt += p.rot;
    if (kt >= nk_) kt -= nk_;
    if (GEMM_SPEC && !GATHER) {
      // waves 4-7 stage the rows of waves w - 4 and w (128 rows apart: the second half through the scalar offset)
      if (wave >= 4) {
        const uint32_t la = lds0 + (uint32_t)(buf * GEMM_BUFBYTES + (wave & 3) * 4096 + i * 1024);
        const int hs = 128 * K * 2;
        dma16(la, p.a_off[i], a_rs, kt * (GEMM_BK * 2), true);
        dma16(la + GEMM_OPBYTES, p.w_off[i], w_rs, kt * (GEMM_BK * 2), false);
        dma16(la + 4 * 4096, p.a_off[i], a_rs, kt * (GEMM_BK * 2) + hs, true);
        dma16(la + 4 * 4096 + GEMM_OPBYTES, p.w_off[i], w_rs, kt * (GEMM_BK * 2) + hs, false);
      }
      return;
    }
    const uint32_t la = lds0 + (uint32_t)(buf * GEMM_BUFBYTES + wave * 4096 + i * 1024);
    dma16(la, p.a_off[i], a_rs, kt * (GEMM_BK * 2), true);
    dma16(la + GEMM_OPBYTES, p.w_off[i], w_rs, kt * (GEMM_BK * 2), false);
  };
  // four pieces (consecutive KiB of one operand's image) from ONE M0 setting: the instruction's 12-bit offset moves the LDS
  // destination AND the source address, so the per-lane source offsets are taken relative to it (vo[i] - 1024 i >= 0: row >= 8 i)
  auto dma4 = [&](uint32_t lds_byte, const uint32_t (&vo)[4], __amdgpu_buffer_rsrc_t rs, int so) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %2, %6, %7 offen lds\n\t"
                 "buffer_load_dwordx4 %3, %6, %7 offen offset:1024 lds\n\t"
                 "buffer_load_dwordx4 %4, %6, %7 offen offset:2048 lds\n\t"
                 "buffer_load_dwordx4 %5, %6, %7 offen offset:3072 lds\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_byte), "v"(vo[0]), "v"(vo[1] - 1024u), "v"(vo[2] - 2048u), "v"(vo[3] - 3072u), "s"(rs), "s"(so) : "memory");
  };
  auto stage_all = [&](const TileP &p, int buf, int kt) {
    kt += p.rot;
    if (kt >= nk_) kt -= nk_;
    const uint32_t la = lds0 + (uint32_t)(buf * GEMM_BUFBYTES + wave * 4096);
    dma4(la, p.a_off, a_rs, kt * (GEMM_BK * 2));
    dma4(la + GEMM_OPBYTES, p.w_off, w_rs, kt * (GEMM_BK * 2));
  };
  auto stage_piece = [&](const TileP &p, int buf, int kt, int pc) {  // piece pc of K-tile kt: 0..3 = A pieces, 4..7 = W pieces
    kt += p.rot;
    if (kt >= nk_) kt -= nk_;
    const uint32_t la = lds0 + (uint32_t)(buf * GEMM_BUFBYTES + wave * 4096 + (pc & 3) * 1024);
    if (pc < 4) dma16(la, p.a_off[pc & 3], a_rs, kt * (GEMM_BK * 2), true);
    else dma16(la + GEMM_OPBYTES, p.w_off[pc & 3], w_rs, kt * (GEMM_BK * 2), false);
  };
  // Cross-tile prefetch: the first K-tile of the NEXT tile is put in flight (into buffer 0) right after the last K-tile
  // of this one, so its DMA latency runs under the epilogue (bias / GELU / stores), which stages C through buffer 1 only.
  // Needs the last K-tile in buffer 1, i.e. an even number of K-tiles (768 / 64, 3072 / 64).
  const bool can_prefetch = (nk_ & 1) == 0;
  TileP cur;
  float4 cur_bv;  // bias[n0 + 4 lane ..] of the tile (every wave loads it: no branch, no early wait; wave 0 publishes it)
  bool have = false;
  for (int ti = pg * nslots + slot, step = 0; ti < chunk_len; ti += nph * nslots, ++step) {
  if (!have) {
    tile_params(ti, step, cur);
    cur_bv = *reinterpret_cast<const float4 *>(bias + cur.n0 + lane * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) stage_p(cur, 0, 0, i);
  }
  const int m0 = __builtin_amdgcn_readfirstlane(cur.m0), n0 = __builtin_amdgcn_readfirstlane(cur.n0);
  cur.rot = __builtin_amdgcn_readfirstlane(cur.rot);
  auto stage1 = [&](int buf, int kt, int i) { stage_p(cur, buf, kt, i); };

  f32x16 acc[2][4];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nb][mb][r] = 0.f;

  const int nk = K / GEMM_BK;
  auto read_frags = [&](const char *lb, int ks, bf16x8 (&wf)[2], bf16x8 (&af)[4]) {
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) wf[nb] = *reinterpret_cast<const bf16x8 *>(lb + w_base + nb * 4096 + fr_off[ks]);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) af[mb] = *reinterpret_cast<const bf16x8 *>(lb + a_base + mb * 4096 + fr_off[ks]);
  };
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // K-tile 0 (staged above or prefetched under the previous epilogue)
  if (wave == 0) *reinterpret_cast<float4 *>(bias_lds + lane * 4) = cur_bv;  // read after the K loop's barriers
  __syncthreads();
  // Software pipeline inside a K-tile: the fragment reads of k-substep ks+1 are issued BEFORE the 8 MFMAs of substep ks
  // (two register sets), and the 8 LDS-DMA pieces of the next K-tile are spread over the four MFMA groups (2 per
  // group) instead of being issued in one ~500-cycle burst right after the barrier.
  bf16x8 wf0[2], af0[4], wf1[2], af1[4];
  read_frags(smem, 0, wf0, af0);
#define GEMM_STEP(MORE, KS, WC, AC, WN, AN)                                                             \
    if ((KS) < 3 && GEMM_ABL != 3 && GEMM_ABL < 5) read_frags(lb, (KS) + 1, WN, AN);                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    if (GEMM_ABL == 2 || GEMM_ABL >= 5) { asm volatile("" ::"v"(WC[0]), "v"(WC[1]), "v"(AC[0]), "v"(AC[1]), "v"(AC[2]), "v"(AC[3])); } \
    if (GEMM_ABL != 2 && GEMM_ABL < 5) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WC[0], AC[0], acc[0][0], 0, 0, 0);            \
    if (GEMM_ABL != 2 && GEMM_ABL < 5) acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WC[0], AC[1], acc[0][1], 0, 0, 0);            \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    if (MORE && GEMM_ABL != 1 && GEMM_BURST == 0) stage1(buf ^ 1, kt + 1, (KS));                                          \
    if (MORE && GEMM_BURST == 1 && (KS) == 0) { stage1(buf ^ 1, kt + 1, 0); stage1(buf ^ 1, kt + 1, 1); stage1(buf ^ 1, kt + 1, 2); stage1(buf ^ 1, kt + 1, 3); } \
    if (MORE && GEMM_BURST == 2 && (KS) < 2) { stage1(buf ^ 1, kt + 1, 2 * (KS)); stage1(buf ^ 1, kt + 1, 2 * (KS) + 1); } \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    if (GEMM_ABL != 2 && GEMM_ABL < 5) acc[0][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WC[0], AC[2], acc[0][2], 0, 0, 0);            \
    if (GEMM_ABL != 2 && GEMM_ABL < 5) acc[0][3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WC[0], AC[3], acc[0][3], 0, 0, 0);            \
    if (GEMM_ABL != 2 && GEMM_ABL < 5) acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WC[1], AC[0], acc[1][0], 0, 0, 0);            \
    if (GEMM_ABL != 2 && GEMM_ABL < 5) acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WC[1], AC[1], acc[1][1], 0, 0, 0);            \
    if (GEMM_ABL != 2 && GEMM_ABL < 5) acc[1][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WC[1], AC[2], acc[1][2], 0, 0, 0);            \
    if (GEMM_ABL != 2 && GEMM_ABL < 5) acc[1][3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WC[1], AC[3], acc[1][3], 0, 0, 0);            \
    __builtin_amdgcn_sched_barrier(0);
#if GEMM_ROTATE
  // Rotated software pipeline: the last MFMA group (k-substep 3) of K-tile t is issued AFTER the barrier that ends the tile,
  // behind the fragment reads of K-tile t+1's substep 0 -- the matrix pipe restarts right at the barrier release while those
  // reads are in flight.  (Every wave's reads of tile t are complete before the barrier: lgkmcnt(0) in front of it.)
  // GEMM_PACE = E > 0: the wave's 8 LDS-DMA pieces of the next K-tile are issued ONE at a time, after every E-th MFMA, from
  // a wave-dependent offset: the CU's texture-address path moves 64 B / clk (16 clk per 1-KiB piece, 64 pieces per K-tile),
  // so a burst of pieces blocks the issuing waves -- in order, in front of their MFMAs -- for hundreds of cycles.
#define GEMM_MF1(M, WC, AC, NB, MB)                                                                   \
    acc[NB][MB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WC[NB], AC[MB], acc[NB][MB], 0, 0, 0);      \
    if (GEMM_PACE > 0 && more_k && (M) % GEMM_PACE == 0 && (M) / GEMM_PACE < 8) {                      \
      __builtin_amdgcn_sched_barrier(0);                                                             \
      if (dma_off == 0) stage_piece(cur, buf ^ 1, kt + 1, (M) / GEMM_PACE);                           \
      __builtin_amdgcn_sched_barrier(0);                                                             \
    }                                                                                                \
    if (GEMM_PACE > 1 && more_k && (M) % GEMM_PACE == 1 && (M) / GEMM_PACE < 8) {                      \
      __builtin_amdgcn_sched_barrier(0);                                                             \
      if (dma_off == 1) stage_piece(cur, buf ^ 1, kt + 1, (M) / GEMM_PACE);                           \
      __builtin_amdgcn_sched_barrier(0);                                                             \
    }                                                                                                \
    if (GEMM_PACE > 2 && more_k && (M) % GEMM_PACE == 2 && (M) / GEMM_PACE < 8) {                      \
      __builtin_amdgcn_sched_barrier(0);                                                             \
      if (dma_off == 2) stage_piece(cur, buf ^ 1, kt + 1, (M) / GEMM_PACE);                           \
      __builtin_amdgcn_sched_barrier(0);                                                             \
    }                                                                                                \
    if (GEMM_PACE > 3 && more_k && (M) % GEMM_PACE == 3 && (M) / GEMM_PACE < 8) {                      \
      __builtin_amdgcn_sched_barrier(0);                                                             \
      if (dma_off == 3) stage_piece(cur, buf ^ 1, kt + 1, (M) / GEMM_PACE);                           \
      __builtin_amdgcn_sched_barrier(0);                                                             \
    }
#define GEMM_MF8(M0, WC, AC)                                                                         \
    GEMM_MF1((M0) + 0, WC, AC, 0, 0) GEMM_MF1((M0) + 1, WC, AC, 0, 1) GEMM_MF1((M0) + 2, WC, AC, 0, 2) GEMM_MF1((M0) + 3, WC, AC, 0, 3) \
    GEMM_MF1((M0) + 4, WC, AC, 1, 0) GEMM_MF1((M0) + 5, WC, AC, 1, 1) GEMM_MF1((M0) + 6, WC, AC, 1, 2) GEMM_MF1((M0) + 7, WC, AC, 1, 3)
  const int dma_off = GEMM_PACE > 0 ? __builtin_amdgcn_readfirstlane(((wave & 3) + (wave >> 2) * (GEMM_PACE / 2)) % GEMM_PACE) : 0;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    const char *lb = smem + buf * GEMM_BUFBYTES;
    const bool more_k = kt + 1 < nk;
    // (entry: fragments of substep 0 of this tile are being read into set 0; set 1 holds substep 3 of the previous tile)
    if (kt > 0) { GEMM_MF8(0, wf1, af1) }
    else if (GEMM_PACE > 0 && more_k) {  // first K-tile of the tile: no pending group -- its share of the pieces goes out at once
#pragma unroll
      for (int m = 0; m < 8; ++m)
        if (m % GEMM_PACE == dma_off && m / GEMM_PACE < 8) stage_piece(cur, buf ^ 1, kt + 1, m / GEMM_PACE);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (more_k && GEMM_PACE == 0 && (GEMM_YPOS == 0 || wave < 4)) {
      if (GEMM_BURST == 3) stage_all(cur, buf ^ 1, kt + 1);
      else if (GEMM_BURST == 1) { stage1(buf ^ 1, kt + 1, 0); stage1(buf ^ 1, kt + 1, 1); stage1(buf ^ 1, kt + 1, 2); stage1(buf ^ 1, kt + 1, 3); }
      else { stage1(buf ^ 1, kt + 1, 0); stage1(buf ^ 1, kt + 1, 1); }
    }
    __builtin_amdgcn_sched_barrier(0);
    read_frags(lb, 1, wf1, af1);
    __builtin_amdgcn_sched_barrier(0);
    GEMM_MF8(8, wf0, af0)
    __builtin_amdgcn_sched_barrier(0);
    if (more_k && GEMM_YPOS == 1 && wave >= 4) { stage1(buf ^ 1, kt + 1, 0); stage1(buf ^ 1, kt + 1, 1); stage1(buf ^ 1, kt + 1, 2); stage1(buf ^ 1, kt + 1, 3); }
    __builtin_amdgcn_sched_barrier(0);
    if (more_k && GEMM_PACE == 0 && GEMM_BURST != 1 && GEMM_BURST != 3) { stage1(buf ^ 1, kt + 1, 2); stage1(buf ^ 1, kt + 1, 3); }
    __builtin_amdgcn_sched_barrier(0);
    read_frags(lb, 2, wf0, af0);
    __builtin_amdgcn_sched_barrier(0);
    GEMM_MF8(16, wf1, af1)
    __builtin_amdgcn_sched_barrier(0);
    if (more_k && GEMM_YPOS == 2 && wave >= 4) { stage1(buf ^ 1, kt + 1, 0); stage1(buf ^ 1, kt + 1, 1); stage1(buf ^ 1, kt + 1, 2); stage1(buf ^ 1, kt + 1, 3); }
    __builtin_amdgcn_sched_barrier(0);
    read_frags(lb, 3, wf1, af1);
    __builtin_amdgcn_sched_barrier(0);
    GEMM_MF8(24, wf0, af0)
    __builtin_amdgcn_sched_barrier(0);
#if GEMM_STAMP
    const bool st_on = blockIdx.x == GEMM_STAMP_BLOCK && step == 2;
    unsigned long long tB = 0, tC = 0, tA = 0;
    if (st_on) { tB = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (st_on) { tC = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    __builtin_amdgcn_s_barrier();
    if (st_on) { tA = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) { unsigned long long *sp = g_stamps + ((size_t)wave * 64 + kt) * 4; sp[0] = tB; sp[1] = tC; sp[2] = tA; } }
#else
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#endif
    __builtin_amdgcn_sched_barrier(0);
    if (more_k) read_frags(smem + (buf ^ 1) * GEMM_BUFBYTES, 0, wf0, af0);
  }
  // (the last tile's substep 3 is still pending: issued below, after the next tile's prefetch has been put in flight)
#else
  for (int kt = 0; kt < nk - 1; ++kt) {
    const int buf = kt & 1;
    const char *lb = smem + buf * GEMM_BUFBYTES;
    GEMM_STEP(true, 0, wf0, af0, wf1, af1)
    GEMM_STEP(true, 1, wf1, af1, wf0, af0)
    GEMM_STEP(true, 2, wf0, af0, wf1, af1)
    GEMM_STEP(true, 3, wf1, af1, wf0, af0)
#if GEMM_ABL == 5  // probe: is the LDS-DMA stream latency- or bandwidth-bound?  One whole K-tile stays in flight across the barrier
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#endif
    if (GEMM_ABL != 3 && GEMM_ABL < 5) read_frags(smem + (buf ^ 1) * GEMM_BUFBYTES, 0, wf0, af0);
  }
  {
    const int kt = nk - 1, buf = kt & 1;
    const char *lb = smem + buf * GEMM_BUFBYTES;
    GEMM_STEP(false, 0, wf0, af0, wf1, af1)
    GEMM_STEP(false, 1, wf1, af1, wf0, af0)
    GEMM_STEP(false, 2, wf0, af0, wf1, af1)
    GEMM_STEP(false, 3, wf1, af1, wf0, af0)
    __syncthreads();  // every wave is done reading the K-loop buffers: the epilogue reuses them
  }
#endif
#undef GEMM_STEP

  // ---- next tile's first K-tile in flight under this tile's epilogue
  const bool more = EPI != 3 && can_prefetch && ti + nph * nslots < chunk_len;  // (EPI 3: the LayerNorm epilogue needs the registers)
  TileP nxt;
  float4 nxt_bv;
  if (more) {
    tile_params(ti + nph * nslots, step + 1, nxt);
    nxt_bv = *reinterpret_cast<const float4 *>(bias + nxt.n0 + lane * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) stage_p(nxt, 0, 0, i);
  }
#if GEMM_ROTATE
  {
    const bool more_k = false;
    const int kt = 0, buf = 0;
    (void)kt; (void)buf;
    GEMM_MF8(0, wf1, af1)
  }
#undef GEMM_MF8
#undef GEMM_MF1
#endif
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
  //      two passes of 64 rows per wave through buffer 1 (8 KiB per wave, 16-byte slots XOR-swizzled by row)
  char *cw = smem + GEMM_BUFBYTES + wave * (64 * 128);
  char *Cb = reinterpret_cast<char *>(C);
  if (GEMM_EABL == 2) {
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) asm volatile("" ::"v"(acc[nb][mb]));
  }
#pragma unroll
  for (int ps = 0; ps < (GEMM_EABL == 2 ? 0 : 2); ++ps) {
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int nl = nb * 32 + 8 * g + 4 * hi;  // local column of the 4 values
        const float4 bv = EPI == 3 ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4 *>(bias_lds + wn * 64 + nl);
#pragma unroll
        for (int mh = 0; mh < 2; ++mh) {
          const int mb = ps * 2 + mh;
          float v0 = acc[nb][mb][4 * g + 0] + bv.x, v1 = acc[nb][mb][4 * g + 1] + bv.y;
          float v2 = acc[nb][mb][4 * g + 2] + bv.z, v3 = acc[nb][mb][4 * g + 3] + bv.w;
          if (EPI == 1) {
            v0 = gelu_erf(v0);
            v1 = gelu_erf(v1);
            v2 = gelu_erf(v2);
            v3 = gelu_erf(v3);
          }
          if (EPI == 2) {
            v0 = fmaxf(v0, 0.f);
            v1 = fmaxf(v1, 0.f);
            v2 = fmaxf(v2, 0.f);
            v3 = fmaxf(v3, 0.f);
          }
          const int row = mh * 32 + l31;
          const int slot = (nl >> 3) ^ (row & 7);
          *reinterpret_cast<uint2 *>(cw + row * 128 + slot * 16 + (nl & 4) * 2) = make_uint2(cvt_pk_bf16_f32(v0, v1), cvt_pk_bf16_f32(v2, v3));
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = it * 8 + (lane >> 3), q = lane & 7;
      const uint4 v = *reinterpret_cast<const uint4 *>(cw + row * 128 + ((q ^ (row & 7)) << 4));
      const int m = m0 + wm * 128 + ps * 64 + row;
      if (GEMM_EABL == 1) { asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); }
      else if (GATHER) *reinterpret_cast<uint4 *>(Cb + ((size_t)m * GEMM_BN + wn * 64 + q * 8) * 2) = v;
      else if (m < M) {
        if (GEMM_CNT) {
          typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
          const u32x4 vv = {v.x, v.y, v.z, v.w};
          __builtin_nontemporal_store(vv, reinterpret_cast<u32x4 *>(Cb + ((size_t)m * N + n0 + wn * 64 + q * 8) * 2));
        }
        else *reinterpret_cast<uint4 *>(Cb + ((size_t)m * N + n0 + wn * 64 + q * 8) * 2) = v;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  __syncthreads();  // every wave has read its staged outputs: the next tile's K loop may overwrite buffer 1
  have = more;
  if (more) {
    cur = nxt;
    cur_bv = nxt_bv;
  }
  }  // tile loop
}

}  // namespace unopose

using namespace unopose;

#ifndef GEMM_WSUB_KB
#define GEMM_WSUB_KB 2400  // W panels of one column group: what may stay resident in a 4 MiB L2 beside the streaming A panels
#endif
#ifndef GEMM_CGN
#define GEMM_CGN -1  // experiments: force the number of column groups (0 = the lock-step schedule)
#endif
// Number of column groups of the resident-W schedule, or 0 for the lock-step schedule: the choice that moves fewer bytes
// over the L2 miss path.  resident: every XCD reads its rows of A once per column group (cgn x A in total) and W once;
// lock-step: A once, W once per round of every XCD.
static int resident_w_groups(int tiles_m, int tiles_n, int K, int grid) {
  if (GEMM_CGN >= 0) return GEMM_CGN;
  if (grid < 64 || tiles_m < 16) return 0;
  const double panel = 256.0 * K * 2, a_bytes = panel * tiles_m, w_bytes = panel * tiles_n;
  const double groups = tiles_m / 8.0 / GEMM_GM, rounds = tiles_m / 8.0 * tiles_n / (grid / 8);
  const double lockstep = a_bytes + w_bytes * 8 * (groups < rounds ? groups : rounds);  // W once per patch of co-resident tiles
  int best = 0;
  double best_cost = lockstep;
  for (int c = 1; c <= 4 && c <= tiles_n; ++c) {
    const int ncol = (tiles_n + c - 1) / c;
    if (ncol * panel > GEMM_WSUB_KB * 1024.0) continue;
    const double cost = c * a_bytes + 8 * w_bytes;
    if (cost < best_cost) best_cost = cost, best = c;
  }
  return best;
}

extern "C" {

int unopose_gemm_bf16_tile(void) { return GEMM_BM; }

int unopose_linear_bf16(const void *A, const void *W, const float *bias, void *C, long M, int N, int K, int epilogue,
                        unopose_stream_t stream) {
  UNOPOSE_REQUIRE(A && W && bias && C, "linear_bf16: null pointer");
  UNOPOSE_REQUIRE(M >= 1 && M < (1L << 31) && N >= GEMM_BN && N % GEMM_BN == 0 && K >= GEMM_BK && K % GEMM_BK == 0,
                  "linear_bf16: needs N %% 256 == 0 and K %% 64 == 0 (got M=%ld N=%d K=%d)", M, N, K);
  UNOPOSE_REQUIRE((size_t)M * K * 2 < (1UL << 32) && (size_t)N * K * 2 < (1UL << 32), "linear_bf16: operand larger than 4 GiB");
  UNOPOSE_REQUIRE(epilogue >= 0 && epilogue <= 2, "linear_bf16: epilogue must be 0 (bias), 1 (bias + GELU) or 2 (bias + ReLU)");
  const int tiles_m = cdiv(M, GEMM_BM), tiles_n = N / GEMM_BN, tiles = tiles_m * tiles_n;
  hipStream_t s = (hipStream_t)stream;
  static int n_cu = 0;  // one persistent workgroup per CU
  if (n_cu == 0) {
    int dev = 0, cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu < 8)
      cu = 256;
    n_cu = cu & ~7;
  }
  const int grid = tiles >= n_cu ? n_cu : ((tiles + 7) & ~7);
  const int cgn = resident_w_groups(tiles_m, tiles_n, K, grid);
  if (epilogue == 1)
    hipLaunchKernelGGL(gemm_bf16_kernel<1>, dim3(grid), dim3(512), 0, s, (const u16 *)A, (const u16 *)W, bias, (u16 *)C, (int)M, N,
                       K, tiles_n, tiles, cgn);
  else if (epilogue == 2)
    hipLaunchKernelGGL(gemm_bf16_kernel<2>, dim3(grid), dim3(512), 0, s, (const u16 *)A, (const u16 *)W, bias, (u16 *)C, (int)M, N,
                       K, tiles_n, tiles, cgn);
  else
    hipLaunchKernelGGL(gemm_bf16_kernel<0>, dim3(grid), dim3(512), 0, s, (const u16 *)A, (const u16 *)W, bias, (u16 *)C, (int)M, N,
                       K, tiles_n, tiles, cgn);
  return check_launch("linear_bf16");
}

int unopose_linear_add_layernorm_bf16(const void *A, const void *W, const float *bias, const void *resid, const float *ln_w,
                                      const float *ln_b, float eps, void *C, long M, int K, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(A && W && bias && resid && ln_w && ln_b && C, "linear_add_layernorm_bf16: null pointer");
  UNOPOSE_REQUIRE(M >= 1 && M < (1L << 31) && K >= GEMM_BK && K % GEMM_BK == 0, "linear_add_layernorm_bf16: needs K %% 64 == 0 (got M=%ld K=%d)", M, K);
  UNOPOSE_REQUIRE((size_t)M * K * 2 < (1UL << 32), "linear_add_layernorm_bf16: operand larger than 4 GiB");
  const int tiles = cdiv(M, GEMM_BM);
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0, cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu < 8)
      cu = 256;
    n_cu = cu & ~7;
  }
  const int grid = tiles >= n_cu ? n_cu : ((tiles + 7) & ~7);
  hipLaunchKernelGGL((gemm_bf16_kernel<3, false>), dim3(grid), dim3(512), 0, (hipStream_t)stream, (const u16 *)A, (const u16 *)W, bias,
                     (u16 *)C, (int)M, GEMM_BN, K, 1, tiles, 0, (const int *)nullptr, (const int *)nullptr, (const u16 *)resid, ln_w, ln_b, eps);
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
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0, cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu < 8)
      cu = 256;
    n_cu = cu & ~7;
  }
  const int grid = max_tiles >= n_cu ? n_cu : ((max_tiles + 7) & ~7);
  hipLaunchKernelGGL((gemm_bf16_kernel<0, true>), dim3(grid), dim3(512), 0, (hipStream_t)stream, (const u16 *)A, (const u16 *)W, bias,
                     (u16 *)C, (int)M, N, K, 1, 0, 0, row_list, tile_info);
  return check_launch("linear_bf16_gather");
}

}  // extern "C"
