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
//   * two LDS buffers (2 x 64 KiB).  All 8 DMA pieces a wave owes the next K-tile go out in one burst right after the
//     barrier (every piece then has the whole K-step to land; measured against spreading them over the MFMA groups, pacing
//     them one per 2 / 3 / 4 MFMAs from wave-dependent offsets, and letting one wave of every SIMD issue all of them:
//     DESIGN.md section 7), fragment reads run one k-substep ahead of the MFMAs (two register sets), one vmcnt(0) + barrier
//     per K-tile;
//   * ROTATED software pipeline: the last MFMA group (k-substep 3) of a K-tile is issued AFTER the barrier that ends the
//     tile, behind the fragment reads of the next tile's substep 0 -- the matrix pipe restarts at the barrier release
//     while those reads are in flight;
//   * every tile starts its K walk at a tile-dependent K-tile (the sum is order-independent): concurrently running
//     tiles then touch different 128-byte columns of their panels at any instant;
//   * 8 waves as 2 (M) x 4 (N), 128 x 64 outputs per wave, v_mfma_f32_32x32x16_bf16 with the operands SWAPPED
//     (rows of the MFMA result = output columns n): a lane then owns 4 consecutive n of one output row, which
//     packs to 8-byte LDS writes in the epilogue;
//   * epilogue on the fp32 accumulators: + bias, optional GELU / ReLU / residual + LayerNorm, bf16, staged through LDS
//     (XOR-swizzled, the K-loop buffers are free by then) and written as whole 128-byte row segments with NON-TEMPORAL
//     stores (a round of tiles writes 4 MiB per XCD -- the size of its L2 -- and nothing re-reads C before the next
//     launch); the next tile's first K-tile streams in meanwhile.
#include "gemm_common.h"

namespace unopose {

#ifndef GEMM_ABL
#define GEMM_ABL 0  // scripts/ubench/gemm_var.py: 1 = no LDS-DMA in the K loop, 2 = no MFMAs, 3 = no fragment reads, 6 = DMA only
#endif
#ifndef GEMM_EABL
#define GEMM_EABL 0  // epilogue ablations: 1 = no global stores, 2 = no epilogue at all (accumulators kept live)
#endif
#ifndef GEMM_SAME
#define GEMM_SAME 0  // probe: every tile streams the operands of tile (0, 0) -- an all-hit L2 stream under the full K loop
#endif
#ifndef GEMM_STAMP
#define GEMM_STAMP 0  // probe: s_memtime stamps around the K-step's waits of one workgroup (scripts/ubench/gv_stamp.py)
#endif
#if GEMM_STAMP
__device__ unsigned long long g_stamps[8 * 64 * 4];
extern "C" int unopose_gemm_read_stamps(unsigned long long *host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), sizeof(g_stamps)); }
#endif
#define GEMM_BK 64
constexpr bool kMfma = GEMM_ABL != 2 && GEMM_ABL < 5, kFrag = GEMM_ABL != 3 && GEMM_ABL < 5, kDma = GEMM_ABL != 1;

// GATHER (grouped, row-gathered form; unopose_linear_bf16_gather): output row r of tile t is A row row_list[256 t + r]
// times the 256-row weight block of the group tile t belongs to (tile_info[1 + g] = first tile of group g, g = 0..N/256;
// tile_info[0] = number of tiles, read on the device: the host never learns it); C is (tiles * 256, 256).
// EPI 3 (N == 256 only: a row is one tile wide): C = LayerNorm(A W^T + bias + resid) * ln_w + ln_b, the post-LN glue of the
// matcher's transformer layers (transformer.py:151-193) -- the residual add and the LayerNorm run on the fp32 accumulators.
#ifndef GEMM_PP
#define GEMM_PP 0  // 1 / 2: ping-pong K loop (see the kernel); 2 = all 8 DMA pieces in phase 0
#endif
#ifndef GEMM_ROTX
#define GEMM_ROTX 5  // K-tile rotation between XCDs (-1: spread evenly, xcd * nk / 8) and between steps
#endif
#ifndef GEMM_ROTS
#define GEMM_ROTS 3
#endif
template <int EPI, bool GATHER = false>  // EPI 0: bias; 1: bias + GELU; 2: bias + ReLU; 3: bias + residual + LayerNorm
__global__ __launch_bounds__(512, 1) void gemm_bf16_kernel(const u16 *__restrict__ A, const u16 *__restrict__ W,
                                                           const float *__restrict__ bias, u16 *__restrict__ C, int M,
                                                           int N, int K, int tiles_n, int tiles_arg, int nt_store,
                                                           const int *__restrict__ row_list = nullptr,
                                                           const int *__restrict__ tile_info = nullptr,
                                                           const u16 *__restrict__ resid = nullptr, const float *__restrict__ ln_w = nullptr,
                                                           const float *__restrict__ ln_b = nullptr, float ln_eps = 0.f, int lda = 0,
                                                           int ldw = 0, int ldc = 0) {
  // row strides in elements (0 = dense; unopose_linear_bf16_ld).  (Padding the 6144-byte rows of the ViT's hidden activation by 64
  // elements was tried against L2 channel camping: no gain on the fc1 -> fc2 pair)
  const int LDA = lda ? lda : K, LDW = ldw ? ldw : K, LDC = ldc ? ldc : N;
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
  const int l31 = lane & 31, hi = lane >> 5;
  // ---- persistent, lock-stepped tile walk.  The grid is ONE workgroup per CU (gridDim.x <= 256, a multiple of 8;
  // 128 KiB of LDS admits one per CU).  Workgroup b sits on XCD b % 8 (observed dispatch rule: a SPEED assumption
  // only) and is slot b / 8 of that XCD; XCD x owns one contiguous range of the tile sequence and its slots take
  // tiles slot, slot + nslots, ... of it.  All workgroups start together and every tile costs the same, so the ~32
  // tiles an XCD has in flight are 32 CONSECUTIVE tiles walking K in lock step: a (GEMM_GM x 32/GEMM_GM) patch of the
  // output that shares GEMM_GM A panels and 32/GEMM_GM W panels K-slice by K-slice in that XCD's L2.  (Measured against a
  // schedule that keeps the W panels of a column group resident in L2 -- fewer L2 misses, same time: the K loop is not
  // bound by the miss path)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int cq = tiles >> 3, cr = tiles & 7;
  const int chunk_base = xcd < cr ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq, chunk_len = cq + (xcd < cr ? 1 : 0);
  const int tiles_m = tiles / tiles_n, per_group = GEMM_GM * tiles_n;
  const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc((void *)A, 0, (int)((size_t)M * LDA * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc((void *)W, 0, (int)((size_t)N * LDW * 2), 0x00020000);
  const int nk = K / GEMM_BK;

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
    // LDS-DMA piece j = wave * 4 + i covers tile rows 8j .. 8j+7; per-lane byte offset in the VGPR, K-tile offset in an
    // SGPR; rows past M (ragged last tile) fall outside the descriptor -> zeros
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wave * 32 + i * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((row >> 1) & 7);
      int arow = (GEMM_SAME ? 0 : p.m0) + row;
      if (GATHER) arow = max(row_list[p.m0 + row], 0);  // padding rows of a group (-1) compute on row 0; nobody reads them
      p.a_off[i] = (uint32_t)(((size_t)arow * LDA + c * 8) * 2);
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
  auto stage_tile = [&](const TileP &p, int buf, int kt) {  // the wave's 4 A + 4 W pieces of K-tile kt
    kt += p.rot;
    if (kt >= nk) kt -= nk;
    const uint32_t la = lds0 + (uint32_t)(buf * GEMM_BUFBYTES + wave * 4096);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      gemm_dma16(la + i * 1024, p.a_off[i], a_rs, kt * (GEMM_BK * 2));
      gemm_dma16(la + i * 1024 + GEMM_OPBYTES, p.w_off[i], w_rs, kt * (GEMM_BK * 2));
    }
  };
  // Cross-tile prefetch: the first K-tile of the NEXT tile is put in flight (into buffer 0) right after the last K-tile
  // of this one, so its DMA latency runs under the epilogue (bias / GELU / stores), which stages C through buffer 1 only.
  // Needs the last K-tile in buffer 1, i.e. an even number of K-tiles (768 / 64, 3072 / 64).
  const bool can_prefetch = (nk & 1) == 0;
  TileP cur;
  float4 cur_bv;  // bias[n0 + 4 lane ..] of the tile (every wave loads it: no branch, no early wait; wave 0 publishes it)
  bool have = false;
  for (int ti = slot, step = 0; ti < chunk_len; ti += nslots, ++step) {
    if (!have) {
      tile_params(ti, step, cur);
      cur_bv = *reinterpret_cast<const float4 *>(bias + cur.n0 + lane * 4);
      stage_tile(cur, 0, 0);
    }
    const int m0 = __builtin_amdgcn_readfirstlane(cur.m0), n0 = __builtin_amdgcn_readfirstlane(cur.n0);
    cur.rot = __builtin_amdgcn_readfirstlane(cur.rot);

    f32x16 acc[2][4];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][mb][r] = 0.f;

    auto read_frags = [&](const char *lb, int ks, bf16x8(&wf)[2], bf16x8(&af)[4]) {
      if (!kFrag) return;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) wf[nb] = *reinterpret_cast<const bf16x8 *>(lb + w_base + nb * 4096 + fr_off[ks]);
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) af[mb] = *reinterpret_cast<const bf16x8 *>(lb + a_base + mb * 4096 + fr_off[ks]);
    };
    auto mfma8 = [&](const bf16x8(&wf)[2], const bf16x8(&af)[4]) {
      if (!kMfma) {
        asm volatile("" ::"v"(wf[0]), "v"(wf[1]), "v"(af[0]), "v"(af[1]), "v"(af[2]), "v"(af[3]));
        return;
      }
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) acc[nb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nb], af[mb], acc[nb][mb], 0, 0, 0);
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // K-tile 0 (staged above or prefetched under the previous epilogue)
    if (wave == 0) *reinterpret_cast<float4 *>(bias_lds + lane * 4) = cur_bv;  // read after the K loop's barriers
    __syncthreads();
    bf16x8 wf0[2], af0[4], wf1[2], af1[4];
#if GEMM_PP
    // ---- ping-pong K loop (GEMM_PP): the two wave groups of the workgroup (wm = 0: waves 0-3, wm = 1: waves 4-7 -- one wave of each
    // per SIMD) run ONE BARRIER apart, so that while one group issues the 8 MFMAs of a k-substep (256 matrix-pipe cycles, s_setprio 1)
    // the other issues its LDS fragment reads and its LDS-DMA pieces, and vice versa: a K-tile is 4 phases of
    //     [6 ds_read_b128 (+ DMA pieces of K-tile kt + 1 in phases 0 / 1)]  barrier  [lgkmcnt(0), 8 MFMAs]  barrier
    // Ordering rules (a reader may be one barrier ahead of an issuer): the wave's own DMA of K-tile kt + 1 is waited for (vmcnt 0)
    // BEFORE the first barrier of phase 3 -- every wave of the other group has then passed that barrier before anybody reads the
    // buffer in phase 0 of K-tile kt + 1; the fragment reads of phase 3 are retired (lgkmcnt 0) before that same barrier, so the
    // DMA of K-tile kt + 2 -- issued after the NEXT barrier at the earliest -- cannot overtake a read of the buffer it overwrites.
    auto phase_mfma = [&](const bf16x8(&wf)[2], const bf16x8(&af)[4], bool wait_reads) {
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      if (wait_reads) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_setprio(1);
      mfma8(wf, af);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    };
    auto stage_part = [&](const TileP &p, int buf, int kt, int i0, int i1) {  // pieces i0 .. i1-1 (A and W) of the wave's 4 + 4
      kt += p.rot;
      if (kt >= nk) kt -= nk;
      const uint32_t la = lds0 + (uint32_t)(buf * GEMM_BUFBYTES + wave * 4096);
      for (int i = i0; i < i1; ++i) {
        gemm_dma16(la + i * 1024, p.a_off[i], a_rs, kt * (GEMM_BK * 2));
        gemm_dma16(la + i * 1024 + GEMM_OPBYTES, p.w_off[i], w_rs, kt * (GEMM_BK * 2));
      }
    };
    if (wm == 1) __builtin_amdgcn_s_barrier();  // group 1 runs one barrier behind group 0
    for (int kt = 0; kt < nk; ++kt) {
      const int buf = kt & 1;
      const char *lb = smem + buf * GEMM_BUFBYTES;
      const bool more_k = kt + 1 < nk && kDma;
      read_frags(lb, 0, wf0, af0);
      __builtin_amdgcn_sched_barrier(0);
      if (more_k) stage_part(cur, buf ^ 1, kt + 1, 0, GEMM_PP == 2 ? 4 : 2);
      phase_mfma(wf0, af0, true);
      read_frags(lb, 1, wf1, af1);
      __builtin_amdgcn_sched_barrier(0);
      if (more_k && GEMM_PP != 2) stage_part(cur, buf ^ 1, kt + 1, 2, 4);
      phase_mfma(wf1, af1, true);
      read_frags(lb, 2, wf0, af0);
      phase_mfma(wf0, af0, true);
      read_frags(lb, 3, wf1, af1);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      phase_mfma(wf1, af1, false);
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();  // both groups have executed the same number of barriers again
#else
    read_frags(smem, 0, wf0, af0);
    for (int kt = 0; kt < nk; ++kt) {
      const int buf = kt & 1;
      const char *lb = smem + buf * GEMM_BUFBYTES;
      // (entry: the fragments of substep 0 of this K-tile are being read into set 0; set 1 holds substep 3 of the previous one)
      if (kt > 0) mfma8(wf1, af1);
      __builtin_amdgcn_sched_barrier(0);
      if (kt + 1 < nk && kDma) stage_tile(cur, buf ^ 1, kt + 1);
      __builtin_amdgcn_sched_barrier(0);
      read_frags(lb, 1, wf1, af1);
      __builtin_amdgcn_sched_barrier(0);
      mfma8(wf0, af0);
      __builtin_amdgcn_sched_barrier(0);
      read_frags(lb, 2, wf0, af0);
      __builtin_amdgcn_sched_barrier(0);
      mfma8(wf1, af1);
      __builtin_amdgcn_sched_barrier(0);
      read_frags(lb, 3, wf1, af1);
      __builtin_amdgcn_sched_barrier(0);
      mfma8(wf0, af0);
      __builtin_amdgcn_sched_barrier(0);
      // every read of this K-tile has landed (the DMA of K-tile kt + 2 may overwrite it after the barrier), the wave's own
      // pieces of K-tile kt + 1 have landed
#if GEMM_STAMP
      const bool st_on = blockIdx.x == 16 && step == 2;
      unsigned long long tB = 0, tC = 0, tA = 0;
      if (st_on) { tB = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (st_on) { tC = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
      __builtin_amdgcn_s_barrier();
      if (st_on) {
        tA = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) { unsigned long long *sp = g_stamps + ((size_t)wave * 64 + kt) * 4; sp[0] = tB; sp[1] = tC; sp[2] = tA; }
      }
#else
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#endif
      __builtin_amdgcn_sched_barrier(0);
      if (kt + 1 < nk) read_frags(smem + (buf ^ 1) * GEMM_BUFBYTES, 0, wf0, af0);
    }

#endif
    // ---- next tile's first K-tile in flight under this tile's epilogue; then the pending MFMA group of the last K-tile
    const bool more = EPI != 3 && can_prefetch && ti + nslots < chunk_len;  // (EPI 3: the LayerNorm epilogue needs the registers)
    TileP nxt;
    float4 nxt_bv;
    if (more) {
      tile_params(ti + nslots, step + 1, nxt);
      nxt_bv = *reinterpret_cast<const float4 *>(bias + nxt.n0 + lane * 4);
      stage_tile(nxt, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
#if !GEMM_PP
    mfma8(wf1, af1);
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
            const int row = mh * 32 + l31;
            const int slot16 = (nl >> 3) ^ (row & 7);
            *reinterpret_cast<uint2 *>(cw + row * 128 + slot16 * 16 + (nl & 4) * 2) = make_uint2(cvt_pk_bf16_f32(v0, v1), cvt_pk_bf16_f32(v2, v3));
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
        const size_t off = GATHER ? ((size_t)m * GEMM_BN + wn * 64 + q * 8) * 2 : ((size_t)m * LDC + n0 + wn * 64 + q * 8) * 2;
        if (GEMM_EABL == 1) {
          asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
        } else if (GATHER || m < M) {
          if (nt_store) {
            const u32x4 vv = {v.x, v.y, v.z, v.w};
            __builtin_nontemporal_store(vv, reinterpret_cast<u32x4 *>(Cb + off));
          } else {
            *reinterpret_cast<uint4 *>(Cb + off) = v;
          }
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

// Outputs larger than the chip's L2 (8 x 4 MiB) are written with non-temporal stores: they cannot stay cached until the
// next launch reads them, and a round of tiles would otherwise push the operand panels out of L2.
static inline int use_nt_store(long M, int N) { return (size_t)M * N * 2 > (32u << 20) ? 1 : 0; }

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
  const int n_cu = gemm_cu_count();
  const int grid = tiles >= n_cu ? n_cu : ((tiles + 7) & ~7);
  const int nt = use_nt_store(M, N);
  if (epilogue == 1)
    hipLaunchKernelGGL(gemm_bf16_kernel<1>, dim3(grid), dim3(512), 0, s, (const u16 *)A, (const u16 *)W, bias, (u16 *)C, (int)M, N,
                       K, tiles_n, tiles, nt);
  else if (epilogue == 2)
    hipLaunchKernelGGL(gemm_bf16_kernel<2>, dim3(grid), dim3(512), 0, s, (const u16 *)A, (const u16 *)W, bias, (u16 *)C, (int)M, N,
                       K, tiles_n, tiles, nt);
  else
    hipLaunchKernelGGL(gemm_bf16_kernel<0>, dim3(grid), dim3(512), 0, s, (const u16 *)A, (const u16 *)W, bias, (u16 *)C, (int)M, N,
                       K, tiles_n, tiles, nt);
  return check_launch("linear_bf16");
}

int unopose_linear_bf16_ld(const void *A, int lda, const void *W, int ldw, const float *bias, void *C, int ldc, long M, int N, int K,
                           int epilogue, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(A && W && bias && C, "linear_bf16_ld: null pointer");
  UNOPOSE_REQUIRE(M >= 1 && M < (1L << 31) && N >= GEMM_BN && N % GEMM_BN == 0 && K >= GEMM_BK && K % GEMM_BK == 0,
                  "linear_bf16_ld: needs N %% 256 == 0 and K %% 64 == 0 (got M=%ld N=%d K=%d)", M, N, K);
  UNOPOSE_REQUIRE(lda >= K && ldw >= K && ldc >= N && lda % 8 == 0 && ldw % 8 == 0 && ldc % 8 == 0,
                  "linear_bf16_ld: row strides must cover the rows and be multiples of 8 elements (lda=%d ldw=%d ldc=%d)", lda, ldw, ldc);
  UNOPOSE_REQUIRE((size_t)M * lda * 2 < (1UL << 32) && (size_t)N * ldw * 2 < (1UL << 32), "linear_bf16_ld: operand larger than 4 GiB");
  UNOPOSE_REQUIRE(epilogue >= 0 && epilogue <= 2, "linear_bf16_ld: epilogue must be 0 (bias), 1 (bias + GELU) or 2 (bias + ReLU)");
  const int tiles_m = cdiv(M, GEMM_BM), tiles_n = N / GEMM_BN, tiles = tiles_m * tiles_n;
  hipStream_t s = (hipStream_t)stream;
  const int n_cu = gemm_cu_count();
  const int grid = tiles >= n_cu ? n_cu : ((tiles + 7) & ~7);
  const int nt = use_nt_store(M, N);
#define UNOPOSE_LD_LAUNCH(E)                                                                                                                \
  hipLaunchKernelGGL(gemm_bf16_kernel<E>, dim3(grid), dim3(512), 0, s, (const u16 *)A, (const u16 *)W, bias, (u16 *)C, (int)M, N, K, tiles_n, \
                     tiles, nt, (const int *)nullptr, (const int *)nullptr, (const u16 *)nullptr, (const float *)nullptr,                     \
                     (const float *)nullptr, 0.f, lda, ldw, ldc)
  if (epilogue == 1)
    UNOPOSE_LD_LAUNCH(1);
  else if (epilogue == 2)
    UNOPOSE_LD_LAUNCH(2);
  else
    UNOPOSE_LD_LAUNCH(0);
#undef UNOPOSE_LD_LAUNCH
  return check_launch("linear_bf16_ld");
}

int unopose_linear_add_layernorm_bf16(const void *A, const void *W, const float *bias, const void *resid, const float *ln_w,
                                      const float *ln_b, float eps, void *C, long M, int K, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(A && W && bias && resid && ln_w && ln_b && C, "linear_add_layernorm_bf16: null pointer");
  UNOPOSE_REQUIRE(M >= 1 && M < (1L << 31) && K >= GEMM_BK && K % GEMM_BK == 0, "linear_add_layernorm_bf16: needs K %% 64 == 0 (got M=%ld K=%d)", M, K);
  UNOPOSE_REQUIRE((size_t)M * K * 2 < (1UL << 32), "linear_add_layernorm_bf16: operand larger than 4 GiB");
  const int tiles = cdiv(M, GEMM_BM);
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
