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

#include "gemm_kernel.h"

using namespace unopose;

// Outputs larger than the chip's L2 (8 x 4 MiB) are written with non-temporal stores: they cannot stay cached until the
// next launch reads them, and a round of tiles would otherwise push the operand panels out of L2.
static inline int use_nt_store(long M, int N) { return (size_t)M * N * 2 > (32u << 20) ? 1 : 0; }

// Shape policy: when the 256 x 256 tiles cannot give 5 / 8 of the CUs one (`UNOPOSE_GEMM_SMALL_TILES` overrides the limit for A/Bs:
// scripts/gemm_policy_ab.sh), the GEMM runs on gemm_small.hip's 128 x 128 (64 x 256 with the LayerNorm epilogue) tiles.
// (Measured and not kept: giving the 256-tile kernel only whole rounds of tiles and the last row panels to the small kernel --
// fc2 at M = 87 936 is 1032 tiles on 256 CUs -- gains 0 - 3 % per shape, 0.07 ms per forward: the few tiles of a last round
// already run faster than those of a full one.)
namespace unopose {
int gemm_small_tiles_limit() {
#ifdef UNOPOSE_PROBE_BUILD  // (scripts/gemm_policy_ab.sh: probe builds only)
  static const int v = [] {
    const char *e = getenv("UNOPOSE_GEMM_SMALL_TILES");
    return e && *e ? atoi(e) : -1;
  }();
#else
  const int v = -1;
#endif
  // default: below 5/8 of the CUs.  At 77 % fill (198 tiles: the 224 x 224 ViT's proj / fc2, M = 16 704) the 256-tile kernel is 8 - 17 %
  // faster than 786 small tiles, at 39 % (100 tiles) the small tiles win by 28 % (profiles/r04_gemm_policy_224.txt)
  return v >= 0 ? v : gemm_cu_count() * 5 / 8;
}
}  // namespace unopose
static int small_tiles_limit() { return unopose::gemm_small_tiles_limit(); }

// Ticket slots of the dynamic tile scheduling: a ring of 1024 slots of 16 ints per device (zeroed once; every launch's last workgroup
// re-zeroes its slot).  Consecutive launches take consecutive slots, so launches of different streams that run at the same time never
// share one (a slot comes round again after 1024 launches: 20 forwards later).
namespace unopose {
int *gemm_sched_slot(hipStream_t stream) {
#ifdef UNOPOSE_PROBE_BUILD  // `UNOPOSE_GEMM_DYN=0`: static tile lists (A/B, probe builds only)
  static const bool on = [] {
    const char *e = getenv("UNOPOSE_GEMM_DYN");
    return !(e && *e == '0');
  }();
  if (!on) return nullptr;
#endif
  // A launch being CAPTURED into a hipGraph gets static tile lists: a slot baked into a graph would be replayed while eager launches of
  // other streams cycle through the same ring (tickets shared between two running launches: tiles skipped or computed twice), and the
  // ring's first-use hipMalloc / hipMemset is not legal under capture.
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return nullptr;
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
  // (1024 slots: a slot comes round again after 1024 launches -- about 17 forwards; fewer than that many launches may be in flight)
  return ring[dev] + (size_t)(next_slot[dev]++ & 1023u) * 16;
}
}  // namespace unopose

// The four-wave, one-wave-per-SIMD kernel of round 5 (a generated, hand-placed instruction stream; it TIES this kernel on the ViT shapes:
// profiles/r05_gemm4w_ablate.txt) lives under scripts/ubench/gemm4w/ with its generator and emulator; its variant builds
// (scripts/ubench/gemm4w/g4w_var.py) compile this file with -DUNOPOSE_PROBE_GEMM4W to route the entry points to it.
#ifdef UNOPOSE_PROBE_GEMM4W
namespace unopose {
bool gemm4w_ok(long M, int N, int K, int lda, int ldw, int ldc, int epilogue);
int gemm4w_linear(const void *A, int lda, const void *W, int ldw, const float *bias, void *C, int ldc, long M, int N, int K, int epilogue, int nt,
                  int *sched, hipStream_t s);
}  // namespace unopose
static int g_use_4w = 0;
extern "C" int unopose_gemm4w_enable(int on) {
  const int was = g_use_4w;
  if (on >= 0) g_use_4w = on > 2 ? 2 : on;  // 0: off, 1: shapes with at least one tile per CU, 2: every shape the stream supports
  return was;
}
#endif

static int linear_bf16_dispatch(const void *A, int lda, const void *W, int ldw, const float *bias, void *C, int ldc, long M, int N, int K,
                                int epilogue, hipStream_t s, const char *what) {
  const int tiles_n = N / GEMM_BN;
#ifdef GEMM_PROBE_SKIP_TAIL  // timing probe (WRONG results): the tiles past the last whole round of the CUs are not computed -- what a perfect
                             // split of the tail could gain in the pipelined step (scripts/build_variant.py notail -DGEMM_PROBE_SKIP_TAIL)
  const int tiles_all = cdiv(M, GEMM_BM) * tiles_n, ncu_p = gemm_cu_count();
  const int tiles = (tiles_all > ncu_p && tiles_all % ncu_p <= ncu_p / 8) ? tiles_all - tiles_all % ncu_p : tiles_all;
#else
  const int tiles = cdiv(M, GEMM_BM) * tiles_n;
#endif
#ifdef UNOPOSE_PROBE_GEMM4W
  if (g_use_4w == 2 && gemm4w_ok(M, N, K, lda, ldw, ldc, epilogue)) {  // (forced: small tile counts go through the stream as well)
    if (int *const sched4 = gemm_sched_slot(s)) {
      gemm4w_linear(A, lda, W, ldw, bias, C, ldc, M, N, K, epilogue, use_nt_store(M, N), sched4, s);
      return check_launch(what);
    }
  }
#endif
  if (tiles < small_tiles_limit()) return gemm_small_linear(A, W, bias, C, M, N, K, lda, ldw, ldc, epilogue, s);
  const int n_cu = gemm_cu_count();
  const int grid = tiles >= n_cu ? n_cu : ((tiles + 7) & ~7);
  const int nt = use_nt_store(M, N);
#define UNOPOSE_LD_LAUNCH(E)                                                                                                                \
  hipLaunchKernelGGL(gemm256_kernel<E>, dim3(grid), dim3(512), 0, s, (const u16 *)A, (const u16 *)W, bias, (u16 *)C, (int)M, N, K, tiles_n, \
                     tiles, nt, (const int *)nullptr, (const int *)nullptr, (const u16 *)nullptr, (const float *)nullptr,                     \
                     (const float *)nullptr, 0.f, lda, ldw, ldc, sched)
#ifdef UNOPOSE_PROBE_GEMM4W
  if (g_use_4w && tiles >= n_cu && gemm4w_ok(M, N, K, lda, ldw, ldc, epilogue)) {
    if (int *const sched4 = gemm_sched_slot(s)) {
      gemm4w_linear(A, lda, W, ldw, bias, C, ldc, M, N, K, epilogue, nt, sched4, s);
      return check_launch(what);
    }
  }
#endif
  int *const sched = tiles > grid ? gemm_sched_slot(s) : nullptr;  // (one tile per workgroup: nothing to schedule)
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
  hipLaunchKernelGGL((gemm256_kernel<3, false>), dim3(grid), dim3(512), 0, (hipStream_t)stream, (const u16 *)A, (const u16 *)W, bias,
                     (u16 *)C, (int)M, GEMM_BN, K, 1, tiles, use_nt_store(M, GEMM_BN), (const int *)nullptr, (const int *)nullptr,
                     (const u16 *)resid, ln_w, ln_b, eps);
  return check_launch("linear_add_layernorm_bf16");
}

int unopose_linear_bf16_kv_vt(const void *A, const void *W, const float *bias, void *C, void *vt, long M, int N, int K, int tokens, int key_pad,
                               unopose_stream_t stream) {
  UNOPOSE_REQUIRE(A && W && bias && C && vt, "linear_bf16_kv_vt: null pointer");
  UNOPOSE_REQUIRE(M >= 1 && M < (1L << 31) && N >= 256 && N % 128 == 0 && K >= GEMM_BK && K % GEMM_BK == 0,
                  "linear_bf16_kv_vt: needs N %% 128 == 0 (>= 256) and K %% 64 == 0 (got M=%ld N=%d K=%d)", M, N, K);
  UNOPOSE_REQUIRE(tokens >= 1 && key_pad >= tokens && M % tokens == 0 && key_pad - tokens <= 64, "linear_bf16_kv_vt: M must be whole clouds of `tokens` rows, key_pad in [tokens, tokens + 64] (got %ld, %d, %d)", M, tokens, key_pad);
  UNOPOSE_REQUIRE((size_t)M * K * 2 < (1UL << 32) && (size_t)N * K * 2 < (1UL << 32) && (size_t)M * N * 2 < (1UL << 32), "linear_bf16_kv_vt: operand larger than 4 GiB");
  return gemm_small_linear_vt(A, W, bias, C, vt, M, N, K, tokens, key_pad, (hipStream_t)stream);
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
  hipLaunchKernelGGL((gemm256_kernel<0, true>), dim3(grid), dim3(512), 0, (hipStream_t)stream, (const u16 *)A, (const u16 *)W, bias,
                     (u16 *)C, (int)M, N, K, 1, 0, 0, row_list, tile_info);
  return check_launch("linear_bf16_gather");
}

// ---- round 6: the ViT's residual + LayerNorm passes folded into the GEMMs around them (gemm_kernel.h EPI 5 / 6 / 7) -------------------
static int fold_launch_shape(long M, int N, int *grid, int *tiles, int *tiles_n) {
  *tiles_n = N / GEMM_BN;
  *tiles = cdiv(M, GEMM_BM) * *tiles_n;
  const int n_cu = gemm_cu_count();
  *grid = *tiles >= n_cu ? n_cu : ((*tiles + 7) & ~7);
  return use_nt_store(M, N);
}

// Start offset of every second workgroup of the residual-epilogue GEMM, in 1/8 ticks (100 MHz) per K-tile: half a tile's K loop (a K-tile
// takes ~2.1 us on this part) = 8 * 105.  Tuned / measured in scripts/ubench/lnfold_ab.py.
static int g_fold_stagger = 8 * 105;
int unopose_gemm_fold_stagger(int eighth_ticks_per_ktile) {
  const int was = g_fold_stagger;
  if (eighth_ticks_per_ktile >= 0) g_fold_stagger = eighth_ticks_per_ktile;
  return was;
}

int unopose_linear_bf16_residual(const void *A, const void *W, const float *bias, float *xres, void *xb, float *stats, long M, int N, int K,
                                 unopose_stream_t stream) {
  UNOPOSE_REQUIRE(A && W && bias && xres && xb && stats, "linear_bf16_residual: null pointer");
  UNOPOSE_REQUIRE(M >= 1 && M < (1L << 31) && N >= GEMM_BN && N % GEMM_BN == 0 && K >= GEMM_BK && K % GEMM_BK == 0,
                  "linear_bf16_residual: needs N %% 256 == 0 and K %% 64 == 0 (got M=%ld N=%d K=%d)", M, N, K);
  UNOPOSE_REQUIRE((size_t)M * K * 2 < (1UL << 32) && (size_t)N * K * 2 < (1UL << 32) && (size_t)M * N * 4 < (1UL << 31), "linear_bf16_residual: operand too large for 32-bit buffer offsets");
  int grid, tiles, tiles_n;
  const int nt = fold_launch_shape(M, N, &grid, &tiles, &tiles_n);
  hipStream_t s = (hipStream_t)stream;
  int *const sched = tiles > grid ? gemm_sched_slot(s) : nullptr;
  hipLaunchKernelGGL((gemm256_kernel<5>), dim3(grid), dim3(512), 0, s, (const u16 *)A, (const u16 *)W, bias, (u16 *)xb, (int)M, N, K, tiles_n, tiles, nt,
                     (const int *)nullptr, (const int *)nullptr, (const u16 *)nullptr, (const float *)nullptr, (const float *)nullptr, 0.f, 0, 0, 0, sched,
                     (void *)xres, (const float *)nullptr, (float2 *)stats, 0, tiles > grid ? g_fold_stagger * (K / GEMM_BK) / 8 : 0);
  return check_launch("linear_bf16_residual");
}

int unopose_linear_bf16_lnfold(const void *A, const void *W, const float *dvec, const float *cvec, const float *stats, int nparts, float eps, void *C,
                               long M, int N, int K, int gelu, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(A && W && dvec && cvec && stats && C, "linear_bf16_lnfold: null pointer");
  UNOPOSE_REQUIRE(M >= 1 && M < (1L << 31) && N >= GEMM_BN && N % GEMM_BN == 0 && K >= GEMM_BK && K % GEMM_BK == 0,
                  "linear_bf16_lnfold: needs N %% 256 == 0 and K %% 64 == 0 (got M=%ld N=%d K=%d)", M, N, K);
  UNOPOSE_REQUIRE(nparts >= 1 && nparts <= GEMM_LNF_MAXPARTS && nparts * GEMM_BN == K, "linear_bf16_lnfold: needs nparts = K / 256 <= %d (got nparts=%d K=%d)", GEMM_LNF_MAXPARTS, nparts, K);
  UNOPOSE_REQUIRE((size_t)M * K * 2 < (1UL << 32) && (size_t)N * K * 2 < (1UL << 32) && (size_t)M * N * 2 < (1UL << 32), "linear_bf16_lnfold: operand larger than 4 GiB");
  int grid, tiles, tiles_n;
  const int nt = fold_launch_shape(M, N, &grid, &tiles, &tiles_n);
  hipStream_t s = (hipStream_t)stream;
  int *const sched = tiles > grid ? gemm_sched_slot(s) : nullptr;
#define UNOPOSE_LNF_LAUNCH(E)                                                                                                                      \
  hipLaunchKernelGGL((gemm256_kernel<E>), dim3(grid), dim3(512), 0, s, (const u16 *)A, (const u16 *)W, dvec, (u16 *)C, (int)M, N, K, tiles_n, tiles, nt, \
                     (const int *)nullptr, (const int *)nullptr, (const u16 *)nullptr, (const float *)nullptr, (const float *)nullptr, eps, 0, 0, 0, sched, \
                     (void *)nullptr, cvec, (float2 *)stats, nparts)
  if (gelu)
    UNOPOSE_LNF_LAUNCH(7);
  else
    UNOPOSE_LNF_LAUNCH(6);
#undef UNOPOSE_LNF_LAUNCH
  return check_launch("linear_bf16_lnfold");
}

}  // extern "C"
