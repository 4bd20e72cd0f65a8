// bf16 "linear" GEMM, FOUR-WAVE form (round 5): one 256-thread workgroup per CU, ONE wavefront per SIMD, each wave owning the whole
// 512-entry register file of its SIMD -- 128 x 128 outputs per wave in a[0:255] -- with the instruction stream HAND-PLACED by a generator
// (gen4w/kernel.py beside this file, included below as the body of one asm statement per epilogue variant; the C++ around it only
// declares the kernel, its LDS and its kernarg block).  What it changes against the 8-wave kernel of gemm_kernel.h:
//   * the epilogue of tile i runs INSIDE the K loop of tile i + 1.  At the seam the accumulators are drained into 128 VGPRs as packed
//     bf16 (v_accvgpr_read + v_cvt_pk in the MFMA gaps of the seam's two K-tiles, whose MFMA order is quadrant-major so that the 16
//     accumulator blocks finish and restart 16 - 20 MFMAs apart); activation, LDS staging and the whole-row stores follow in the
//     MFMA gaps of the next tile's middle K-tiles.  No phase of a launch runs without MFMAs (the 8-wave kernel spends 9 - 25 % of a
//     launch in its epilogue with the matrix pipe idle);
//   * 512 LDS bytes of fragment reads per MFMA instead of 768 (8 ds_read_b128 feed 16 MFMAs), one barrier per K-tile instead of four;
//   * the bias enters through the matrix pipe (first MFMA of a block: bias-as-three-bf16 x ones, C = 0), GELU is evaluated on the
//     bf16-rounded pre-activation (torch autocast's own order of operations: nn.Linear rounds to bf16, GELU follows).
// Same LDS image, swizzle, LDS-DMA pieces, tile walk, K rotation and ticket scheduling as gemm_kernel.h.  The stream was checked
// on a functional emulator before it ever ran (test_emu_cpu.py beside this file).
#include <hip/hip_runtime.h>

#include "_gen/gemm4w_gen.h"
#include "gemm_common.h"

namespace unopose {

struct G4wArgs {
  uint32_t w[G4W_KARG_DWORDS];
};


template <int EPI, int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm4w_kernel(G4wArgs args) {
  __shared__ __attribute__((aligned(1024))) char smem[G4W_LDS_BYTES];
  const void *karg = (const void *)__builtin_amdgcn_kernarg_segment_ptr();  // (`args` itself is read by the stream: s_load from here)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;  // (0: the kernel's only LDS object)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"  // s32 (the ABI's stack pointer) is on the clobber list: this kernel has no stack and ends with the statement
#define G4W_IN "s"(karg), "s"(wave), "s"(blockIdx.x), "s"(lds0)
#define G4W_CL "memory", "vcc", "scc", G4W_ALL_CLOBBERS
#include "gemm4w_clobbers.h"
  if constexpr (EPI == 0 && NT == 0) {
    asm volatile(
#include "_gen/gemm4w_body_e0_nt0.inc"
        : : G4W_IN : G4W_CL);
  } else if constexpr (EPI == 0 && NT == 1) {
    asm volatile(
#include "_gen/gemm4w_body_e0_nt1.inc"
        : : G4W_IN : G4W_CL);
  } else if constexpr (EPI == 1 && NT == 0) {
    asm volatile(
#include "_gen/gemm4w_body_e1_nt0.inc"
        : : G4W_IN : G4W_CL);
  } else if constexpr (EPI == 1 && NT == 1) {
    asm volatile(
#include "_gen/gemm4w_body_e1_nt1.inc"
        : : G4W_IN : G4W_CL);
  } else if constexpr (EPI == 2 && NT == 0) {
    asm volatile(
#include "_gen/gemm4w_body_e2_nt0.inc"
        : : G4W_IN : G4W_CL);
  } else {
    asm volatile(
#include "_gen/gemm4w_body_e2_nt1.inc"
        : : G4W_IN : G4W_CL);
  }
#undef G4W_IN
#undef G4W_CL
#pragma clang diagnostic pop
}

static inline uint32_t g4w_magic31(uint32_t d) { return (uint32_t)(((1ull << 31) + d - 1) / d); }

// Launch parameters: the tile walk of gemm_kernel.h (XCD chunks of a row-major tile sequence; column blocks when W exceeds an XCD's L2)
// as per-XCD tables and division magics for the scalar code of the stream.  Python twin: gen4w/host.py.
static void g4w_fill_args(G4wArgs &a, const void *A, int lda, const void *W, int ldw, const float *bias, void *C, int ldc, long M, int N, int K,
                          int grid, int *sched) {
  for (int i = 0; i < G4W_KARG_DWORDS; ++i) a.w[i] = 0;
  auto p64 = [&](int at, const void *p) {
    a.w[at] = (uint32_t)((uintptr_t)p & 0xFFFFFFFFu);
    a.w[at + 1] = (uint32_t)((uintptr_t)p >> 32);
  };
  p64(G4W_KA_A, A), p64(G4W_KA_W, W), p64(G4W_KA_BIAS, bias), p64(G4W_KA_C, C), p64(G4W_KA_SCHED, sched);
  const int tiles_n = N / 256, tiles_m = (int)((M + 255) / 256), tiles = tiles_m * tiles_n, nk = K / 64;
  int cw = (size_t)N * K * 2 > (size_t)(4u << 20) ? (2400 * 1024) / (256 * K * 2) : 0;
  if (cw < 3) cw = 0;
  const bool colwalk = cw > 0 && tiles_n > cw && ((tiles_m & 7) == 0 || tiles_m >= 512);
  if (!colwalk) cw = 1;
  const int rq = tiles_m >> 3, rr = tiles_m & 7, cq = tiles >> 3, cr = tiles & 7;
  const int ncb = (tiles_n + cw - 1) / cw, cwl = tiles_n - (ncb - 1) * cw;
  a.w[G4W_KA_M] = (uint32_t)M, a.w[G4W_KA_N] = N, a.w[G4W_KA_K] = K;
  a.w[G4W_KA_LDA] = 2 * lda, a.w[G4W_KA_LDW] = 2 * ldw, a.w[G4W_KA_LDC] = 2 * ldc;
  a.w[G4W_KA_NK] = nk, a.w[G4W_KA_TILES_N] = tiles_n, a.w[G4W_KA_NSLOTS] = grid / 8;
  a.w[G4W_KA_MG_TN] = g4w_magic31(tiles_n), a.w[G4W_KA_MG_NS] = g4w_magic31(grid / 8), a.w[G4W_KA_MG_NK] = g4w_magic31(nk);
  a.w[G4W_KA_CW] = cw, a.w[G4W_KA_MG_CW] = g4w_magic31(cw), a.w[G4W_KA_CWL] = cwl, a.w[G4W_KA_MG_CWL] = g4w_magic31(cwl);
  a.w[G4W_KA_NCB1] = ncb - 1, a.w[G4W_KA_COLWALK] = colwalk ? 1 : 0, a.w[G4W_KA_GRID] = grid;
  for (int x = 0; x < 8; ++x) {
    const int rows_x = rq + (x < rr ? 1 : 0);
    a.w[G4W_KA_CBASE + x] = x < cr ? x * (cq + 1) : cr * (cq + 1) + (x - cr) * cq;
    a.w[G4W_KA_RBASE + x] = x < rr ? x * (rq + 1) : rr * (rq + 1) + (x - rr) * rq;
    a.w[G4W_KA_CLEN + x] = colwalk ? rows_x * tiles_n : cq + (x < cr ? 1 : 0);
    const int pb = rows_x * cw > 0 ? rows_x * cw : 1;
    a.w[G4W_KA_PB + x] = pb, a.w[G4W_KA_MG_PB + x] = g4w_magic31(pb);
  }
}

// Eligibility: the stream's structure needs nk >= F + unrolled mid blocks + pre-last + L; row strides must fit the 14-bit stride field
// of the structured descriptors; the division magics are exact for tiles < 2^15.
bool gemm4w_ok(long M, int N, int K, int lda, int ldw, int ldc, int epilogue) {
  const int nk = K / 64;
  const long tiles = ((M + 255) / 256) * (long)(N / 256);
  return epilogue >= 0 && epilogue <= 2 && nk >= (epilogue == 1 ? G4W_MIN_NK_GELU : G4W_MIN_NK_PLAIN) && 2 * lda < 16384 && 2 * ldw < 16384 &&
         2 * ldc < 16384 && tiles < 32768 && M < (1L << 31);
}

int gemm4w_linear(const void *A, int lda, const void *W, int ldw, const float *bias, void *C, int ldc, long M, int N, int K, int epilogue, int nt,
                  int *sched, hipStream_t s) {
  const int tiles = (int)((M + 255) / 256) * (N / 256);
  const int n_cu = gemm_cu_count();
  const int grid = tiles >= n_cu ? n_cu : ((tiles + 7) & ~7);
  G4wArgs a;
  g4w_fill_args(a, A, lda, W, ldw, bias, C, ldc, M, N, K, grid, sched);
#define G4W_LAUNCH(E, T) hipLaunchKernelGGL((gemm4w_kernel<E, T>), dim3(grid), dim3(256), 0, s, a)
  if (epilogue == 1) {
    if (nt) G4W_LAUNCH(1, 1); else G4W_LAUNCH(1, 0);
  } else if (epilogue == 2) {
    if (nt) G4W_LAUNCH(2, 1); else G4W_LAUNCH(2, 0);
  } else {
    if (nt) G4W_LAUNCH(0, 1); else G4W_LAUNCH(0, 0);
  }
#undef G4W_LAUNCH
  return 0;
}

}  // namespace unopose
