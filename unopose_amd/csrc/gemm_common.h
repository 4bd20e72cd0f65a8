// Pieces shared by the two linear-layer GEMMs of libunopose_hip.so (gemm.hip: bf16; gemm_f32.hip: fp32-class, bf16 x 3).
// Both kernels use the same workgroup shape (512 threads, one 256 x 256 output tile at a time, one workgroup per CU),
// the same LDS image of an operand stage ([256 rows][128 B], 16-byte chunks XOR-swizzled by row on the SOURCE address of
// the LDS-DMA) and the same persistent, lock-stepped tile walk.
#pragma once
#include <stdlib.h>

#include "common.h"

namespace unopose {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define GEMM_BM 256
#define GEMM_BN 256
#define GEMM_ROWB 128                     // bytes of one operand row per stage (bf16: 64 k; fp32-class: 32 k as hi | lo)
#define GEMM_OPBYTES (256 * GEMM_ROWB)    // one operand stage: 32 KiB
#define GEMM_BUFBYTES (2 * GEMM_OPBYTES)  // A + W: 64 KiB
#ifndef GEMM_GM
#define GEMM_GM 1  // row panels per group of the tile order (1: row-major, the tiles sharing an A panel are neighbours in the sequence)
#endif
#ifndef GEMM_SKEW
#define GEMM_SKEW 1  // tiles sharing a panel start 0..SKEW-1 K-tiles apart (1: together -- measured: least fabric traffic, same time)
#endif

// 0.5 x (1 + erf(x / sqrt 2)); erf(z) = 1 - (a1 t + ... + a5 t^5) exp(-z^2), t = 1 / (1 + p z), z >= 0
// (Abramowitz-Stegun 7.1.26, |err| < 1.5e-7): the fp32-class epilogue.
__device__ __forceinline__ float gelu_erf(float x) {
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

// GELU for a bf16 result: x Phi(x) with Phi(x) ~ 1 / (1 + exp(-x (c0 + c1 x^2 + c2 x^4))), the coefficients a minimax fit of
// the exact erf form (max |error| 2.5e-5 over all x -- 1/80 of a bf16 ulp at 0.5; tests/test_gemm_gpu.py); x^2 clamped where
// the polynomial would turn over (|x| = 8: Phi is 0 / 1 to 1e-12 there).  7 VALU + 2 transcendentals instead of 12 + 2:
// the epilogue of fc1 is VALU-bound.
__device__ __forceinline__ float gelu_bf16_class(float x) {
  const float x2 = fminf(x * x, 64.f);
  // coefficients pre-multiplied by -log2(e): e = 2^(-x p(x^2) log2 e) = exp(-x p)
  float p = fmaf(1.0142630e-3f, x2, -0.10677572f);
  p = fmaf(p, x2, -2.3011213f);
  const float e = __builtin_amdgcn_exp2f(x * p);
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// LDS-DMA of one 1-KiB piece (8 rows x 128 B): destination = wave-uniform LDS byte address (M0) + lane * 16, source = buffer
// descriptor + per-lane byte offset (VGPR) + scalar offset.  Issued from inline asm: the compiler does not see an LDS write
// and so keeps its own s_waitcnt vmcnt out of the LDS reads; the waits on DMA data are the explicit vmcnt + barrier pairs of
// the kernels.
template <int POL = 0>  // cache policy of the load: 0 default, 1 nt (streamed: first to leave L2), 2 sc1, 3 sc0 sc1 nt
__device__ __forceinline__ void gemm_dma16(uint32_t lds_byte, uint32_t vo, __amdgpu_buffer_rsrc_t rs, int so) {
  unsigned keep;
#define GEMM_DMA16_ASM(POLSTR)                                                                                                          \
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen" POLSTR " lds\n\ts_mov_b32 m0, %0" \
               : "=&s"(keep)                                                                                                            \
               : "s"(__builtin_amdgcn_readfirstlane(lds_byte)), "v"(vo), "s"(rs), "s"(__builtin_amdgcn_readfirstlane(so))               \
               : "memory")
  if constexpr (POL == 1) GEMM_DMA16_ASM(" nt");
  else if constexpr (POL == 2) GEMM_DMA16_ASM(" sc1");
  else if constexpr (POL == 3) GEMM_DMA16_ASM(" sc0 sc1 nt");
  else GEMM_DMA16_ASM("");
#undef GEMM_DMA16_ASM
}

// Small-tile kernels (gemm_small.hip): 128 x 128 tiles for epilogue 0 / 1 / 2, 64 x 256 tiles for the residual + LayerNorm epilogue.
int gemm_small_linear(const void *A, const void *W, const float *bias, void *C, long M, int N, int K, int lda, int ldw, int ldc, int epilogue,
                      hipStream_t s);
int gemm_small_linear_vt(const void *A, const void *W, const float *bias, void *C, void *vt, long M, int N, int K, int tokens, int key_pad, hipStream_t s);
int gemm_small_linear_ln(const void *A, const void *W, const float *bias, const void *resid, const float *ln_w, const float *ln_b, float eps,
                         void *C, long M, int K, hipStream_t s);

int gemm_small_linear_f32(const void *As, const void *Ws, const float *bias, float *C, void *Cs, long M, int N, int K, int epilogue, hipStream_t s);
// 256 x 256 tile counts below this run on the small-tile kernels (default: 5 / 8 of the CU count; UNOPOSE_GEMM_SMALL_TILES)
int gemm_small_tiles_limit();

// A ticket slot for one launch of the persistent kernel with dynamic tile scheduling (gemm.hip), or nullptr (static tile lists).
int *gemm_sched_slot(hipStream_t stream);

// One persistent workgroup per CU of the CURRENT device (a multiple of 8: the tile walk is per XCD).
inline int gemm_cu_count() {
  static int n_cu[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (n_cu[dev] == 0) {
    int cu = 0;
    if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu < 8) cu = 256;
    n_cu[dev] = cu & ~7;
#ifdef UNOPOSE_PROBE_BUILD
    const char *e = getenv("UNOPOSE_GEMM_CUS");  // probe: persistent grid capped below the CU count (leaves CUs to another stream)
    if (e && *e && atoi(e) >= 8) n_cu[dev] = min(n_cu[dev], atoi(e) & ~7);
#endif
  }
  return n_cu[dev];
}

}  // namespace unopose
