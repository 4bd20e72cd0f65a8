// Do scalar registers of a wave survive while waves of OTHER kernels share its CU?  (the frames of the fused
// geometry kernels differ run to run beside csrc/attn.hip's token attention -- only in quantities the compiler keeps in SGPRs.)
// Every wave parks NS known values in SGPRs (inline asm s_mov so they stay scalar), spins on VALU work, then checks them with VALU
// compares against values recomputed from scratch, and the same for NV values parked in VGPRs.
#include <hip/hip_runtime.h>
#include <stdint.h>
constexpr int NS = 48, NV = 24;
__global__ __launch_bounds__(256) void sgpr_probe_kernel(unsigned *__restrict__ report, int iters, unsigned salt) {
  const unsigned wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const unsigned base = __builtin_amdgcn_readfirstlane(wave * 1000003u + salt);
  unsigned s[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) asm volatile("s_add_u32 %0, %1, %2" : "=s"(s[k]) : "s"(base), "i"(k * 7 + 1));
  unsigned v[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = base + threadIdx.x * 131u + k * 17u;
  float x = (float)threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    x = fmaf(x, 1.0000001f, 0.25f);
#pragma unroll
    for (int k = 0; k < NV; ++k) asm volatile("" : "+v"(v[k]));
  }
  unsigned bad_s = 0, bad_v = 0, first_idx = 0xffffffffu, first_val = 0;
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    unsigned got;
    asm volatile("v_mov_b32 %0, %1" : "=v"(got) : "s"(s[k]));
    const unsigned want = wave * 1000003u + salt + (unsigned)(k * 7 + 1);
    if (got != want) { ++bad_s; if (first_idx == 0xffffffffu) { first_idx = k; first_val = got; } }
  }
#pragma unroll
  for (int k = 0; k < NV; ++k) bad_v += v[k] != wave * 1000003u + salt + threadIdx.x * 131u + k * 17u;
  if (bad_s | bad_v) {
    const unsigned slot = atomicAdd(report, 1u);
    if (slot < 64) {
      unsigned *r = report + 1 + slot * 6;
      r[0] = wave; r[1] = threadIdx.x & 63; r[2] = bad_s; r[3] = bad_v; r[4] = first_idx; r[5] = first_val;
    }
  }
  if (x == 12345.f) report[1023] = 1;
}
extern "C" int sgpr_probe_launch(void *report, int blocks, int iters, unsigned salt, void *stream) {
  hipLaunchKernelGGL(sgpr_probe_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (unsigned *)report, iters, salt);
  return (int)hipGetLastError();
}
