// Synthetic neighbours for the co-residency finding: which RESOURCE of a kernel sharing the CU makes the frame
// kernels (PE, LRF, Procrustes) change results?  Every kernel here runs 256-thread workgroups for `iters` loop trips and touches one
// resource only; scripts/ubench/coresidency_matrix.py launches them on a second stream beside one victim at a time.
//   0 valu64     fp32 FMA chain, ~16 VGPRs
//   1 valu256    the same with 240 live VGPRs (register-file pressure: at most 2 waves / SIMD)
//   2 mfma       v_mfma_f32_32x32x16_bf16 chain on constant fragments, no memory
//   3 lds        ds_write / ds_read ring in 60 KiB of LDS
//   4 stream     global loads + stores over a private 1 MiB window (L2 / HBM traffic)
//   5 barrier    s_barrier every 8 FMAs
//   6 trans      v_exp / v_rcp / v_sqrt chain (transcendental unit)
//   7 dpp        DPP row_shr / row_bcast reductions in a loop
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void agg_valu64(float *sink, int iters) {
  float x = threadIdx.x * 1e-3f, y = 1.f;
  for (int i = 0; i < iters * 64; ++i) { x = fmaf(x, 1.0000001f, 0.25f); y = fmaf(y, 0.9999999f, x); }
  if (x + y == 12345.f) sink[0] = x;
}
__global__ __launch_bounds__(256) void agg_valu256(float *sink, int iters) {
  float v[240];
#pragma unroll
  for (int k = 0; k < 240; ++k) v[k] = threadIdx.x * 1e-3f + k;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 240; ++k) v[k] = fmaf(v[k], 1.0000001f, 0.25f);
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 240; ++k) s += v[k];
  if (s == 12345.f) sink[0] = s;
}
__global__ __launch_bounds__(256) void agg_mfma(float *sink, int iters) {
  bf16x8 a, b;
#pragma unroll
  for (int k = 0; k < 8; ++k) { a[k] = (__bf16)(0.001f * (threadIdx.x + k)); b[k] = (__bf16)(0.002f * (threadIdx.x - k)); }
  f32x16 c0, c1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
  for (int i = 0; i < iters * 8; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c1, 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r];
  if (s == 12345.f) sink[0] = s;
}
__global__ __launch_bounds__(256) void agg_lds(float *sink, int iters) {
  __shared__ float ring[15 * 1024];
  for (int k = threadIdx.x; k < 15 * 1024; k += 256) ring[k] = k;
  __syncthreads();
  float s = 0.f;
  int p = threadIdx.x;
  for (int i = 0; i < iters * 16; ++i) {
    s += ring[p];
    ring[p] = s * 0.5f;
    p = (p + 257) % (15 * 1024);
  }
  if (s == 12345.f) sink[0] = s;
}
__global__ __launch_bounds__(256) void agg_stream(float *buf, int iters) {
  float4 *w = reinterpret_cast<float4 *>(buf) + (size_t)blockIdx.x * 65536;  // 1 MiB per workgroup
  float4 acc = {0, 0, 0, 0};
  for (int i = 0; i < iters; ++i) {
    const int o = ((i * 256 + threadIdx.x) & 65535);
    const float4 v = w[o];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    w[(o + 32768) & 65535] = acc;
  }
}
__global__ __launch_bounds__(256) void agg_barrier(float *sink, int iters) {
  float x = threadIdx.x * 1e-3f;
  for (int i = 0; i < iters * 8; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) x = fmaf(x, 1.0000001f, 0.25f);
    __syncthreads();
  }
  if (x == 12345.f) sink[0] = x;
}
__global__ __launch_bounds__(256) void agg_trans(float *sink, int iters) {
  float x = 1.f + threadIdx.x * 1e-3f;
  for (int i = 0; i < iters * 16; ++i) {
    x = __builtin_amdgcn_exp2f(x * 1e-3f) + __builtin_amdgcn_rcpf(x + 1.f) + __builtin_amdgcn_sqrtf(x);
  }
  if (x == 12345.f) sink[0] = x;
}
__global__ __launch_bounds__(256) void agg_dpp(float *sink, int iters) {
  float x = threadIdx.x * 1e-3f;
  for (int i = 0; i < iters * 16; ++i) {
    float t = x;
    t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x111, 0xf, 0xf, false));  // row_shr:1
    t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x112, 0xf, 0xf, false));  // row_shr:2
    t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x142, 0xa, 0xf, false));  // row_bcast:15
    t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x143, 0xc, 0xf, false));  // row_bcast:31
    x = t * 1e-3f + 0.5f;
  }
  if (x == 12345.f) sink[0] = x;
}
extern "C" int aggressor_launch(int which, void *buf, int blocks, int iters, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  float *p = (float *)buf;
  switch (which) {
    case 0: hipLaunchKernelGGL(agg_valu64, dim3(blocks), dim3(256), 0, s, p, iters); break;
    case 1: hipLaunchKernelGGL(agg_valu256, dim3(blocks), dim3(256), 0, s, p, iters); break;
    case 2: hipLaunchKernelGGL(agg_mfma, dim3(blocks), dim3(256), 0, s, p, iters); break;
    case 3: hipLaunchKernelGGL(agg_lds, dim3(blocks), dim3(256), 0, s, p, iters); break;
    case 4: hipLaunchKernelGGL(agg_stream, dim3(blocks), dim3(256), 0, s, p, iters); break;
    case 5: hipLaunchKernelGGL(agg_barrier, dim3(blocks), dim3(256), 0, s, p, iters); break;
    case 6: hipLaunchKernelGGL(agg_trans, dim3(blocks), dim3(256), 0, s, p, iters); break;
    default: hipLaunchKernelGGL(agg_dpp, dim3(blocks), dim3(256), 0, s, p, iters); break;
  }
  return (int)hipGetLastError();
}
