// Small fused row kernels for gfx950 (C ABI part 2): the element-wise glue between the GEMMs of the
// ViT blocks (timm Block: x += ls(attn(norm1(x))), x += ls(mlp(norm2(x))); driven by
// core/unopose/model/oneref_feature_extraction.py:38-41) and of the post-LN transformer layers
// (core/unopose/model/transformer.py:151-193).  Under autocast the reference runs each of these as
// 2-4 separate passes over HBM (LayerNorm in fp32, cast to bf16, scale, add); here each is one pass.
#include <algorithm>

#include "common.h"

namespace unopose {

typedef unsigned short u16;
__device__ __forceinline__ u16 fu_f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (u16)(u >> 16);
}
__device__ __forceinline__ float fu_bf2f(u16 h) { return __uint_as_float(((uint32_t)h) << 16); }

template <bool IN_BF16>
__device__ __forceinline__ float fu_load(const void *p, size_t i) {
  return IN_BF16 ? fu_bf2f(reinterpret_cast<const u16 *>(p)[i]) : reinterpret_cast<const float *>(p)[i];
}

// out[r,:] = LayerNorm(a[r,:] (+ b[r,:])) * w + bias ; one wavefront per row, C <= 64 * 16
template <bool A_BF16, bool B_BF16, bool HAS_B, bool OUT_BF16>
__global__ __launch_bounds__(256) void add_layernorm_kernel(const void *__restrict__ a, const void *__restrict__ b,
                                                            const float *__restrict__ w,
                                                            const float *__restrict__ bias, long rows, int C,
                                                            float eps, void *__restrict__ out) {
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  float v[16];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = i * 64 + lane;
    float x = 0.f;
    if (c < C) {
      x = fu_load<A_BF16>(a, (size_t)r * C + c);
      if (HAS_B) x += fu_load<B_BF16>(b, (size_t)r * C + c);
    }
    v[i] = x;
    s += x;
  }
  const float mean = wave_sum_f32(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = i * 64 + lane;
    const float d = c < C ? v[i] - mean : 0.f;
    q += d * d;
  }
  const float rstd = rsqrtf(wave_sum_f32(q) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = i * 64 + lane;
    if (c < C) {
      const float y = (v[i] - mean) * rstd * w[c] + bias[c];
      if (OUT_BF16)
        reinterpret_cast<u16 *>(out)[(size_t)r * C + c] = fu_f2bf(y);
      else
        reinterpret_cast<float *>(out)[(size_t)r * C + c] = y;
    }
  }
}

// x[r,:] += gamma[:] * y[r,:]   (x fp32 in place, y bf16)  -- LayerScale residual of a ViT block
__global__ __launch_bounds__(256) void scale_residual_kernel(float *__restrict__ x, const u16 *__restrict__ y,
                                                             const float *__restrict__ gamma, long n4, int C) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 xv = reinterpret_cast<float4 *>(x)[i];
    const uint2 yv = reinterpret_cast<const uint2 *>(y)[i];
    const int c = (int)((i * 4) % C);
    const float4 g = *reinterpret_cast<const float4 *>(gamma + c);
    xv.x += g.x * fu_bf2f((u16)(yv.x & 0xFFFF));
    xv.y += g.y * fu_bf2f((u16)(yv.x >> 16));
    xv.z += g.z * fu_bf2f((u16)(yv.y & 0xFFFF));
    xv.w += g.w * fu_bf2f((u16)(yv.y >> 16));
    reinterpret_cast<float4 *>(x)[i] = xv;
  }
}

}  // namespace unopose

using namespace unopose;

extern "C" {

int unopose_add_layernorm(const void *a, int a_bf16, const void *b, int b_bf16, const float *w, const float *bias,
                          long rows, int C, float eps, void *out, int out_bf16, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(a && w && bias && out, "add_layernorm: null pointer");
  UNOPOSE_REQUIRE(rows >= 0 && C >= 1 && C <= 1024, "add_layernorm: C=%d unsupported (<= 1024)", C);
  if (rows == 0) return UNOPOSE_OK;
  dim3 grid((unsigned)((rows + 3) / 4));
  hipStream_t s = (hipStream_t)stream;
#define UNOPOSE_LN(AB, BB, HB, OB)                                                                         \
  hipLaunchKernelGGL((add_layernorm_kernel<AB, BB, HB, OB>), grid, dim3(256), 0, s, a, b, w, bias, rows, C, eps, out)
  const int key = (a_bf16 ? 8 : 0) | (b ? (b_bf16 ? 4 : 0) | 2 : 0) | (out_bf16 ? 1 : 0);
  switch (key) {
    case 0: UNOPOSE_LN(false, false, false, false); break;
    case 1: UNOPOSE_LN(false, false, false, true); break;
    case 2: UNOPOSE_LN(false, false, true, false); break;
    case 3: UNOPOSE_LN(false, false, true, true); break;
    case 6: UNOPOSE_LN(false, true, true, false); break;
    case 7: UNOPOSE_LN(false, true, true, true); break;
    case 8: UNOPOSE_LN(true, false, false, false); break;
    case 9: UNOPOSE_LN(true, false, false, true); break;
    case 10: UNOPOSE_LN(true, false, true, false); break;
    case 11: UNOPOSE_LN(true, false, true, true); break;
    case 14: UNOPOSE_LN(true, true, true, false); break;
    case 15: UNOPOSE_LN(true, true, true, true); break;
    default: UNOPOSE_REQUIRE(false, "add_layernorm: bad dtype combination");
  }
#undef UNOPOSE_LN
  return check_launch("add_layernorm");
}

int unopose_scale_residual(float *x, const void *y_bf16, const float *gamma, long rows, int C,
                           unopose_stream_t stream) {
  UNOPOSE_REQUIRE(x && y_bf16 && gamma, "scale_residual: null pointer");
  UNOPOSE_REQUIRE(rows >= 0 && C >= 4 && C % 4 == 0, "scale_residual: C must be a multiple of 4");
  if (rows == 0) return UNOPOSE_OK;
  const long n4 = rows * C / 4;
  const unsigned blocks = (unsigned)std::min<long>((n4 + 255) / 256, 4096);
  hipLaunchKernelGGL(scale_residual_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, (const u16 *)y_bf16,
                     gamma, n4, C);
  return check_launch("scale_residual");
}

}  // extern "C"
