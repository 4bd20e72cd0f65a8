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
                                                            float eps, void *__restrict__ out, long ldo) {
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
        reinterpret_cast<u16 *>(out)[(size_t)r * ldo + c] = fu_f2bf(y);
      else
        reinterpret_cast<float *>(out)[(size_t)r * ldo + c] = y;
    }
  }
}

// The same, four contiguous channels per lane (16-byte fp32 / 8-byte bf16 accesses): C % 4 == 0, ldo % 4 == 0, C <= 1024.
// Same arithmetic per element as add_layernorm_kernel; the row sums run over a different partition of the row.
template <bool BF>
__device__ __forceinline__ float4 fu_load4(const void *p, size_t i) {
  if (BF) {
    const uint2 r = *reinterpret_cast<const uint2 *>(reinterpret_cast<const u16 *>(p) + i);
    return make_float4(fu_bf2f((u16)(r.x & 0xFFFF)), fu_bf2f((u16)(r.x >> 16)), fu_bf2f((u16)(r.y & 0xFFFF)), fu_bf2f((u16)(r.y >> 16)));
  }
  return *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(p) + i);
}
template <bool A_BF16, bool B_BF16, bool HAS_B, bool OUT_BF16>
__global__ __launch_bounds__(256) void add_layernorm_vec4_kernel(const void *__restrict__ a, const void *__restrict__ b,
                                                                 const float *__restrict__ w, const float *__restrict__ bias, long rows,
                                                                 int C, float eps, void *__restrict__ out, long ldo) {
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  float4 v[4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = (i * 64 + lane) * 4;
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < C) {
      x = fu_load4<A_BF16>(a, (size_t)r * C + c);
      if (HAS_B) {
        const float4 y = fu_load4<B_BF16>(b, (size_t)r * C + c);
        x.x += y.x;
        x.y += y.y;
        x.z += y.z;
        x.w += y.w;
      }
    }
    v[i] = x;
    s += (x.x + x.y) + (x.z + x.w);
  }
  const float mean = wave_sum_f32(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if ((i * 64 + lane) * 4 < C) {
      const float d0 = v[i].x - mean, d1 = v[i].y - mean, d2 = v[i].z - mean, d3 = v[i].w - mean;
      q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
  }
  const float rstd = rsqrtf(wave_sum_f32(q) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < C) {
      const float4 wv = *reinterpret_cast<const float4 *>(w + c), bv = *reinterpret_cast<const float4 *>(bias + c);
      const float y0 = (v[i].x - mean) * rstd * wv.x + bv.x, y1 = (v[i].y - mean) * rstd * wv.y + bv.y;
      const float y2 = (v[i].z - mean) * rstd * wv.z + bv.z, y3 = (v[i].w - mean) * rstd * wv.w + bv.w;
      if (OUT_BF16) {
        uint2 o;
        o.x = (uint32_t)fu_f2bf(y0) | ((uint32_t)fu_f2bf(y1) << 16);
        o.y = (uint32_t)fu_f2bf(y2) | ((uint32_t)fu_f2bf(y3) << 16);
        *reinterpret_cast<uint2 *>(reinterpret_cast<u16 *>(out) + (size_t)r * ldo + c) = o;
      } else {
        *reinterpret_cast<float4 *>(reinterpret_cast<float *>(out) + (size_t)r * ldo + c) = make_float4(y0, y1, y2, y3);
      }
    }
  }
}

// x[r,:] += gamma * y[r,:] (fp32 residual stream, in place) AND out[r,:] = LayerNorm(x[r,:]) (bf16):
// the LayerScale residual of one ViT branch fused with the LayerNorm that opens the next branch, so the
// residual stream is read once instead of twice.  One wavefront per row, C <= 1024.
__global__ __launch_bounds__(256) void scale_residual_layernorm_kernel(float *__restrict__ x, const u16 *__restrict__ y,
                                                                       const float *__restrict__ gamma,
                                                                       const float *__restrict__ w,
                                                                       const float *__restrict__ bias, long rows,
                                                                       int C, float eps, u16 *__restrict__ out) {
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  float v[16];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {  // 4 channels per lane per step: 16-byte / 8-byte accesses
    const int c = (i * 64 + lane) * 4;
    float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < C) {
      xv = *reinterpret_cast<const float4 *>(x + (size_t)r * C + c);
      const uint2 yv = *reinterpret_cast<const uint2 *>(y + (size_t)r * C + c);
      const float4 g = *reinterpret_cast<const float4 *>(gamma + c);
      xv.x += g.x * fu_bf2f((u16)(yv.x & 0xFFFF));
      xv.y += g.y * fu_bf2f((u16)(yv.x >> 16));
      xv.z += g.z * fu_bf2f((u16)(yv.y & 0xFFFF));
      xv.w += g.w * fu_bf2f((u16)(yv.y >> 16));
      *reinterpret_cast<float4 *>(x + (size_t)r * C + c) = xv;
    }
    v[i * 4 + 0] = xv.x; v[i * 4 + 1] = xv.y; v[i * 4 + 2] = xv.z; v[i * 4 + 3] = xv.w;
    s += (xv.x + xv.y) + (xv.z + xv.w);
  }
  const float mean = wave_sum_f32(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < C) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[i * 4 + e] - mean;
        q += d * d;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum_f32(q) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < C) {
      const float4 wv = *reinterpret_cast<const float4 *>(w + c), bv = *reinterpret_cast<const float4 *>(bias + c);
      uint2 o;
      o.x = (uint32_t)fu_f2bf((v[i * 4 + 0] - mean) * rstd * wv.x + bv.x) |
            ((uint32_t)fu_f2bf((v[i * 4 + 1] - mean) * rstd * wv.y + bv.y) << 16);
      o.y = (uint32_t)fu_f2bf((v[i * 4 + 2] - mean) * rstd * wv.z + bv.z) |
            ((uint32_t)fu_f2bf((v[i * 4 + 3] - mean) * rstd * wv.w + bv.w) << 16);
      *reinterpret_cast<uint2 *>(out + (size_t)r * C + c) = o;
    }
  }
}

// The fp32 twin (the reference's default precision, csrc/gemm_f32.hip): x[r,:] += gamma * y[r,:] with y FP32 (y == NULL: no update),
// and LayerNorm(x[r,:]) written in the SPLIT layout the fp32-class GEMM reads (per row and 32-channel block one 128-byte line
// [hi (32 bf16) | lo (32 bf16)]; out == NULL: residual update only).  Replaces LayerScale multiply + residual add + LayerNorm +
// split pass (four trips over the residual stream) of a ViT block branch.
__global__ __launch_bounds__(256) void scale_residual_layernorm_f32_kernel(float *__restrict__ x, const float *__restrict__ y,
                                                                           const float *__restrict__ gamma, const float *__restrict__ w,
                                                                           const float *__restrict__ bias, long rows, int C, float eps,
                                                                           char *__restrict__ out, long out_ld) {
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  float v[16];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = (i * 64 + lane) * 4;
    float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < C) {
      xv = *reinterpret_cast<const float4 *>(x + (size_t)r * C + c);
      if (y) {
        const float4 yv = *reinterpret_cast<const float4 *>(y + (size_t)r * C + c);
        const float4 g = *reinterpret_cast<const float4 *>(gamma + c);
        xv.x += g.x * yv.x; xv.y += g.y * yv.y; xv.z += g.z * yv.z; xv.w += g.w * yv.w;
        *reinterpret_cast<float4 *>(x + (size_t)r * C + c) = xv;
      }
    }
    v[i * 4 + 0] = xv.x; v[i * 4 + 1] = xv.y; v[i * 4 + 2] = xv.z; v[i * 4 + 3] = xv.w;
    s += (xv.x + xv.y) + (xv.z + xv.w);
  }
  if (!out) return;
  const float mean = wave_sum_f32(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < C) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[i * 4 + e] - mean;
        q += d * d;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum_f32(q) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < C) {
      const float4 wv = *reinterpret_cast<const float4 *>(w + c), bv = *reinterpret_cast<const float4 *>(bias + c);
      const float n0 = (v[i * 4 + 0] - mean) * rstd * wv.x + bv.x, n1 = (v[i * 4 + 1] - mean) * rstd * wv.y + bv.y;
      const float n2 = (v[i * 4 + 2] - mean) * rstd * wv.z + bv.z, n3 = (v[i * 4 + 3] - mean) * rstd * wv.w + bv.w;
      uint2 h, l;
      h.x = cvt_pk_bf16_f32(n0, n1);
      h.y = cvt_pk_bf16_f32(n2, n3);
      l.x = cvt_pk_bf16_f32(n0 - __uint_as_float(h.x << 16), n1 - __uint_as_float(h.x & 0xffff0000u));
      l.y = cvt_pk_bf16_f32(n2 - __uint_as_float(h.y << 16), n3 - __uint_as_float(h.y & 0xffff0000u));
      // channels c .. c+3 = half of the 8-element chunk c / 8 of 32-channel block c / 32: hi at +0, lo at +64 of the 128-byte line
      char *line = out + (size_t)r * out_ld + (size_t)(c >> 5) * 128 + ((c >> 3) & 3) * 16 + ((c >> 2) & 1) * 8;
      *reinterpret_cast<uint2 *>(line) = h;
      *reinterpret_cast<uint2 *>(line + 64) = l;
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

// Pixel features: F.interpolate(map, (H,W), bilinear, align_corners=False) + the pixel gather of
// get_chosen_pixel_feats (oneref_feature_extraction.py:221-229, utils/model_utils.py:215-227) fused and
// evaluated ONLY at the chosen pixels.  z is the up-projection output in its native order
// (B, side, side, 4, 4, C): the reference's permute to (B,C,4side,4side) is folded into the index math.
// One wavefront per output point; C = 256 -> 4 channels per lane.
template <bool IN_BF16>
__global__ __launch_bounds__(256) void bilinear_sample_kernel(const void *__restrict__ z,
                                                              const long long *__restrict__ choose, int side, int Np,
                                                              int H, int W, int tok_off, int tok_stride,
                                                              float *__restrict__ out) {
  const int b = blockIdx.y, lane = threadIdx.x & 63;
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= Np) return;
  const long long pix = choose[(size_t)b * Np + p];
  const int hw = 4 * side;
  const BilinearTap tp = bilinear_tap(pix, H, W, hw);
  const int y0 = tp.y0, x0 = tp.x0, y1 = tp.y1, x1 = tp.x1;
  const float ly = tp.ly, lx = tp.lx;
  auto at = [&](int Y, int X) -> size_t {  // element offset of map pixel (Y,X), channel 0
    // patch token (Y>>2, X>>2) of image b sits at row tok_off + index of a (B, tok_stride, 16, 256) tensor
    return ((((size_t)b * tok_stride + tok_off + (size_t)(Y >> 2) * side + (X >> 2))) * 16 + (Y & 3) * 4 + (X & 3)) * 256;
  };
  const size_t o00 = at(y0, x0), o01 = at(y0, x1), o10 = at(y1, x0), o11 = at(y1, x1);
  float v[4][4];
  const size_t offs[4] = {o00, o01, o10, o11};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (IN_BF16) {
      const uint2 r = *reinterpret_cast<const uint2 *>(reinterpret_cast<const u16 *>(z) + offs[k] + lane * 4);
      v[k][0] = fu_bf2f((u16)(r.x & 0xFFFF)); v[k][1] = fu_bf2f((u16)(r.x >> 16));
      v[k][2] = fu_bf2f((u16)(r.y & 0xFFFF)); v[k][3] = fu_bf2f((u16)(r.y >> 16));
    } else {
      const float4 r = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(z) + offs[k] + lane * 4);
      v[k][0] = r.x; v[k][1] = r.y; v[k][2] = r.z; v[k][3] = r.w;
    }
  }
  float4 res;
  float *rp = &res.x;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float top = (1.f - lx) * v[0][c] + lx * v[1][c];
    const float bot = (1.f - lx) * v[2][c] + lx * v[3][c];
    rp[c] = (1.f - ly) * top + ly * bot;
  }
  *reinterpret_cast<float4 *>(out + ((size_t)b * Np + p) * 256 + lane * 4) = res;
}

}  // namespace unopose

using namespace unopose;

extern "C" {

int unopose_add_layernorm(const void *a, int a_bf16, const void *b, int b_bf16, const float *w, const float *bias,
                          long rows, int C, float eps, void *out, int out_bf16, unopose_stream_t stream) {
  return unopose_add_layernorm_strided(a, a_bf16, b, b_bf16, w, bias, rows, C, eps, out, out_bf16, C, stream);
}

int unopose_add_layernorm_strided(const void *a, int a_bf16, const void *b, int b_bf16, const float *w,
                                  const float *bias, long rows, int C, float eps, void *out, int out_bf16, long ld_out,
                                  unopose_stream_t stream) {
  UNOPOSE_REQUIRE(a && w && bias && out, "add_layernorm: null pointer");
  UNOPOSE_REQUIRE(rows >= 0 && C >= 1 && C <= 1024 && ld_out >= C, "add_layernorm: C=%d unsupported (<= 1024, ld_out >= C)", C);
  if (rows == 0) return UNOPOSE_OK;
  dim3 grid((unsigned)((rows + 3) / 4));
  hipStream_t s = (hipStream_t)stream;
  const bool vec4 = C % 4 == 0 && ld_out % 4 == 0;  // every model shape (C = 256, 768); the scalar form covers the rest
#define UNOPOSE_LN(AB, BB, HB, OB)                                                                                               \
  do {                                                                                                                           \
    if (vec4)                                                                                                                    \
      hipLaunchKernelGGL((add_layernorm_vec4_kernel<AB, BB, HB, OB>), grid, dim3(256), 0, s, a, b, w, bias, rows, C, eps, out, ld_out); \
    else                                                                                                                         \
      hipLaunchKernelGGL((add_layernorm_kernel<AB, BB, HB, OB>), grid, dim3(256), 0, s, a, b, w, bias, rows, C, eps, out, ld_out);      \
  } while (0)
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

int unopose_bilinear_sample(const void *z, int z_bf16, const long long *choose, int B, int side, int Np, int H, int W,
                            float *out, unopose_stream_t stream) {
  return unopose_bilinear_sample_tokens(z, z_bf16, choose, B, side, Np, H, W, 0, side * side, out, stream);
}

int unopose_bilinear_sample_tokens(const void *z, int z_bf16, const long long *choose, int B, int side, int Np, int H,
                                   int W, int tok_offset, int tok_stride, float *out, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(z && choose && out, "bilinear_sample: null pointer");
  UNOPOSE_REQUIRE(B >= 0 && side >= 1 && Np >= 0 && H >= 1 && W >= 1 && B <= 65535 && tok_offset >= 0 &&
                      tok_stride >= tok_offset + side * side,
                  "bilinear_sample: bad sizes");
  if (B == 0 || Np == 0) return UNOPOSE_OK;
  dim3 grid(cdiv(Np, 4), B);
  if (z_bf16)
    hipLaunchKernelGGL(bilinear_sample_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, z, choose, side, Np, H, W,
                       tok_offset, tok_stride, out);
  else
    hipLaunchKernelGGL(bilinear_sample_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, z, choose, side, Np, H,
                       W, tok_offset, tok_stride, out);
  return check_launch("bilinear_sample");
}

int unopose_scale_residual_layernorm(float *x, const void *y_bf16, const float *gamma, const float *w, const float *bias,
                                     long rows, int C, float eps, void *out_bf16, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(x && y_bf16 && gamma && w && bias && out_bf16, "scale_residual_layernorm: null pointer");
  UNOPOSE_REQUIRE(rows >= 0 && C >= 4 && C % 4 == 0 && C <= 1024, "scale_residual_layernorm: C must be a multiple of 4, <= 1024");
  if (rows == 0) return UNOPOSE_OK;
  hipLaunchKernelGGL(scale_residual_layernorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, x, (const u16 *)y_bf16, gamma, w, bias, rows, C, eps, (u16 *)out_bf16);
  return check_launch("scale_residual_layernorm");
}

int unopose_scale_residual_layernorm_f32(float *x, const float *y, const float *gamma, const float *w, const float *bias, long rows, int C, float eps,
                                         void *out_split, long out_ld_bytes, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(x && (y || out_split) && (!y || gamma) && (!out_split || (w && bias)), "scale_residual_layernorm_f32: null pointer");
  UNOPOSE_REQUIRE(rows >= 0 && C >= 32 && C % 32 == 0 && C <= 1024, "scale_residual_layernorm_f32: C must be a multiple of 32, <= 1024");
  UNOPOSE_REQUIRE(out_ld_bytes == 0 || (out_ld_bytes >= (long)C * 4 && out_ld_bytes % 128 == 0), "scale_residual_layernorm_f32: output rows must be >= 4 C bytes apart, whole 128-byte lines (got %ld)", out_ld_bytes);
  if (rows == 0) return UNOPOSE_OK;
  hipLaunchKernelGGL(scale_residual_layernorm_f32_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, y, gamma, w,
                     bias, rows, C, eps, (char *)out_split, out_ld_bytes ? out_ld_bytes : (long)C * 4);
  return check_launch("scale_residual_layernorm_f32");
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
