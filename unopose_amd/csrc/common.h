// Shared helpers for the gfx950 kernels of libunopose_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/unopose_hip.h"

namespace unopose {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Last error text, readable through unopose_last_error().
void set_error(const char *fmt, ...);

inline int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return UNOPOSE_ELAUNCH;
  }
  return UNOPOSE_OK;
}

#define UNOPOSE_REQUIRE(cond, ...)        \
  do {                                    \
    if (!(cond)) {                        \
      ::unopose::set_error(__VA_ARGS__);  \
      return UNOPOSE_EINVAL;              \
    }                                     \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Opt a kernel into more than 64 KiB of dynamic LDS, once per DEVICE (the attribute belongs to the function on the current device:
// a process that drives several GPUs must set it on each).  `done`: a static bool[64] owned by the call site.
inline int lds_optin(bool (&done)[64], const void *fn, size_t bytes, const char *what) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!done[dev]) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
      set_error("%s: cannot reserve %zu bytes of LDS", what, bytes);
      return UNOPOSE_ELAUNCH;
    }
    done[dev] = true;
  }
  return UNOPOSE_OK;
}

// Packed RNE fp32 -> bf16 (low half = a, high half = b): lowers to ONE v_cvt_pk_bf16_f32 on gfx950.
// Deliberately the compiler's own vector conversion, not inline asm: the hazard recogniser does not look
// inside asm statements, and an asm VALU op next to an in-flight MFMA on overlapping registers silently
// corrupts results (found with a software-pipelined attention variant; scripts/ubench).
typedef float unopose_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 unopose_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cvt_pk_bf16_f32(float a, float b) {
  const unopose_f32x2 v = {a, b};
  union {
    unopose_bf16x2 h;
    uint32_t u;
  } r;
  r.h = __builtin_convertvector(v, unopose_bf16x2);
  return r.u;
}

// ---- bilinear tap of a pixel of the (H, W) crop in the (hw, hw) feature map
// torch upsample_bilinear2d, align_corners=False: src = max(0, (dst + 0.5) * in/out - 0.5)  (oneref_feature_extraction.py:229)
struct BilinearTap {
  int y0, x0, y1, x1;
  float ly, lx;
};
__device__ __forceinline__ BilinearTap bilinear_tap(long long pix, int H, int W, int hw) {
  BilinearTap t;
  const int py = (int)(pix / W), px = (int)(pix - (long long)py * W);
  const float sy = fmaxf(((float)py + 0.5f) * ((float)hw / (float)H) - 0.5f, 0.f);
  const float sx = fmaxf(((float)px + 0.5f) * ((float)hw / (float)W) - 0.5f, 0.f);
  t.y0 = min((int)sy, hw - 1);
  t.x0 = min((int)sx, hw - 1);
  t.y1 = t.y0 < hw - 1 ? t.y0 + 1 : t.y0;
  t.x1 = t.x0 < hw - 1 ? t.x0 + 1 : t.x0;
  t.ly = sy - (float)t.y0;
  t.lx = sx - (float)t.x0;
  return t;
}

#ifndef UNOPOSE_SUM_SHFL
#define UNOPOSE_SUM_SHFL 0
#endif
#ifndef UNOPOSE_ROW_BCAST
#define UNOPOSE_ROW_BCAST 0  // 1: cross-row steps of the wave reductions through DPP row_bcast:15 / :31
#endif
// ---- wave64 DPP reductions (gfx9 row_shr within the 16-lane rows; the four row results combined through v_readlane) --------
// dpp_ctrl: row_shr:n = 0x110+n, row_bcast:15 = 0x142, row_bcast:31 = 0x143.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v, uint32_t identity) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, CTRL, ROW_MASK, 0xF, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float v, float identity) {
  return __int_as_float(
      __builtin_amdgcn_update_dpp(__float_as_int(identity), __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint64_t dpp_u64(uint64_t v) {
  uint32_t lo = dpp_u32<CTRL, ROW_MASK>((uint32_t)v, 0u);
  uint32_t hi = dpp_u32<CTRL, ROW_MASK>((uint32_t)(v >> 32), 0u);
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t umax64(uint64_t a, uint64_t b) { return a > b ? a : b; }

// max over the 16 lanes of each DPP row; valid in lane 15 of the row.
__device__ __forceinline__ uint64_t row_max_u64(uint64_t v) {
  v = umax64(v, dpp_u64<0x111, 0xF>(v));
  v = umax64(v, dpp_u64<0x112, 0xF>(v));
  v = umax64(v, dpp_u64<0x114, 0xF>(v));
  v = umax64(v, dpp_u64<0x118, 0xF>(v));
  return v;
}
// max over all 64 lanes; wave-uniform result (read from lane 63).
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
  v = row_max_u64(v);
#if UNOPOSE_ROW_BCAST
  v = umax64(v, dpp_u64<0x142, 0xA>(v));
  v = umax64(v, dpp_u64<0x143, 0xC>(v));
  uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)v, 63);
  uint32_t hi = __builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 63);
  return ((uint64_t)hi << 32) | lo;
#else
  uint64_t r[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    r[i] = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 16 * i + 15) << 32) |
           (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 16 * i + 15);
  return umax64(umax64(r[0], r[1]), umax64(r[2], r[3]));
#endif
}

#ifndef UNOPOSE_SUM_VGPR
#define UNOPOSE_SUM_VGPR 0  // 1: the four row results reach every lane through ds_bpermute (VGPR to VGPR) instead of v_readlane (SGPR)
#endif
// lane index the compiler cannot see through (keeps ds_bpermute from being folded into v_readlane)
__device__ __forceinline__ int opaque_lane(int l) {
  asm volatile("" : "+v"(l));
  return l;
}
__device__ __forceinline__ float lane_bcast_vgpr(float v, int src_lane) {
  return __int_as_float(__builtin_amdgcn_ds_bpermute(opaque_lane(src_lane) << 2, __float_as_int(v)));
}

__device__ __forceinline__ float wave_sum_f32(float v) {
#if UNOPOSE_SUM_SHFL
  // probe: no DPP at all -- the same pairing tree through ds_bpermute
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d);
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
#endif
  v += dpp_f32<0x111, 0xF>(v, 0.f);
  v += dpp_f32<0x112, 0xF>(v, 0.f);
  v += dpp_f32<0x114, 0xF>(v, 0.f);
  v += dpp_f32<0x118, 0xF>(v, 0.f);
#if UNOPOSE_SUM_VGPR
  {
    const float q0 = lane_bcast_vgpr(v, 15), q1 = lane_bcast_vgpr(v, 31), q2 = lane_bcast_vgpr(v, 47), q3 = lane_bcast_vgpr(v, 63);
    return (q3 + q2) + (q1 + q0);
  }
#endif
#if UNOPOSE_ROW_BCAST
  v += dpp_f32<0x142, 0xA>(v, 0.f);
  v += dpp_f32<0x143, 0xC>(v, 0.f);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
#else
  // the four row sums (lane 15 of every 16-lane row) combined in the order the row_bcast chain used: (r3 + r2) + (r1 + r0)
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 15)), r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 47)), r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
  return (r3 + r2) + (r1 + r0);
#endif
}
__device__ __forceinline__ float wave_max_f32(float v) {
  const float ninf = -__builtin_inff();
  v = fmaxf(v, dpp_f32<0x111, 0xF>(v, ninf));
  v = fmaxf(v, dpp_f32<0x112, 0xF>(v, ninf));
  v = fmaxf(v, dpp_f32<0x114, 0xF>(v, ninf));
  v = fmaxf(v, dpp_f32<0x118, 0xF>(v, ninf));
#if UNOPOSE_ROW_BCAST
  v = fmaxf(v, dpp_f32<0x142, 0xA>(v, ninf));
  v = fmaxf(v, dpp_f32<0x143, 0xC>(v, ninf));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
#else
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 15)), r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 47)), r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
#endif
}

__device__ __forceinline__ int lane_id() {
  return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

}  // namespace unopose
