// Depth rasteriser for BOP's Visible Surface Discrepancy (C ABI part 3).
// bop_toolkit_lib/pose_error.py:17-101 renders the object model in the estimated and in the ground-truth pose
// (renderer.render_object(...)["depth"]) and compares the two depth maps with the test image; the reference delegates that to
// an OpenGL renderer.  Here: one z-buffer per pose, a thread per (pose, triangle), atomicMin on the bit pattern of the positive
// float depth.  Conventions of the toolkit: camera looks along +z, pixel (x, y) is the ray ((x - cx) / fx, (y - cy) / fy, 1)
// (integer pixel coordinates are pixel centres, misc.py:142-162), depth = z of the nearest surface, 0 where nothing projects.
// No clipping: triangles with a vertex at z <= 0 are dropped (BOP objects sit in front of the camera).
#include <algorithm>

#include "common.h"

namespace unopose {

__global__ __launch_bounds__(256) void raster_fill_kernel(uint32_t *__restrict__ buf, long n, uint32_t v) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) buf[i] = v;
}

__global__ __launch_bounds__(256) void raster_depth_kernel(const float *__restrict__ verts, const int *__restrict__ faces, int F,
                                                          const float *__restrict__ Rt, const float *__restrict__ K4, int H, int W,
                                                          uint32_t *__restrict__ zbuf) {
  const int f = blockIdx.x * 256 + threadIdx.x, p = blockIdx.y;
  if (f >= F) return;
  const float *R = Rt + (size_t)p * 12, *t = R + 9, *K = K4 + (size_t)p * 4;
  float X[3], Y[3], Z[3], u[3], v[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float *q = verts + (size_t)faces[f * 3 + i] * 3;
    X[i] = R[0] * q[0] + R[1] * q[1] + R[2] * q[2] + t[0];
    Y[i] = R[3] * q[0] + R[4] * q[1] + R[5] * q[2] + t[1];
    Z[i] = R[6] * q[0] + R[7] * q[1] + R[8] * q[2] + t[2];
    if (!(Z[i] > 1e-6f)) return;
    u[i] = K[0] * X[i] / Z[i] + K[2];
    v[i] = K[1] * Y[i] / Z[i] + K[3];
  }
  const float area = (u[1] - u[0]) * (v[2] - v[0]) - (u[2] - u[0]) * (v[1] - v[0]);
  if (fabsf(area) < 1e-12f) return;
  const int x0 = max(0, (int)ceilf(fminf(fminf(u[0], u[1]), u[2]))), x1 = min(W - 1, (int)floorf(fmaxf(fmaxf(u[0], u[1]), u[2])));
  const int y0 = max(0, (int)ceilf(fminf(fminf(v[0], v[1]), v[2]))), y1 = min(H - 1, (int)floorf(fmaxf(fmaxf(v[0], v[1]), v[2])));
  const float inv = 1.f / area, iz0 = 1.f / Z[0], iz1 = 1.f / Z[1], iz2 = 1.f / Z[2];
  uint32_t *zb = zbuf + (size_t)p * H * W;
  for (int y = y0; y <= y1; ++y)
    for (int x = x0; x <= x1; ++x) {
      const float px = (float)x, py = (float)y;
      // barycentrics of the pixel centre (edges included; both windings)
      const float b0 = ((u[1] - px) * (v[2] - py) - (u[2] - px) * (v[1] - py)) * inv;
      const float b1 = ((u[2] - px) * (v[0] - py) - (u[0] - px) * (v[2] - py)) * inv;
      const float b2 = 1.f - b0 - b1;
      if (b0 < 0.f || b1 < 0.f || b2 < 0.f) continue;
      const float z = 1.f / (b0 * iz0 + b1 * iz1 + b2 * iz2);  // perspective-correct depth (1/z is linear on the screen)
      if (z > 0.f) atomicMin(zb + (size_t)y * W + x, __float_as_uint(z));
    }
}

__global__ __launch_bounds__(256) void raster_finish_kernel(uint32_t *__restrict__ buf, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    if (buf[i] >= 0x7F800000u) buf[i] = 0u;  // untouched pixels: depth 0
}

}  // namespace unopose

using namespace unopose;

extern "C" int unopose_render_depth(const float *verts, int V, const int *faces, int F, const float *Rt, const float *K4, int P, int H, int W,
                                    float *depth, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(verts && faces && Rt && K4 && depth, "render_depth: null pointer");
  UNOPOSE_REQUIRE(V >= 3 && F >= 1 && P >= 1 && P <= 65535 && H >= 1 && W >= 1, "render_depth: bad sizes (V=%d F=%d P=%d H=%d W=%d)", V, F, P, H, W);
  hipStream_t s = (hipStream_t)stream;
  const long n = (long)P * H * W;
  const int g = (int)std::min<long>((n + 255) / 256, 8192);
  hipLaunchKernelGGL(raster_fill_kernel, dim3(g), dim3(256), 0, s, (uint32_t *)depth, n, 0x7F800000u);
  hipLaunchKernelGGL(raster_depth_kernel, dim3(cdiv(F, 256), P), dim3(256), 0, s, verts, faces, F, Rt, K4, H, W, (uint32_t *)depth);
  hipLaunchKernelGGL(raster_finish_kernel, dim3(g), dim3(256), 0, s, (uint32_t *)depth, n);
  return check_launch("render_depth");
}
