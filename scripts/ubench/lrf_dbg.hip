// Variants of geom.hip's lrf_global_kernel to find what goes wrong beside a library GEMM.
//   RED 0: block_sum_256 / block_max_256 as in geom.hip (DPP wave reduction);  RED 1: __shfl_xor wave reduction
//   LOOP 0: for (i = tid; i < N; i += 256) (divergent last trip);  LOOP 1: uniform trip count, predicated body
#include "../../unopose_amd/csrc/geom.hip"
namespace unopose {
template <int RED>
__device__ __forceinline__ float vsum(float v, float *red) {
  if (RED == 0) return block_sum_256(v, red);
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
template <int RED>
__device__ __forceinline__ float vmax(float v, float *red) {
  if (RED == 0) return block_max_256(v, red);
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
#define LRF_FOR(i) for (int i##0 = 0; i##0 < (LOOP ? ((N + 255) & ~255) : N); i##0 += 256) if (const int i = i##0 + tid; LOOP == 0 ? (i##0 == 0 ? true : true) && i < N : i < N)
template <int RED, int LOOP, bool DBG = false>
__global__ __launch_bounds__(256) void lrf_variant_kernel(const float *__restrict__ pts, int N, float *__restrict__ out, int use_ref_rad, float *__restrict__ dbg = nullptr) {
  __shared__ float red[4];
  const int tid = threadIdx.x;
  const float *P = pts + (size_t)blockIdx.x * N * 3;
  float *O = out + (size_t)blockIdx.x * N * 3;
  const int NP = LOOP ? ((N + 255) & ~255) : N;
  float sx = 0, sy = 0, sz = 0;
  for (int i = tid; i < NP; i += 256) if (i < N) { sx += P[i * 3 + 0]; sy += P[i * 3 + 1]; sz += P[i * 3 + 2]; }
  const float inv_n = 1.f / (float)N;
  const float cx = vsum<RED>(sx, red) * inv_n, cy = vsum<RED>(sy, red) * inv_n, cz = vsum<RED>(sz, red) * inv_n;
  float a00 = 0, a01 = 0, a02 = 0, a11 = 0, a12 = 0, a22 = 0, rmax = 0;
  for (int i = tid; i < NP; i += 256) if (i < N) {
    const float x = cx - P[i * 3 + 0], y = cy - P[i * 3 + 1], z = cz - P[i * 3 + 2];
    a00 += x * x; a01 += x * y; a02 += x * z; a11 += y * y; a12 += y * z; a22 += z * z;
    rmax = fmaxf(rmax, sqrtf(x * x + y * y + z * z));
  }
  a00 = vsum<RED>(a00, red) * inv_n; a01 = vsum<RED>(a01, red) * inv_n; a02 = vsum<RED>(a02, red) * inv_n;
  a11 = vsum<RED>(a11, red) * inv_n; a12 = vsum<RED>(a12, red) * inv_n; a22 = vsum<RED>(a22, red) * inv_n;
  const float r = use_ref_rad ? 1.f : vmax<RED>(rmax, red);
  Vec3 e0, e1, z0; float l0, l1, l2;
  eig_sym3(a00, a01, a02, a11, a12, a22, e0, e1, z0, l0, l1, l2);
  float vote = 0;
  for (int i = tid; i < NP; i += 256) if (i < N) {
    const float x = cx - P[i * 3 + 0], y = cy - P[i * 3 + 1], z = cz - P[i * 3 + 2];
    const float pr = z0.x * x + z0.y * y + z0.z * z;
    vote += (pr > 1e-3f ? 1.f : 0.f) - (pr < -1e-3f ? 1.f : 0.f);
  }
  vote = vsum<RED>(vote, red);
  const Vec3 zp = vote < 0.f ? scale(z0, -1.f) : z0;
  float vx = 0, vy = 0, vz = 0;
  for (int i = tid; i < NP; i += 256) if (i < N) {
    const Vec3 xn = v3(P[i * 3 + 0] - cx, P[i * 3 + 1] - cy, P[i * 3 + 2] - cz);
    const float nrm = dot(zp, xn);
    const Vec3 vi = sub(xn, scale(zp, nrm));
    float alpha = r - sqrtf(dot(xn, xn));
    alpha *= alpha;
    const float ab = alpha * (nrm * nrm);
    vx += ab * vi.x; vy += ab * vi.y; vz += ab * vi.z;
  }
  const float pvx = vx, pvy = vy, pvz = vz;
  vx = vsum<RED>(vx, red); vy = vsum<RED>(vy, red); vz = vsum<RED>(vz, red);
  Vec3 xp, yp;
  finish_frame(zp, v3(vx, vy, vz), xp, yp);
  for (int i = tid; i < N; i += 256) {
    const Vec3 q = v3((P[i * 3 + 0] - cx) / r, (P[i * 3 + 1] - cy) / r, (P[i * 3 + 2] - cz) / r);
    O[i * 3 + 0] = dot(xp, q); O[i * 3 + 1] = dot(yp, q); O[i * 3 + 2] = dot(zp, q);
  }
  if (DBG) {  // after everything else: per-thread view of every block-level scalar + the thread's own partial sums
    float *T = dbg + ((size_t)blockIdx.x * 256 + tid) * 16;
    const float d[16] = {cx, cy, cz, r, vote, zp.x, zp.y, zp.z, vx, vy, vz, pvx, pvy, pvz, a00, a22};
    for (int k = 0; k < 16; ++k) T[k] = d[k];
  }
}

// W: ONE wavefront per cloud -- no LDS, no barriers; reductions by __shfl_xor only
__device__ __forceinline__ float wsum(float v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }
__device__ __forceinline__ float wmax(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o)); return v; }
__global__ __launch_bounds__(64) void lrf_wave_kernel(const float *__restrict__ pts, int N, float *__restrict__ out, float *__restrict__ dbg) {
  const int tid = threadIdx.x;
  const float *P = pts + (size_t)blockIdx.x * N * 3;
  float *O = out + (size_t)blockIdx.x * N * 3;
  float sx = 0, sy = 0, sz = 0;
  for (int i = tid; i < N; i += 64) { sx += P[i * 3 + 0]; sy += P[i * 3 + 1]; sz += P[i * 3 + 2]; }
  const float inv_n = 1.f / (float)N;
  const float cx = wsum(sx) * inv_n, cy = wsum(sy) * inv_n, cz = wsum(sz) * inv_n;
  float a00 = 0, a01 = 0, a02 = 0, a11 = 0, a12 = 0, a22 = 0, rmax = 0;
  for (int i = tid; i < N; i += 64) {
    const float x = cx - P[i * 3 + 0], y = cy - P[i * 3 + 1], z = cz - P[i * 3 + 2];
    a00 += x * x; a01 += x * y; a02 += x * z; a11 += y * y; a12 += y * z; a22 += z * z;
    rmax = fmaxf(rmax, sqrtf(x * x + y * y + z * z));
  }
  a00 = wsum(a00) * inv_n; a01 = wsum(a01) * inv_n; a02 = wsum(a02) * inv_n; a11 = wsum(a11) * inv_n; a12 = wsum(a12) * inv_n; a22 = wsum(a22) * inv_n;
  const float r = wmax(rmax);
  Vec3 e0, e1, z0; float l0, l1, l2;
  eig_sym3(a00, a01, a02, a11, a12, a22, e0, e1, z0, l0, l1, l2);
  float vote = 0;
  for (int i = tid; i < N; i += 64) {
    const float x = cx - P[i * 3 + 0], y = cy - P[i * 3 + 1], z = cz - P[i * 3 + 2];
    const float pr = z0.x * x + z0.y * y + z0.z * z;
    vote += (pr > 1e-3f ? 1.f : 0.f) - (pr < -1e-3f ? 1.f : 0.f);
  }
  vote = wsum(vote);
  const Vec3 zp = vote < 0.f ? scale(z0, -1.f) : z0;
  float vx = 0, vy = 0, vz = 0;
  for (int i = tid; i < N; i += 64) {
    const Vec3 xn = v3(P[i * 3 + 0] - cx, P[i * 3 + 1] - cy, P[i * 3 + 2] - cz);
    const float nrm = dot(zp, xn);
    const Vec3 vi = sub(xn, scale(zp, nrm));
    float alpha = r - sqrtf(dot(xn, xn));
    alpha *= alpha;
    const float ab = alpha * (nrm * nrm);
    vx += ab * vi.x; vy += ab * vi.y; vz += ab * vi.z;
  }
  const float pvx = vx, pvy = vy, pvz = vz;
  vx = wsum(vx); vy = wsum(vy); vz = wsum(vz);
  Vec3 xp, yp;
  finish_frame(zp, v3(vx, vy, vz), xp, yp);
  if (dbg) {
    float *T = dbg + ((size_t)blockIdx.x * 64 + tid) * 16;
    const float d[16] = {cx, cy, cz, r, vote, zp.x, zp.y, zp.z, vx, vy, vz, pvx, pvy, pvz, a00, a22};
    for (int k = 0; k < 16; ++k) T[k] = d[k];
  }
  for (int i = tid; i < N; i += 64) {
    const Vec3 q = v3((P[i * 3 + 0] - cx) / r, (P[i * 3 + 1] - cy) / r, (P[i * 3 + 2] - cz) / r);
    O[i * 3 + 0] = dot(xp, q); O[i * 3 + 1] = dot(yp, q); O[i * 3 + 2] = dot(zp, q);
  }
}
}  // namespace unopose
extern "C" void run_variant(int variant, const float *pts, int B, int N, float *out, void *stream, float *dbg) {
  using namespace unopose;
  hipStream_t s = (hipStream_t)stream;
  switch (variant) {
    case 0: hipLaunchKernelGGL((lrf_variant_kernel<0, 0>), dim3(B), dim3(256), 0, s, pts, N, out, 0); break;
    case 1: hipLaunchKernelGGL((lrf_variant_kernel<1, 0>), dim3(B), dim3(256), 0, s, pts, N, out, 0); break;
    case 2: hipLaunchKernelGGL((lrf_variant_kernel<0, 1>), dim3(B), dim3(256), 0, s, pts, N, out, 0); break;
    case 3: hipLaunchKernelGGL((lrf_variant_kernel<1, 1>), dim3(B), dim3(256), 0, s, pts, N, out, 0); break;
    case 7: hipLaunchKernelGGL(lrf_wave_kernel, dim3(B), dim3(64), 0, s, pts, N, out, dbg); break;
    case 6: hipLaunchKernelGGL((lrf_variant_kernel<1, 0, true>), dim3(B), dim3(256), 0, s, pts, N, out, 0, dbg); break;
    default: hipLaunchKernelGGL(lrf_global_kernel, dim3(B), dim3(256), 0, s, pts, N, out, 0); break;  // geom.hip's own, this TU
  }
}
