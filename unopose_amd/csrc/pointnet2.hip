// PointNet2 set-abstraction operators for gfx950 (MI355X), C ABI part 1.
//
// Replaces the reference's CUDA extension core/unopose/model/pointnet2/_ext_src
// (sampling_gpu.cu, ball_query_gpu.cu, group_points_gpu.cu, interpolate_gpu.cu).
// These are written for CDNA4's 64-wide wavefronts from scratch:
//   * FPS: one 512-thread workgroup per cloud, points and the running
//     min-distance resident in VGPRs, wave reduction over DPP row_shr/row_bcast
//     on a packed (distance, tie-key) u64, ONE s_barrier per iteration.
//   * ball_query: one wavefront per centre, 64 candidates per step from an SoA
//     LDS tile, v_cmp ballot + mbcnt prefix for order-preserving compaction,
//     coalesced index writes.
//   * group_points: cloud staged in LDS, int4 index loads, float4 streaming
//     stores (HBM-write-bound).
// Built with -ffp-contract=off: fp32 distance expressions are evaluated exactly
// as written (no FMA), so index outputs are bit-identical to oracle/.
#include "common.h"

namespace unopose {

// ============================================================ FPS ==========
// Reference semantics (sampling_gpu.cu:74-178): idx[0]=0, temp=1e10, per
// iteration temp[k]=min(d(k,old),temp[k]); winner = argmax temp with the tie
// order of the reference's shared-memory tree: min (bitrev(k mod bs), k).
constexpr int FPS_T = 512;

template <int PPT, bool LDS_XYZ>
__global__ __launch_bounds__(FPS_T) void fps_kernel(const float *__restrict__ xyz, int n, int m,
                                                    int bs_log2, int32_t *__restrict__ idxs) {
  extern __shared__ float4 smem4[];
  uint64_t *slots = reinterpret_cast<uint64_t *>(smem4);  // [2][16]
  float *sxyz = reinterpret_cast<float *>(smem4) + 64;      // n*3 floats (if LDS_XYZ)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const float *pts = xyz + (size_t)blockIdx.x * n * 3;
  int32_t *out = idxs + (size_t)blockIdx.x * m;

  float px[PPT], py[PPT], pz[PPT], pt[PPT];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int k = i * FPS_T + tid;
    if (k < n) {
      px[i] = pts[k * 3 + 0];
      py[i] = pts[k * 3 + 1];
      pz[i] = pts[k * 3 + 2];
      pt[i] = 1e10f;
    } else {
      px[i] = py[i] = pz[i] = 0.f;
      pt[i] = -2.f;  // never beats best = -1
    }
  }
  if (LDS_XYZ) {
    for (int e = tid; e < n * 3; e += FPS_T) sxyz[e] = pts[e];
  }
  // tie key of this thread's points: bitrev_bs(k mod bs) is the same for all of
  // them because bs divides FPS_T; k / bs grows with the slot index.
  const uint32_t kmod = (uint32_t)tid & ((1u << bs_log2) - 1u);
  const uint32_t brev = bs_log2 ? (__brev(kmod) >> (32 - bs_log2)) : 0u;
  const uint32_t key_hi = brev << 23;

  if (tid == 0) out[0] = 0;
  float x1 = pts[0], y1 = pts[1], z1 = pts[2];
  __syncthreads();

  for (int j = 1; j < m; ++j) {
    float best = -1.f;
    int besti = 0;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const float dx = px[i] - x1, dy = py[i] - y1, dz = pz[i] - z1;
      const float d = dx * dx + dy * dy + dz * dz;
      const float t = fminf(d, pt[i]);
      pt[i] = t;
      const bool gt = t > best;
      best = gt ? t : best;
      besti = gt ? i : besti;
    }
    const uint32_t k = (uint32_t)besti * FPS_T + (uint32_t)tid;
    const uint32_t key = key_hi | (k >> bs_log2);
    uint64_t packed = best >= 0.f ? (((uint64_t)__float_as_uint(best) << 32) | (uint32_t)~key) : 0ull;
    packed = wave_max_u64(packed);
    uint64_t *slot = slots + (j & 1) * 16;
    if (lane == 0) slot[wave] = packed;
    __syncthreads();
    uint64_t v = lane < (FPS_T / 64) ? slot[lane] : 0ull;
    v = row_max_u64(v);
    const uint32_t wlo = __builtin_amdgcn_readlane((int)(uint32_t)v, 15);
    const uint32_t wkey = ~wlo;
    const uint32_t r = wkey >> 23, q = wkey & 0x7FFFFFu;
    const uint32_t wmod = bs_log2 ? (__brev(r) >> (32 - bs_log2)) : 0u;
    const int old = (int)((q << bs_log2) | wmod);
    if (tid == 0) out[j] = old;
    if (LDS_XYZ) {
      x1 = sxyz[old * 3 + 0];
      y1 = sxyz[old * 3 + 1];
      z1 = sxyz[old * 3 + 2];
    } else {
      x1 = pts[old * 3 + 0];
      y1 = pts[old * 3 + 1];
      z1 = pts[old * 3 + 2];
    }
  }
}

template <int PPT>
static int launch_fps(const float *xyz, int B, int N, int M, int bs_log2, int32_t *idx, hipStream_t s) {
  const size_t xyz_bytes = (size_t)N * 3 * sizeof(float);
  if (xyz_bytes <= 60 * 1024) {
    hipLaunchKernelGGL((fps_kernel<PPT, true>), dim3(B), dim3(FPS_T), 256 + xyz_bytes, s, xyz, N, M, bs_log2, idx);
  } else {
    hipLaunchKernelGGL((fps_kernel<PPT, false>), dim3(B), dim3(FPS_T), 256, s, xyz, N, M, bs_log2, idx);
  }
  return check_launch("furthest_point_sampling");
}

// ===================================================== gather_points =======
__global__ __launch_bounds__(256) void gather_points_kernel(const float *__restrict__ points,
                                                            const int32_t *__restrict__ idx, int C, int N, int M,
                                                            float *__restrict__ out) {
  const int b = blockIdx.z;
  const int32_t *I = idx + (size_t)b * M;
  for (int c = blockIdx.y; c < C; c += gridDim.y) {
    const float *P = points + ((size_t)b * C + c) * N;
    float *O = out + ((size_t)b * C + c) * M;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < M; j += gridDim.x * 256) O[j] = P[I[j]];
  }
}

__global__ __launch_bounds__(256) void gather_points_grad_kernel(const float *__restrict__ grad_out,
                                                                 const int32_t *__restrict__ idx, int C, int N, int M,
                                                                 float *__restrict__ grad_points) {
  const int b = blockIdx.z;
  const int32_t *I = idx + (size_t)b * M;
  for (int c = blockIdx.y; c < C; c += gridDim.y) {
    float *G = grad_points + ((size_t)b * C + c) * N;
    const float *O = grad_out + ((size_t)b * C + c) * M;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < M; j += gridDim.x * 256) atomicAdd(G + I[j], O[j]);
  }
}

// ======================================================== ball_query =======
// Reference (ball_query_gpu.cu:14-49): for centre j scan k ascending, keep the
// first `nsample` with d2 < r^2 (strict), pad the tail with the first hit;
// rows without a hit are all zero.
constexpr int BQ_CHUNK = 2048;  // points per LDS tile (SoA, 24 KB)

template <int CPW>
__global__ __launch_bounds__(256) void ball_query_kernel(const float *__restrict__ new_xyz,
                                                         const float *__restrict__ xyz, int N, int M, float radius2,
                                                         int nsample, int32_t *__restrict__ idx) {
  __shared__ float sx[BQ_CHUNK], sy[BQ_CHUNK], sz[BQ_CHUNK];
  const int b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float *P = xyz + (size_t)b * N * 3;
  const float *Q = new_xyz + (size_t)b * M * 3;
  const int c0 = (blockIdx.x * 4 + wave) * CPW;

  float cx[CPW], cy[CPW], cz[CPW];
  int cnt[CPW], first[CPW];
#pragma unroll
  for (int i = 0; i < CPW; ++i) {
    const int c = c0 + i;
    const bool ok = c < M;
    cx[i] = ok ? Q[c * 3 + 0] : 0.f;
    cy[i] = ok ? Q[c * 3 + 1] : 0.f;
    cz[i] = ok ? Q[c * 3 + 2] : 0.f;
    cnt[i] = ok ? 0 : nsample;  // out-of-range centres are "full": never scanned
    first[i] = 0;
  }

  for (int base = 0; base < N; base += BQ_CHUNK) {
    const int len = min(BQ_CHUNK, N - base);
    __syncthreads();
    for (int e = tid; e < len * 3; e += 256) {
      const float v = P[(size_t)base * 3 + e];
      const int p = e / 3, comp = e - p * 3;
      (comp == 0 ? sx : comp == 1 ? sy : sz)[p] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
      if (cnt[i] >= nsample) continue;  // wave-uniform
      int32_t *row = idx + ((size_t)b * M + (c0 + i)) * nsample;
      for (int k0 = 0; k0 < len; k0 += 64) {
        const int k = k0 + lane;
        bool hit = false;
        if (k < len) {
          const float x = sx[k], y = sy[k], z = sz[k];
          const float d2 = (cx[i] - x) * (cx[i] - x) + (cy[i] - y) * (cy[i] - y) + (cz[i] - z) * (cz[i] - z);
          hit = d2 < radius2;
        }
        const unsigned long long mask = __ballot(hit);
        if (mask) {
          const int pre = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                         __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
          const int pos = cnt[i] + pre;
          if (hit && pos < nsample) row[pos] = base + k;
          if (cnt[i] == 0) first[i] = base + k0 + __builtin_ctzll(mask);
          cnt[i] += __builtin_popcountll(mask);
          if (cnt[i] >= nsample) break;
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < CPW; ++i) {
    const int c = c0 + i;
    if (c >= M) continue;
    int32_t *row = idx + ((size_t)b * M + c) * nsample;
    const int fill = cnt[i] > 0 ? first[i] : 0;
    for (int l = min(cnt[i], nsample) + lane; l < nsample; l += 64) row[l] = fill;
  }
}

// ====================================================== group_points =======
// out[b,c,j,k] = points[b,c,idx[b,j,k]]; MS = M*S flattened.  LDS variant: the
// whole (C,N) slab of one cloud is staged once per block.
__global__ __launch_bounds__(256) void group_points_lds_kernel(const float *__restrict__ points,
                                                               const int32_t *__restrict__ idx, int C, int N,
                                                               int MS4, int per_block4, float *__restrict__ out) {
  extern __shared__ float4 smem4[];
  float *sp = reinterpret_cast<float *>(smem4);
  const int b = blockIdx.y, tid = threadIdx.x;
  const float *P = points + (size_t)b * C * N;
  const int CN = C * N;
  if ((CN & 3) == 0 && (((uintptr_t)P) & 15) == 0) {
    const float4 *P4 = reinterpret_cast<const float4 *>(P);
    for (int e = tid; e < CN / 4; e += 256) smem4[e] = P4[e];
  } else {
    for (int e = tid; e < CN; e += 256) sp[e] = P[e];
  }
  __syncthreads();
  const int4 *I4 = reinterpret_cast<const int4 *>(idx) + (size_t)b * MS4;
  f32x4 *O4 = reinterpret_cast<f32x4 *>(out) + (size_t)b * C * MS4;
  const int e_begin = blockIdx.x * per_block4;
  const int e_end = min(MS4, e_begin + per_block4);
  for (int e = e_begin + tid; e < e_end; e += 256) {
    const int4 ii = I4[e];
    for (int c = 0; c < C; ++c) {
      const float *s = sp + c * N;
      f32x4 v = {s[ii.x], s[ii.y], s[ii.z], s[ii.w]};
      __builtin_nontemporal_store(v, O4 + (size_t)c * MS4 + e);
    }
  }
}

// generic variant (any C, N, MS): gather straight from global (L2-resident).
__global__ __launch_bounds__(256) void group_points_kernel(const float *__restrict__ points,
                                                           const int32_t *__restrict__ idx, int C, int N, long MS,
                                                           float *__restrict__ out) {
  const int b = blockIdx.z;
  const int32_t *I = idx + (size_t)b * MS;
  for (int c = blockIdx.y; c < C; c += gridDim.y) {
    const float *P = points + ((size_t)b * C + c) * N;
    float *O = out + ((size_t)b * C + c) * MS;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < MS; e += (long)gridDim.x * 256) O[e] = P[I[e]];
  }
}

__global__ __launch_bounds__(256) void group_points_grad_kernel(const float *__restrict__ grad_out,
                                                                const int32_t *__restrict__ idx, int C, int N, long MS,
                                                                float *__restrict__ grad_points) {
  const int b = blockIdx.z;
  const int32_t *I = idx + (size_t)b * MS;
  for (int c = blockIdx.y; c < C; c += gridDim.y) {
    float *G = grad_points + ((size_t)b * C + c) * N;
    const float *O = grad_out + ((size_t)b * C + c) * MS;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < MS; e += (long)gridDim.x * 256) atomicAdd(G + I[e], O[e]);
  }
}

// ================================================== three_nn / interp ======
// ABI completeness (never reached by UNOPose.forward).
__global__ __launch_bounds__(256) void three_nn_kernel(const float *__restrict__ unknown,
                                                       const float *__restrict__ known, int n, int m,
                                                       float *__restrict__ dist2, int32_t *__restrict__ idx) {
  const int b = blockIdx.y;
  const float *U = unknown + (size_t)b * n * 3;
  const float *K = known + (size_t)b * m * 3;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const float ux = U[j * 3 + 0], uy = U[j * 3 + 1], uz = U[j * 3 + 2];
  double best1 = 1e40, best2 = 1e40, best3 = 1e40;
  int besti1 = 0, besti2 = 0, besti3 = 0;
  for (int k = 0; k < m; ++k) {
    const float x = K[k * 3 + 0], y = K[k * 3 + 1], z = K[k * 3 + 2];
    const float d = (ux - x) * (ux - x) + (uy - y) * (uy - y) + (uz - z) * (uz - z);
    if (d < best1) {
      best3 = best2; besti3 = besti2;
      best2 = best1; besti2 = besti1;
      best1 = d; besti1 = k;
    } else if (d < best2) {
      best3 = best2; besti3 = besti2;
      best2 = d; besti2 = k;
    } else if (d < best3) {
      best3 = d; besti3 = k;
    }
  }
  float *D = dist2 + ((size_t)b * n + j) * 3;
  int32_t *I = idx + ((size_t)b * n + j) * 3;
  D[0] = (float)best1; D[1] = (float)best2; D[2] = (float)best3;
  I[0] = besti1; I[1] = besti2; I[2] = besti3;
}

__global__ __launch_bounds__(256) void three_interpolate_kernel(const float *__restrict__ points,
                                                                const int32_t *__restrict__ idx,
                                                                const float *__restrict__ weight, int c, int m, int n,
                                                                float *__restrict__ out) {
  const int b = blockIdx.z;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const float *W = weight + ((size_t)b * n + j) * 3;
  const int32_t *I = idx + ((size_t)b * n + j) * 3;
  const float w1 = W[0], w2 = W[1], w3 = W[2];
  const int i1 = I[0], i2 = I[1], i3 = I[2];
  for (int l = blockIdx.y; l < c; l += gridDim.y) {
    const float *P = points + ((size_t)b * c + l) * m;
    out[((size_t)b * c + l) * n + j] = P[i1] * w1 + P[i2] * w2 + P[i3] * w3;
  }
}

__global__ __launch_bounds__(256) void three_interpolate_grad_kernel(const float *__restrict__ grad_out,
                                                                     const int32_t *__restrict__ idx,
                                                                     const float *__restrict__ weight, int c, int n,
                                                                     int m, float *__restrict__ grad_points) {
  const int b = blockIdx.z;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const float *W = weight + ((size_t)b * n + j) * 3;
  const int32_t *I = idx + ((size_t)b * n + j) * 3;
  const float w1 = W[0], w2 = W[1], w3 = W[2];
  const int i1 = I[0], i2 = I[1], i3 = I[2];
  for (int l = blockIdx.y; l < c; l += gridDim.y) {
    float *G = grad_points + ((size_t)b * c + l) * m;
    const float g = grad_out[((size_t)b * c + l) * n + j];
    atomicAdd(G + i1, g * w1);
    atomicAdd(G + i2, g * w2);
    atomicAdd(G + i3, g * w3);
  }
}

}  // namespace unopose

using namespace unopose;

extern "C" {

int unopose_furthest_point_sampling(const float *xyz, int B, int N, int M, int32_t *idx, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(xyz && idx, "furthest_point_sampling: null pointer");
  UNOPOSE_REQUIRE(B >= 0 && N >= 1 && M >= 0, "furthest_point_sampling: bad sizes B=%d N=%d M=%d", B, N, M);
  if (B == 0 || M == 0) return UNOPOSE_OK;
  UNOPOSE_REQUIRE(N <= FPS_T * 32, "furthest_point_sampling: N=%d exceeds the register-resident limit %d", N,
                  FPS_T * 32);
  hipStream_t s = (hipStream_t)stream;
  int bs_log2 = 0;  // bs = min(512, 2^floor(log2 N))  (cuda_utils.h:18-24)
  while ((2 << bs_log2) <= N && bs_log2 < 9) ++bs_log2;
  const int ppt = cdiv(N, FPS_T);
#define UNOPOSE_FPS_CASE(P) \
  if (ppt <= P) return launch_fps<P>(xyz, B, N, M, bs_log2, idx, s)
  UNOPOSE_FPS_CASE(1);
  UNOPOSE_FPS_CASE(2);
  UNOPOSE_FPS_CASE(4);
  UNOPOSE_FPS_CASE(6);
  UNOPOSE_FPS_CASE(8);
  UNOPOSE_FPS_CASE(10);
  UNOPOSE_FPS_CASE(12);
  UNOPOSE_FPS_CASE(16);
  UNOPOSE_FPS_CASE(24);
  UNOPOSE_FPS_CASE(32);
#undef UNOPOSE_FPS_CASE
  return UNOPOSE_EINVAL;
}

int unopose_gather_points(const float *points, const int32_t *idx, int B, int C, int N, int M, float *out,
                          unopose_stream_t stream) {
  UNOPOSE_REQUIRE(points && idx && out, "gather_points: null pointer");
  UNOPOSE_REQUIRE(B >= 0 && C >= 0 && N >= 1 && M >= 0, "gather_points: bad sizes");
  if (B == 0 || C == 0 || M == 0) return UNOPOSE_OK;
  dim3 grid(cdiv(M, 256), min(C, 1024), B);
  UNOPOSE_REQUIRE(B <= 65535, "gather_points: B too large");
  hipLaunchKernelGGL(gather_points_kernel, grid, dim3(256), 0, (hipStream_t)stream, points, idx, C, N, M, out);
  return check_launch("gather_points");
}

int unopose_gather_points_grad(const float *grad_out, const int32_t *idx, int B, int C, int N, int M,
                               float *grad_points, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(grad_out && idx && grad_points, "gather_points_grad: null pointer");
  UNOPOSE_REQUIRE(B >= 0 && C >= 0 && N >= 1 && M >= 0 && B <= 65535, "gather_points_grad: bad sizes");
  if (B == 0 || C == 0 || M == 0) return UNOPOSE_OK;
  dim3 grid(cdiv(M, 256), min(C, 1024), B);
  hipLaunchKernelGGL(gather_points_grad_kernel, grid, dim3(256), 0, (hipStream_t)stream, grad_out, idx, C, N, M,
                     grad_points);
  return check_launch("gather_points_grad");
}

int unopose_ball_query(const float *new_xyz, const float *xyz, int B, int N, int M, float radius, int nsample,
                       int32_t *idx, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(new_xyz && xyz && idx, "ball_query: null pointer");
  UNOPOSE_REQUIRE(B >= 0 && N >= 0 && M >= 0 && nsample >= 0 && B <= 65535, "ball_query: bad sizes");
  if (B == 0 || M == 0 || nsample == 0) return UNOPOSE_OK;
  const float r2 = radius * radius;  // fp32 product, as the reference kernel computes it
  hipStream_t s = (hipStream_t)stream;
  // enough blocks to fill 256 CUs, but as many centres per wave as that allows
  const long centres = (long)B * M;
  if (centres >= 8L * 4 * 2048) {
    hipLaunchKernelGGL(ball_query_kernel<8>, dim3(cdiv(M, 32), B), dim3(256), 0, s, new_xyz, xyz, N, M, r2, nsample,
                       idx);
  } else if (centres >= 4L * 4 * 2048) {
    hipLaunchKernelGGL(ball_query_kernel<4>, dim3(cdiv(M, 16), B), dim3(256), 0, s, new_xyz, xyz, N, M, r2, nsample,
                       idx);
  } else if (centres >= 2L * 4 * 1024) {
    hipLaunchKernelGGL(ball_query_kernel<2>, dim3(cdiv(M, 8), B), dim3(256), 0, s, new_xyz, xyz, N, M, r2, nsample,
                       idx);
  } else {
    hipLaunchKernelGGL(ball_query_kernel<1>, dim3(cdiv(M, 4), B), dim3(256), 0, s, new_xyz, xyz, N, M, r2, nsample,
                       idx);
  }
  return check_launch("ball_query");
}

int unopose_group_points(const float *points, const int32_t *idx, int B, int C, int N, int M, int S, float *out,
                         unopose_stream_t stream) {
  UNOPOSE_REQUIRE(points && idx && out, "group_points: null pointer");
  UNOPOSE_REQUIRE(B >= 0 && C >= 0 && N >= 1 && M >= 0 && S >= 0 && B <= 65535, "group_points: bad sizes");
  const long MS = (long)M * S;
  if (B == 0 || C == 0 || MS == 0) return UNOPOSE_OK;
  hipStream_t s = (hipStream_t)stream;
  const size_t slab = (size_t)C * N * sizeof(float);
  const bool vec_ok = (MS % 4 == 0) && ((uintptr_t)idx % 16 == 0) && ((uintptr_t)out % 16 == 0) &&
                      MS / 4 < (1L << 30);
  if (slab <= 64 * 1024 && vec_ok) {
    const int MS4 = (int)(MS / 4);
    // ~16K output elements per channel per block; at least 1 block
    const int per_block4 = 4096;
    dim3 grid(cdiv(MS4, per_block4), B);
    hipLaunchKernelGGL(group_points_lds_kernel, grid, dim3(256), slab, s, points, idx, C, N, MS4, per_block4, out);
  } else {
    dim3 grid((unsigned)min((long)cdiv(MS, 256), 4096L), min(C, 256), B);
    hipLaunchKernelGGL(group_points_kernel, grid, dim3(256), 0, s, points, idx, C, N, MS, out);
  }
  return check_launch("group_points");
}

int unopose_group_points_grad(const float *grad_out, const int32_t *idx, int B, int C, int N, int M, int S,
                              float *grad_points, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(grad_out && idx && grad_points, "group_points_grad: null pointer");
  UNOPOSE_REQUIRE(B >= 0 && C >= 0 && N >= 1 && M >= 0 && S >= 0 && B <= 65535, "group_points_grad: bad sizes");
  const long MS = (long)M * S;
  if (B == 0 || C == 0 || MS == 0) return UNOPOSE_OK;
  dim3 grid((unsigned)min((long)cdiv(MS, 256), 4096L), min(C, 256), B);
  hipLaunchKernelGGL(group_points_grad_kernel, grid, dim3(256), 0, (hipStream_t)stream, grad_out, idx, C, N, MS,
                     grad_points);
  return check_launch("group_points_grad");
}

int unopose_three_nn(const float *unknown, const float *known, int B, int n, int m, float *dist2, int32_t *idx,
                     unopose_stream_t stream) {
  UNOPOSE_REQUIRE(unknown && known && dist2 && idx, "three_nn: null pointer");
  UNOPOSE_REQUIRE(B >= 0 && n >= 0 && m >= 0 && B <= 65535, "three_nn: bad sizes");
  if (B == 0 || n == 0) return UNOPOSE_OK;
  hipLaunchKernelGGL(three_nn_kernel, dim3(cdiv(n, 256), B), dim3(256), 0, (hipStream_t)stream, unknown, known, n, m,
                     dist2, idx);
  return check_launch("three_nn");
}

int unopose_three_interpolate(const float *points, const int32_t *idx, const float *weight, int B, int c, int m, int n,
                              float *out, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(points && idx && weight && out, "three_interpolate: null pointer");
  UNOPOSE_REQUIRE(B >= 0 && c >= 0 && m >= 1 && n >= 0 && B <= 65535, "three_interpolate: bad sizes");
  if (B == 0 || c == 0 || n == 0) return UNOPOSE_OK;
  hipLaunchKernelGGL(three_interpolate_kernel, dim3(cdiv(n, 256), min(c, 1024), B), dim3(256), 0, (hipStream_t)stream,
                     points, idx, weight, c, m, n, out);
  return check_launch("three_interpolate");
}

int unopose_three_interpolate_grad(const float *grad_out, const int32_t *idx, const float *weight, int B, int c, int n,
                                   int m, float *grad_points, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(grad_out && idx && weight && grad_points, "three_interpolate_grad: null pointer");
  UNOPOSE_REQUIRE(B >= 0 && c >= 0 && m >= 1 && n >= 0 && B <= 65535, "three_interpolate_grad: bad sizes");
  if (B == 0 || c == 0 || n == 0) return UNOPOSE_OK;
  hipLaunchKernelGGL(three_interpolate_grad_kernel, dim3(cdiv(n, 256), min(c, 1024), B), dim3(256), 0,
                     (hipStream_t)stream, grad_out, idx, weight, c, n, m, grad_points);
  return check_launch("three_interpolate_grad");
}

}  // extern "C"
