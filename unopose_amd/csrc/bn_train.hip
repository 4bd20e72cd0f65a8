// Training-mode BatchNorm2d + ReLU of the positional encoding's SharedMLP (C ABI part 3: training path).
//
// Replaces, per layer of core/unopose/model/pointnet2/pytorch_utils.py:25-132 under train(): nn.BatchNorm2d with batch statistics
// (MIOpenBatchNormFwdTrainSpatial / BwdSpatial: 73.6 ms of the 254 ms step at BASELINE configs[3], ~2 TB/s on tensors of up to
// 4.3 GB) and the separate ReLU forward / threshold-backward passes.  The tensors are (B, C, N, S) fp32, channel-major slabs of
// L = N * S contiguous values, so everything is streaming work:
//   forward   stats pass (sum, sum of squares per (b, c) chunk -> fp32 partials, combined in double per channel, running statistics
//             updated exactly as nn.BatchNorm2d does: biased variance for the normalisation, unbiased for running_var) + apply pass
//             y = max(x a + b, 0),  a = gamma rstd,  b = beta - mean a;
//   backward  reduction pass (sum dz, sum dz xhat with dz = dy [x a + b > 0], xhat = (x - mean) rstd: the ReLU mask is RECOMPUTED
//             from x, nothing but x, mean, rstd is kept from the forward) + apply pass dx = a (dz - dbeta / M - xhat dgamma / M).
// Every pass moves 16 bytes per lane per access; a workgroup owns one chunk of one (b, c) slab, so scale and shift are scalars.
#include "common.h"

namespace unopose {

constexpr int BN_CHUNK = 16384;  // elements of a slab per workgroup (256 threads x 16 float4)

__device__ __forceinline__ float2 block_sum2(float a, float b, float2 *red) {
  a = wave_sum_f32(a);
  b = wave_sum_f32(b);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) red[wave] = make_float2(a, b);
  __syncthreads();
  float2 r = red[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) {
    r.x += red[w].x;
    r.y += red[w].y;
  }
  return r;
}

// part[((c * B + b) * nchunk + chunk)] = (sum x, sum x^2) of that chunk
__global__ __launch_bounds__(256) void bn_stats_kernel(const float *__restrict__ x, int C, long L, int nchunk, float2 *__restrict__ part) {
  __shared__ float2 red[4];
  const int chunk = blockIdx.x, c = blockIdx.y, b = blockIdx.z, B = gridDim.z;
  const float *p = x + ((size_t)b * C + c) * L;
  const long e0 = (long)chunk * BN_CHUNK, e1 = min(e0 + BN_CHUNK, L);
  float s = 0.f, ss = 0.f;
  for (long e = e0 + threadIdx.x * 4; e < e1; e += 256 * 4) {
    const float4 v = *reinterpret_cast<const float4 *>(p + e);
    s += (v.x + v.y) + (v.z + v.w);
    ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  const float2 r = block_sum2(s, ss, red);
  if (threadIdx.x == 0) part[((size_t)c * B + b) * nchunk + chunk] = r;
}

// one workgroup per channel: mean / rstd of the batch, running statistics (momentum < 0: cumulative average with `count` batches seen)
__global__ __launch_bounds__(64) void bn_finalize_kernel(const float2 *__restrict__ part, int per_channel, double count, float eps,
                                                         float momentum, float *__restrict__ mean_out, float *__restrict__ rstd_out,
                                                         float *__restrict__ running_mean, float *__restrict__ running_var) {
  const int c = blockIdx.x;
  double s = 0.0, ss = 0.0;
  for (int i = threadIdx.x; i < per_channel; i += 64) {
    const float2 v = part[(size_t)c * per_channel + i];
    s += (double)v.x;
    ss += (double)v.y;
  }
  for (int d = 32; d >= 1; d >>= 1) {
    s += __shfl_xor(s, d);
    ss += __shfl_xor(ss, d);
  }
  if (threadIdx.x == 0) {
    const double mean = s / count;
    const double var = fmax(ss / count - mean * mean, 0.0);  // biased: what normalises the batch
    mean_out[c] = (float)mean;
    rstd_out[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
      const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
      running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unbiased);
    }
  }
}

__global__ __launch_bounds__(256) void bn_relu_apply_kernel(const float *__restrict__ x, int C, long L, const float *__restrict__ mean,
                                                            const float *__restrict__ rstd, const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, float *__restrict__ y) {
  const int chunk = blockIdx.x, c = blockIdx.y, b = blockIdx.z;
  const size_t base = ((size_t)b * C + c) * L;
  const float a = gamma[c] * rstd[c], sh = beta[c] - mean[c] * a;
  const long e0 = (long)chunk * BN_CHUNK, e1 = min(e0 + BN_CHUNK, L);
  for (long e = e0 + threadIdx.x * 4; e < e1; e += 256 * 4) {
    const float4 v = *reinterpret_cast<const float4 *>(x + base + e);
    float4 o;
    o.x = fmaxf(fmaf(v.x, a, sh), 0.f);
    o.y = fmaxf(fmaf(v.y, a, sh), 0.f);
    o.z = fmaxf(fmaf(v.z, a, sh), 0.f);
    o.w = fmaxf(fmaf(v.w, a, sh), 0.f);
    *reinterpret_cast<float4 *>(y + base + e) = o;
  }
}

// part = (sum dz, sum dz xhat) per chunk
__global__ __launch_bounds__(256) void bn_relu_bwd_stats_kernel(const float *__restrict__ x, const float *__restrict__ dy, int C, long L, int nchunk,
                                                                const float *__restrict__ mean, const float *__restrict__ rstd,
                                                                const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                float2 *__restrict__ part) {
  __shared__ float2 red[4];
  const int chunk = blockIdx.x, c = blockIdx.y, b = blockIdx.z, B = gridDim.z;
  const size_t base = ((size_t)b * C + c) * L;
  const float m = mean[c], r = rstd[c], a = gamma[c] * r, sh = beta[c] - m * a;
  const long e0 = (long)chunk * BN_CHUNK, e1 = min(e0 + BN_CHUNK, L);
  float s = 0.f, sx = 0.f;
  for (long e = e0 + threadIdx.x * 4; e < e1; e += 256 * 4) {
    const float4 v = *reinterpret_cast<const float4 *>(x + base + e), g = *reinterpret_cast<const float4 *>(dy + base + e);
    const float xs[4] = {v.x, v.y, v.z, v.w}, gs[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float dz = fmaf(xs[i], a, sh) > 0.f ? gs[i] : 0.f;
      s += dz;
      sx += dz * ((xs[i] - m) * r);
    }
  }
  const float2 t = block_sum2(s, sx, red);
  if (threadIdx.x == 0) part[((size_t)c * B + b) * nchunk + chunk] = t;
}

// dbeta[c], dgamma[c] from the partials (double accumulation)
__global__ __launch_bounds__(64) void bn_bwd_finalize_kernel(const float2 *__restrict__ part, int per_channel, float *__restrict__ dgamma,
                                                             float *__restrict__ dbeta) {
  const int c = blockIdx.x;
  double s = 0.0, sx = 0.0;
  for (int i = threadIdx.x; i < per_channel; i += 64) {
    const float2 v = part[(size_t)c * per_channel + i];
    s += (double)v.x;
    sx += (double)v.y;
  }
  for (int d = 32; d >= 1; d >>= 1) {
    s += __shfl_xor(s, d);
    sx += __shfl_xor(sx, d);
  }
  if (threadIdx.x == 0) {
    dbeta[c] = (float)s;
    dgamma[c] = (float)sx;
  }
}

__global__ __launch_bounds__(256) void bn_relu_bwd_apply_kernel(const float *__restrict__ x, const float *__restrict__ dy, int C, long L,
                                                                const float *__restrict__ mean, const float *__restrict__ rstd,
                                                                const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                const float *__restrict__ dgamma, const float *__restrict__ dbeta, float inv_count,
                                                                float *__restrict__ dx) {
  const int chunk = blockIdx.x, c = blockIdx.y, b = blockIdx.z;
  const size_t base = ((size_t)b * C + c) * L;
  const float m = mean[c], r = rstd[c], a = gamma[c] * r, sh = beta[c] - m * a;
  const float k0 = dbeta[c] * inv_count, k1 = dgamma[c] * inv_count;
  const long e0 = (long)chunk * BN_CHUNK, e1 = min(e0 + BN_CHUNK, L);
  for (long e = e0 + threadIdx.x * 4; e < e1; e += 256 * 4) {
    const float4 v = *reinterpret_cast<const float4 *>(x + base + e), g = *reinterpret_cast<const float4 *>(dy + base + e);
    const float xs[4] = {v.x, v.y, v.z, v.w}, gs[4] = {g.x, g.y, g.z, g.w};
    float o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float dz = fmaf(xs[i], a, sh) > 0.f ? gs[i] : 0.f;
      o[i] = a * (dz - k0 - ((xs[i] - m) * r) * k1);
    }
    *reinterpret_cast<float4 *>(dx + base + e) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// ---- the LAST layer of the SharedMLP: relu(batch_norm(x)) followed by the max over the S neighbours (fine matcher PE,
// oneref_predator_fine_point_matching.py:167-174 in train()).  Fused, the 128-channel activation (10.7 GB at configs[3]) is never written:
// forward reads x once more after the statistics pass and keeps (max, argmax) per (b, c, n); backward gets the pooled gradient, so the
// two channel sums run over B N gathered elements instead of B N S, and the apply pass reads x and writes dx (no dense dy, no zero fill,
// no scatter).  A row (b, c, n) is S contiguous floats, S a power of two in [32, 256]: 256 / S rows per wave access of 1 KiB.
template <int S>
__global__ __launch_bounds__(256) void bn_relu_maxpool_kernel(const float *__restrict__ x, int C, long rows, const float *__restrict__ mean,
                                                              const float *__restrict__ rstd, const float *__restrict__ gamma,
                                                              const float *__restrict__ beta, int N, float *__restrict__ out,
                                                              int32_t *__restrict__ idx) {
  constexpr int LPR = S / 4, RPW = 64 / LPR;  // lanes per row, rows per wave access
  const int lane = threadIdx.x & 63, sub = lane / LPR, li = lane % LPR;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwave = (long)gridDim.x * 4;
  for (long r0 = wave * RPW; r0 < rows; r0 += nwave * RPW) {
    const long row = r0 + sub;  // (b * C + c) * N + n
    float best = -1.f;
    int bi = 0;
    if (row < rows) {
      const int c = (int)((row / N) % C);
      const float a = gamma[c] * rstd[c], sh = beta[c] - mean[c] * a;
      const float4 v = *reinterpret_cast<const float4 *>(x + row * S + 4 * li);
      const float xs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float y = fmaxf(fmaf(xs[i], a, sh), 0.f);
        if (y > best) best = y, bi = 4 * li + i;
      }
    }
#pragma unroll
    for (int d = 1; d < LPR; d <<= 1) {  // first maximum wins: smaller index on ties
      const float ob = __shfl_xor(best, d);
      const int oi = __shfl_xor(bi, d);
      if (ob > best || (ob == best && oi < bi)) best = ob, bi = oi;
    }
    if (li == 0 && row < rows) {
      out[row] = best;
      idx[row] = bi;
    }
  }
}

// part[c * B + b] = (sum dz, sum dz xhat) over the N pooled rows of slab (b, c): dz = g where the winner's pre-activation is positive
__global__ __launch_bounds__(256) void bn_relu_maxpool_bwd_stats_kernel(const float *__restrict__ x, const float *__restrict__ g,
                                                                        const int32_t *__restrict__ idx, int C, int N, int S,
                                                                        const float *__restrict__ mean, const float *__restrict__ rstd,
                                                                        const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                        float2 *__restrict__ part) {
  __shared__ float2 red[4];
  const int c = blockIdx.x, b = blockIdx.y, B = gridDim.y;
  const size_t slab = (size_t)b * C + c;
  const float m = mean[c], r = rstd[c], a = gamma[c] * r, sh = beta[c] - m * a;
  float s = 0.f, sx = 0.f;
  for (int n = threadIdx.x; n < N; n += 256) {
    const float xv = x[(slab * N + n) * S + idx[slab * N + n]];
    const float dz = fmaf(xv, a, sh) > 0.f ? g[slab * N + n] : 0.f;
    s += dz;
    sx += dz * ((xv - m) * r);
  }
  const float2 t = block_sum2(s, sx, red);
  if (threadIdx.x == 0) part[(size_t)c * B + b] = t;
}

template <int S>
__global__ __launch_bounds__(256) void bn_relu_maxpool_bwd_apply_kernel(const float *__restrict__ x, const float *__restrict__ g,
                                                                        const int32_t *__restrict__ idx, int C, long rows, int N,
                                                                        const float *__restrict__ mean, const float *__restrict__ rstd,
                                                                        const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                        const float *__restrict__ dgamma, const float *__restrict__ dbeta,
                                                                        float inv_count, float *__restrict__ dx) {
  constexpr int LPR = S / 4, RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, sub = lane / LPR, li = lane % LPR;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwave = (long)gridDim.x * 4;
  for (long r0 = wave * RPW; r0 < rows; r0 += nwave * RPW) {
    const long row = r0 + sub;
    if (row >= rows) continue;
    const int c = (int)((row / N) % C);
    const float m = mean[c], r = rstd[c], a = gamma[c] * r, sh = beta[c] - m * a;
    const float k0 = dbeta[c] * inv_count, k1 = dgamma[c] * inv_count;
    const int win = idx[row] - 4 * li;  // position of the winner inside this lane's four elements (or outside 0..3)
    const float gv = g[row];
    const float4 v = *reinterpret_cast<const float4 *>(x + row * S + 4 * li);
    const float xs[4] = {v.x, v.y, v.z, v.w};
    float o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float dz = (i == win && fmaf(xs[i], a, sh) > 0.f) ? gv : 0.f;
      o[i] = a * (dz - k0 - ((xs[i] - m) * r) * k1);
    }
    *reinterpret_cast<float4 *>(dx + row * S + 4 * li) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

}  // namespace unopose

using namespace unopose;

extern "C" {

int unopose_bn_train_chunk(void) { return BN_CHUNK; }  // workspace = 2 * B * C * ceil(L / chunk) floats

int unopose_bn_relu_train_forward(const float *x, int B, int C, long L, const float *gamma, const float *beta, float eps, float momentum,
                                  float *running_mean, float *running_var, float *workspace, float *mean, float *rstd, float *y,
                                  unopose_stream_t stream) {
  UNOPOSE_REQUIRE(x && gamma && beta && workspace && mean && rstd && y, "bn_relu_train_forward: null pointer");
  UNOPOSE_REQUIRE(B >= 1 && B <= 65535 && C >= 1 && C <= 65535 && L >= 4 && L % 4 == 0 && (running_mean == nullptr) == (running_var == nullptr),
                  "bn_relu_train_forward: bad sizes (slab length must be a multiple of 4; got B=%d C=%d L=%ld)", B, C, L);
  hipStream_t s = (hipStream_t)stream;
  const int nchunk = cdiv(L, BN_CHUNK);
  dim3 grid(nchunk, C, B);
  hipLaunchKernelGGL(bn_stats_kernel, grid, dim3(256), 0, s, x, C, L, nchunk, (float2 *)workspace);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(64), 0, s, (const float2 *)workspace, B * nchunk, (double)B * (double)L, eps, momentum, mean,
                     rstd, running_mean, running_var);
  hipLaunchKernelGGL(bn_relu_apply_kernel, grid, dim3(256), 0, s, x, C, L, (const float *)mean, (const float *)rstd, gamma, beta, y);
  return check_launch("bn_relu_train_forward");
}

int unopose_bn_relu_train_backward(const float *x, const float *dy, int B, int C, long L, const float *gamma, const float *beta,
                                   const float *mean, const float *rstd, float *workspace, float *dgamma, float *dbeta, float *dx,
                                   unopose_stream_t stream) {
  UNOPOSE_REQUIRE(x && dy && gamma && beta && mean && rstd && workspace && dgamma && dbeta && dx, "bn_relu_train_backward: null pointer");
  UNOPOSE_REQUIRE(B >= 1 && B <= 65535 && C >= 1 && C <= 65535 && L >= 4 && L % 4 == 0, "bn_relu_train_backward: bad sizes (B=%d C=%d L=%ld)", B, C, L);
  hipStream_t s = (hipStream_t)stream;
  const int nchunk = cdiv(L, BN_CHUNK);
  dim3 grid(nchunk, C, B);
  hipLaunchKernelGGL(bn_relu_bwd_stats_kernel, grid, dim3(256), 0, s, x, dy, C, L, nchunk, mean, rstd, gamma, beta, (float2 *)workspace);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(64), 0, s, (const float2 *)workspace, B * nchunk, dgamma, dbeta);
  hipLaunchKernelGGL(bn_relu_bwd_apply_kernel, grid, dim3(256), 0, s, x, dy, C, L, mean, rstd, gamma, beta, (const float *)dgamma,
                     (const float *)dbeta, (float)(1.0 / ((double)B * (double)L)), dx);
  return check_launch("bn_relu_train_backward");
}

static int pool_grid(long rows, int rpw) {
  const long want = (rows + 4L * rpw - 1) / (4L * rpw);
  return (int)(want < 1 ? 1 : want < 8192 ? want : 8192);
}

int unopose_bn_relu_maxpool_train_forward(const float *x, int B, int C, int N, int S, const float *gamma, const float *beta, float eps,
                                          float momentum, float *running_mean, float *running_var, float *workspace, float *mean, float *rstd,
                                          float *out, int32_t *idx, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(x && gamma && beta && workspace && mean && rstd && out && idx, "bn_relu_maxpool_train_forward: null pointer");
  UNOPOSE_REQUIRE(B >= 1 && B <= 65535 && C >= 1 && C <= 65535 && N >= 1 && (S == 32 || S == 64 || S == 128 || S == 256) &&
                      (running_mean == nullptr) == (running_var == nullptr),
                  "bn_relu_maxpool_train_forward: bad sizes (S must be 32, 64, 128 or 256; got B=%d C=%d N=%d S=%d)", B, C, N, S);
  hipStream_t s = (hipStream_t)stream;
  const long L = (long)N * S, rows = (long)B * C * N;
  const int nchunk = cdiv(L, BN_CHUNK);
  hipLaunchKernelGGL(bn_stats_kernel, dim3(nchunk, C, B), dim3(256), 0, s, x, C, L, nchunk, (float2 *)workspace);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(64), 0, s, (const float2 *)workspace, B * nchunk, (double)B * (double)L, eps, momentum, mean,
                     rstd, running_mean, running_var);
#define UNOPOSE_POOL_CASE(SS)                                                                                                                  \
  if (S == SS)                                                                                                                                 \
    hipLaunchKernelGGL(bn_relu_maxpool_kernel<SS>, dim3(pool_grid(rows, 256 / SS)), dim3(256), 0, s, x, C, rows, (const float *)mean,           \
                       (const float *)rstd, gamma, beta, N, out, idx);
  UNOPOSE_POOL_CASE(32) UNOPOSE_POOL_CASE(64) UNOPOSE_POOL_CASE(128) UNOPOSE_POOL_CASE(256)
#undef UNOPOSE_POOL_CASE
  return check_launch("bn_relu_maxpool_train_forward");
}

int unopose_bn_relu_maxpool_train_backward(const float *x, const float *g, const int32_t *idx, int B, int C, int N, int S, const float *gamma,
                                           const float *beta, const float *mean, const float *rstd, float *workspace, float *dgamma, float *dbeta,
                                           float *dx, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(x && g && idx && gamma && beta && mean && rstd && workspace && dgamma && dbeta && dx, "bn_relu_maxpool_train_backward: null pointer");
  UNOPOSE_REQUIRE(B >= 1 && B <= 65535 && C >= 1 && C <= 65535 && N >= 1 && (S == 32 || S == 64 || S == 128 || S == 256),
                  "bn_relu_maxpool_train_backward: bad sizes (B=%d C=%d N=%d S=%d)", B, C, N, S);
  hipStream_t s = (hipStream_t)stream;
  const long rows = (long)B * C * N;
  hipLaunchKernelGGL(bn_relu_maxpool_bwd_stats_kernel, dim3(C, B), dim3(256), 0, s, x, g, idx, C, N, S, mean, rstd, gamma, beta, (float2 *)workspace);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(64), 0, s, (const float2 *)workspace, B, dgamma, dbeta);
  const float inv = (float)(1.0 / ((double)B * (double)N * (double)S));
#define UNOPOSE_POOL_CASE(SS)                                                                                                                  \
  if (S == SS)                                                                                                                                 \
    hipLaunchKernelGGL(bn_relu_maxpool_bwd_apply_kernel<SS>, dim3(pool_grid(rows, 256 / SS)), dim3(256), 0, s, x, g, idx, C, rows, N, mean, rstd, \
                       gamma, beta, (const float *)dgamma, (const float *)dbeta, inv, dx);
  UNOPOSE_POOL_CASE(32) UNOPOSE_POOL_CASE(64) UNOPOSE_POOL_CASE(128) UNOPOSE_POOL_CASE(256)
#undef UNOPOSE_POOL_CASE
  return check_launch("bn_relu_maxpool_train_backward");
}

}  // extern "C"
