// Data-movement glue of the forward as single-pass HIP kernels (C ABI part 2c).  Each replaces a chain of torch
// element-wise / cat / cast launches that rocprof showed as ~6 ms of a 37 ms step (280 launches):
//   * patchify_bf16:        both crops -> the zero-padded bf16 patch matrix of the ViT's patch-embedding GEMM
//                           (timm PatchEmbed, driven at oneref_feature_extraction.py:24-27; was cat + unfold copy + zeros + cast copy)
//   * vit_tokens_layernorm: patch GEMM output + pos_embed, class / register tokens in front (timm _pos_embed, no_embed_class)
//                           and the first block's LayerNorm in the same pass (was add + cat + LayerNorm)
//   * row_dot:              the overlap-score heads nn.Linear(256, 1) (C:66, Fi:89) as a row dot product
//   * normalize_rows:       F.normalize(f) / temp -> bf16 operands of the fine assignment (model_utils.py:260-282)
//   * transpose_pad:        the channel-major zero-padded V image the token attention reads (was zeros + strided copy)
#include <algorithm>

#include "common.h"

namespace unopose {

typedef unsigned short u16;

// ---- (B,3,S,S) fp32 x 2 -> (2 B P, Kp) bf16, P = (S/14)^2, column = c * 196 + py * 14 + px, columns >= 588 zero.
// One workgroup per (crop, patch row gy): the 3 x 14 x S strip is read with coalesced rows and written patch-major.
//      SPLIT: the fp32 values in the split layout of csrc/gemm_f32.hip instead (per row and 32-column block one 128-byte line [hi (32 bf16) |
//      lo (32 bf16)]; Kp % 32 == 0): the patch matrix of the no-autocast forward, no unfold copy / zero fill / split pass.
template <bool SPLIT>
__global__ __launch_bounds__(256) void patchify_kernel(const float *__restrict__ a, int na, const float *__restrict__ b, int S,
                                                       int Kp, u16 *__restrict__ out) {
  const int g = S / 14;
  const int crop = blockIdx.x / g, gy = blockIdx.x - crop * g;
  const float *src = (crop < na ? a + (size_t)crop * 3 * S * S : b + (size_t)(crop - na) * 3 * S * S);
  u16 *dst = out + ((size_t)crop * g * g + (size_t)gy * g) * Kp * (SPLIT ? 2 : 1);
  // element e of the strip: patch gx, column k (k < Kp); consecutive threads take consecutive (gx, k) pairs of the OUTPUT
  // (2-byte coalesced stores); the reads of one patch row segment (14 floats) stay inside a 56-byte window: L1-served
  const int total = g * (Kp / 2);
  for (int e = threadIdx.x; e < total; e += 256) {
    const int gx = e / (Kp / 2), k2 = (e - gx * (Kp / 2)) * 2;
    float v[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = k2 + j;
      if (k < 588) {
        const int c = k / 196, r = k - c * 196, py = r / 14, px = r - py * 14;
        v[j] = src[((size_t)c * S + gy * 14 + py) * S + gx * 14 + px];
      } else {
        v[j] = 0.f;
      }
    }
    if (SPLIT) {
      const uint32_t hi = cvt_pk_bf16_f32(v[0], v[1]);
      const uint32_t lo = cvt_pk_bf16_f32(v[0] - __uint_as_float(hi << 16), v[1] - __uint_as_float(hi & 0xffff0000u));
      u16 *line = dst + (size_t)gx * Kp * 2 + (size_t)(k2 >> 5) * 64 + (k2 & 31);  // (u16 units: 64 per 128-byte line, hi half first)
      *reinterpret_cast<uint32_t *>(line) = hi;
      *reinterpret_cast<uint32_t *>(line + 32) = lo;
    } else {
      *reinterpret_cast<uint32_t *>(dst + (size_t)gx * Kp + k2) = cvt_pk_bf16_f32(v[0], v[1]);
    }
  }
}

// ---- tokens + first LayerNorm.  One wavefront per token row; C = 768 = 64 lanes x 12 (three float4 per lane).
// x[r] (fp32, the residual stream) = prefix token (cls, reg0..) for t < npre, else patch[b, t - npre] (bf16) + pos[t - npre];
// n1[r] (bf16) = LayerNorm(x[r]) * w + bias.
//      F32 (the no-autocast forward): patch is fp32, n1 comes out in the split layout of csrc/gemm_f32.hip.
template <bool F32>
__global__ __launch_bounds__(256) void vit_tokens_layernorm_kernel(const void *__restrict__ patch_, const float *__restrict__ pos,
                                                                   const float *__restrict__ prefix, int npre, int P, long rows,
                                                                   const float *__restrict__ w, const float *__restrict__ bias, float eps,
                                                                   float *__restrict__ x, u16 *__restrict__ n1) {
  const u16 *patch = reinterpret_cast<const u16 *>(patch_);
  constexpr int C = 768;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  const int T = npre + P;
  const long bimg = r / T;
  const int t = (int)(r - bimg * T);
  float v[12];
  if (t < npre) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float4 q = *reinterpret_cast<const float4 *>(prefix + (size_t)t * C + i * 256 + lane * 4);
      v[4 * i] = q.x, v[4 * i + 1] = q.y, v[4 * i + 2] = q.z, v[4 * i + 3] = q.w;
    }
  } else {
    const u16 *pr = patch + ((size_t)bimg * P + (t - npre)) * C;
    const float *ps = pos + (size_t)(t - npre) * C;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float4 q = *reinterpret_cast<const float4 *>(ps + i * 256 + lane * 4);
      if (F32) {
        const float4 f = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(patch_) + ((size_t)bimg * P + (t - npre)) * C + i * 256 + lane * 4);
        v[4 * i] = f.x + q.x, v[4 * i + 1] = f.y + q.y, v[4 * i + 2] = f.z + q.z, v[4 * i + 3] = f.w + q.w;
        continue;
      }
      const uint2 h = *reinterpret_cast<const uint2 *>(pr + i * 256 + lane * 4);
      v[4 * i] = __uint_as_float(h.x << 16) + q.x;
      v[4 * i + 1] = __uint_as_float(h.x & 0xffff0000u) + q.y;
      v[4 * i + 2] = __uint_as_float(h.y << 16) + q.z;
      v[4 * i + 3] = __uint_as_float(h.y & 0xffff0000u) + q.w;
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 12; ++i) s += v[i];
  const float mean = wave_sum_f32(s) * (1.f / C);
  float q2 = 0.f;
#pragma unroll
  for (int i = 0; i < 12; ++i) q2 += (v[i] - mean) * (v[i] - mean);
  const float rstd = rsqrtf(wave_sum_f32(q2) * (1.f / C) + eps);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int c = i * 256 + lane * 4;
    *reinterpret_cast<float4 *>(x + (size_t)r * C + c) = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
    const float4 ww = *reinterpret_cast<const float4 *>(w + c), bb = *reinterpret_cast<const float4 *>(bias + c);
    const float y0 = (v[4 * i] - mean) * rstd * ww.x + bb.x, y1 = (v[4 * i + 1] - mean) * rstd * ww.y + bb.y;
    const float y2 = (v[4 * i + 2] - mean) * rstd * ww.z + bb.z, y3 = (v[4 * i + 3] - mean) * rstd * ww.w + bb.w;
    if (F32) {
      uint2 h, l;
      h.x = cvt_pk_bf16_f32(y0, y1);
      h.y = cvt_pk_bf16_f32(y2, y3);
      l.x = cvt_pk_bf16_f32(y0 - __uint_as_float(h.x << 16), y1 - __uint_as_float(h.x & 0xffff0000u));
      l.y = cvt_pk_bf16_f32(y2 - __uint_as_float(h.y << 16), y3 - __uint_as_float(h.y & 0xffff0000u));
      char *line = reinterpret_cast<char *>(n1) + (size_t)r * C * 4 + (size_t)(c >> 5) * 128 + ((c >> 3) & 3) * 16 + ((c >> 2) & 1) * 8;
      *reinterpret_cast<uint2 *>(line) = h;
      *reinterpret_cast<uint2 *>(line + 64) = l;
    } else {
      *reinterpret_cast<uint2 *>(n1 + (size_t)r * C + c) = make_uint2(cvt_pk_bf16_f32(y0, y1), cvt_pk_bf16_f32(y2, y3));
    }
  }
}

// ---- out[r] = sum_c x[r, c] * w[c] + b   (x bf16 or fp32, C = 256: one wavefront per row, 4 channels per lane)
template <bool X_BF16, bool OUT_BF16>
__global__ __launch_bounds__(256) void row_dot_kernel(const void *__restrict__ x, const float *__restrict__ w, float b, long rows,
                                                      void *__restrict__ out) {
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  const float4 ww = *reinterpret_cast<const float4 *>(w + lane * 4);
  float4 v;
  if (X_BF16) {
    const uint2 h = *reinterpret_cast<const uint2 *>(reinterpret_cast<const u16 *>(x) + (size_t)r * 256 + lane * 4);
    v = make_float4(__uint_as_float(h.x << 16), __uint_as_float(h.x & 0xffff0000u), __uint_as_float(h.y << 16), __uint_as_float(h.y & 0xffff0000u));
  } else {
    v = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(x) + (size_t)r * 256 + lane * 4);
  }
  const float s = wave_sum_f32((v.x * ww.x + v.y * ww.y) + (v.z * ww.z + v.w * ww.w)) + b;
  if (lane == 0) {
    if (OUT_BF16) reinterpret_cast<u16 *>(out)[r] = (u16)(cvt_pk_bf16_f32(s, 0.f) & 0xffffu);
    else reinterpret_cast<float *>(out)[r] = s;
  }
}

// ---- out[r, :] = bf16( x[r, :] / max(||x[r, :]||, 1e-12) / temp )   (F.normalize(f, p=2, dim=-1) / temp of
//      compute_feature_similarity, model_utils.py:260-282), 256-wide rows, one wavefront per row
//      OUT_F32: the same bf16-rounded values stored as fp32 (the operand type of csrc/bmm_f32.hip: no cast launch in between)
//      OUT_F32 == 2: the unrounded fp32 values (F.normalize(x) / temp at the reference's default precision)
template <bool X_BF16, int OUT_F32>
__global__ __launch_bounds__(256) void normalize_rows_kernel(const void *__restrict__ x, long rows, float temp, void *__restrict__ outv) {
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  float4 v;
  if (X_BF16) {
    const uint2 h = *reinterpret_cast<const uint2 *>(reinterpret_cast<const u16 *>(x) + (size_t)r * 256 + lane * 4);
    v = make_float4(__uint_as_float(h.x << 16), __uint_as_float(h.x & 0xffff0000u), __uint_as_float(h.y << 16), __uint_as_float(h.y & 0xffff0000u));
  } else {
    v = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(x) + (size_t)r * 256 + lane * 4);
  }
  const float nrm = fmaxf(sqrtf(wave_sum_f32((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w))), 1e-12f);
  const float y0 = v.x / nrm / temp, y1 = v.y / nrm / temp, y2 = v.z / nrm / temp, y3 = v.w / nrm / temp;
  if (OUT_F32 == 2) {
    *reinterpret_cast<float4 *>(reinterpret_cast<float *>(outv) + (size_t)r * 256 + lane * 4) = make_float4(y0, y1, y2, y3);
    return;
  }
  const uint2 pk = make_uint2(cvt_pk_bf16_f32(y0, y1), cvt_pk_bf16_f32(y2, y3));
  if (OUT_F32)
    *reinterpret_cast<float4 *>(reinterpret_cast<float *>(outv) + (size_t)r * 256 + lane * 4) =
        make_float4(__uint_as_float(pk.x << 16), __uint_as_float(pk.x & 0xffff0000u), __uint_as_float(pk.y << 16), __uint_as_float(pk.y & 0xffff0000u));
  else
    *reinterpret_cast<uint2 *>(reinterpret_cast<u16 *>(outv) + (size_t)r * 256 + lane * 4) = pk;
}

// ---- vt[b, c, j] = v[b, j, c] for j < m, 0 for m <= j < pad   (the channel-major, zero-padded V image of csrc/attn.hip;
//      v rows `ld` elements apart inside the k | v projection output).  One workgroup per (64 channels, cloud).
__global__ __launch_bounds__(256) void transpose_pad_kernel(const u16 *__restrict__ v, long ld, int m, int C, int pad, u16 *__restrict__ vt) {
  extern __shared__ u16 tile[];  // [64][pad + 2]
  const int b = blockIdx.y, c0 = blockIdx.x * 64, tp = pad + 2;
  const u16 *src = v + (size_t)b * m * ld + c0;
  for (int i = threadIdx.x; i < 64 * pad; i += 256) {
    const int j = i >> 6, c = i & 63;
    tile[c * tp + j] = j < m ? src[(size_t)j * ld + c] : (u16)0;
  }
  __syncthreads();
  u16 *dst = vt + ((size_t)b * C + c0) * pad;
  for (int i = threadIdx.x; i < 64 * (pad / 2); i += 256) {
    const int c = i / (pad / 2), j2 = (i - c * (pad / 2)) * 2;
    *reinterpret_cast<uint32_t *>(dst + (size_t)c * pad + j2) = (uint32_t)tile[c * tp + j2] | ((uint32_t)tile[c * tp + j2 + 1] << 16);
  }
}

// the same for fp32 data (the value image of csrc/attn_f32.hip; 32 channels per workgroup)
__global__ __launch_bounds__(256) void transpose_pad_f32_kernel(const float *__restrict__ v, long ld, int m, int C, int pad, float *__restrict__ vt) {
  extern __shared__ float tile_f[];  // [32][pad + 1]
  const int b = blockIdx.y, c0 = blockIdx.x * 32, tp = pad + 1;
  const float *src = v + (size_t)b * m * ld + c0;
  for (int i = threadIdx.x; i < 32 * pad; i += 256) {
    const int j = i >> 5, c = i & 31;
    tile_f[c * tp + j] = j < m ? src[(size_t)j * ld + c] : 0.f;
  }
  __syncthreads();
  float *dst = vt + ((size_t)b * C + c0) * pad;
  for (int i = threadIdx.x; i < 32 * pad; i += 256) {
    const int c = i / pad, j = i - c * pad;
    dst[(size_t)c * pad + j] = tile_f[c * tp + j];
  }
}

// ---- out[b, p + j, :] = feats[b, idx[b, j] - off, :] (rows of `words` 4-byte units); idx - off < 0 selects alt[b, :] (the
//      background token of the sparse-to-dense block, transformer.py:655-662); with `prepend` row 0 of every batch is alt[b, :] too.
//      One launch instead of index cast + clamp + gather + compare + where + cat.
template <typename IDX>
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint32_t *__restrict__ feats, const IDX *__restrict__ idx,
                                                          const uint32_t *__restrict__ alt, int alt_words, int N, int J, int words, int off,
                                                          int prepend, uint32_t *__restrict__ out) {
  const int b = blockIdx.y, rows = J + prepend;
  const long total = (long)rows * words;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int r = (int)(e / words), w = (int)(e - (long)r * words);
    long src = -1;
    if (r >= prepend) src = (long)idx[(size_t)b * J + (r - prepend)] - off;
    uint32_t v;
    if (src < 0)
      v = alt ? alt[(size_t)b * alt_words + w] : 0u;
    else
      v = feats[((size_t)b * N + (size_t)min(src, (long)N - 1)) * words + w];
    out[((size_t)b * rows + r) * words + w] = v;
  }
}

// ---- nearest partner + "any partner within a radius" of every point of one cloud in the other (training labels, loss_utils.py:150-176:
//      `dist = sqrt(pairwise_distance(a, b)); close = dist <= thr; dist.min(2) / dist.min(1)`): the (B, n, m) distance matrix is never
//      formed (537 MB per tensor and ~15 elementwise / reduction passes at 4096 x 4096).  The value of an entry is computed exactly as the
//      reference's expansion does, whichever index is reduced: d(i, j) = sqrt(max((|a_i|^2 - 2 a_i.b_j) + |b_j|^2, 0)) with the squares and
//      the dot product summed left to right; `over_b` = reduce over j for every i (rows), else over i for every j (columns).  Ties: the
//      first index.
__global__ __launch_bounds__(256) void nearest_partner_kernel(const float *__restrict__ a, const float *__restrict__ b, int n, int m, int over_b,
                                                              float thr, float *__restrict__ dmin, int32_t *__restrict__ arg,
                                                              uint8_t *__restrict__ anyc) {
  __shared__ float4 tile[256];  // (x, y, z, |.|^2) of 256 points of the reduced cloud
  const int bb = blockIdx.y, nout = over_b ? n : m, nred = over_b ? m : n;
  const float *own = (over_b ? a : b) + (size_t)bb * nout * 3, *oth = (over_b ? b : a) + (size_t)bb * nred * 3;
  const int o = blockIdx.x * 256 + threadIdx.x;
  float px = 0.f, py = 0.f, pz = 0.f;
  if (o < nout) px = own[o * 3], py = own[o * 3 + 1], pz = own[o * 3 + 2];
  const float pp = (px * px + py * py) + pz * pz;
  float best = 3.4e38f;
  int bi = 0;
  bool any = false;
  for (int t0 = 0; t0 < nred; t0 += 256) {
    __syncthreads();
    const int q = t0 + threadIdx.x;
    if (q < nred) {
      const float x = oth[q * 3], y = oth[q * 3 + 1], z = oth[q * 3 + 2];
      tile[threadIdx.x] = make_float4(x, y, z, (x * x + y * y) + z * z);
    }
    __syncthreads();
    const int cnt = min(256, nred - t0);
    for (int u = 0; u < cnt; ++u) {
      const float4 c = tile[u];
      // a is the FIRST operand of the reference's expansion whichever side this thread's point is on
      const float dotv = over_b ? (px * c.x + py * c.y) + pz * c.z : (c.x * px + c.y * py) + c.z * pz;
      const float aa = over_b ? pp : c.w, bbq = over_b ? c.w : pp;
      const float d = sqrtf(fmaxf((aa - 2.f * dotv) + bbq, 0.f));
      any |= d <= thr;
      if (d < best) best = d, bi = t0 + u;
    }
  }
  if (o < nout) {
    dmin[(size_t)bb * nout + o] = best;
    arg[(size_t)bb * nout + o] = bi;
    anyc[(size_t)bb * nout + o] = any ? 1 : 0;
  }
}

}  // namespace unopose

using namespace unopose;

namespace unopose {

// ---- Round 5: the last fp32 reductions / elementwise ops of the eval forward that ran as torch kernels (`sum`, `mean`, `norm`, `mul`,
// `add`, `sigmoid`: the seven at::native kernels with packed fp32 instructions of tests/test_torch_glue_isa_gpu.py's reviewed list) --
// tiny tensors, one launch each, deterministic reductions (fixed order, no atomics).

__device__ __forceinline__ float block256_sum(float v, float *sh) {   // deterministic: wave sums, then the four partials in order
  v = wave_sum_f32(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  const float r = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  __syncthreads();
  return r;
}
__device__ __forceinline__ float block256_max(float v, float *sh) {
  v = wave_max_f32(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  const float r = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
  __syncthreads();
  return r;
}

// radius[b] = max_i | p_i - mean(p) |  (oneref_grf_predator_pose_estimation_model.py: `torch.norm(pts - pts.mean(1), dim=2).max(1)[0]`)
__global__ __launch_bounds__(256) void cloud_radius_kernel(const float *__restrict__ pts, int N, float *__restrict__ radius) {
  __shared__ float sh[4];
  const float *P = pts + (size_t)blockIdx.x * N * 3;
  // the mean in double precision, rounded once: the correctly rounded value is what any accurate fp32 summation order agrees with most often
  // (the FPS that follows is run on pts / radius: its indices are compared bit for bit with the reference's)
  __shared__ double shd[3][4];
  double sx = 0., sy = 0., sz = 0.;
  for (int i = threadIdx.x; i < N; i += 256) {
    sx += (double)P[3 * i];
    sy += (double)P[3 * i + 1];
    sz += (double)P[3 * i + 2];
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    sx += __shfl_xor(sx, o);
    sy += __shfl_xor(sy, o);
    sz += __shfl_xor(sz, o);
  }
  if ((threadIdx.x & 63) == 0) shd[0][threadIdx.x >> 6] = sx, shd[1][threadIdx.x >> 6] = sy, shd[2][threadIdx.x >> 6] = sz;
  __syncthreads();
  const float mx = (float)(((shd[0][0] + shd[0][1]) + (shd[0][2] + shd[0][3])) / (double)N), my = (float)(((shd[1][0] + shd[1][1]) + (shd[1][2] + shd[1][3])) / (double)N),
              mz = (float)(((shd[2][0] + shd[2][1]) + (shd[2][2] + shd[2][3])) / (double)N);
  float r = 0.f;
  for (int i = threadIdx.x; i < N; i += 256) {
    const float dx = P[3 * i] - mx, dy = P[3 * i + 1] - my, dz = P[3 * i + 2] - mz;
    r = fmaxf(r, sqrtf((dx * dx + dy * dy) + dz * dz));
  }
  r = block256_max(r, sh);
  if (threadIdx.x == 0) radius[blockIdx.x] = r;
}

// out = x / (radius[b] + eps)  (mode 0)  or  x * (radius[b] + eps)  (mode 1), x (B, n) fp32
__global__ __launch_bounds__(256) void scale_by_radius_kernel(const float *__restrict__ x, int n, const float *__restrict__ radius, float eps, int mode,
                                                              float *__restrict__ out) {
  const float s = radius[blockIdx.y] + eps;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const size_t at = (size_t)blockIdx.y * n + i;
  out[at] = mode ? x[at] * s : x[at] / s;
}

// ---- rows copied between strided places: dst row r = src row r (src rows `src_stride` bytes apart, 0 = the same row for every r: a
//      background token put in front of every pair of a (2B, 1 + n, C) tensor without a concatenation), 4-byte words
__global__ __launch_bounds__(256) void copy_rows_kernel(const uint32_t *__restrict__ src, long src_words, uint32_t *__restrict__ dst, long dst_words,
                                                        int words) {
  const size_t r = blockIdx.y;
  for (int w = blockIdx.x * 256 + threadIdx.x; w < words; w += gridDim.x * 256) dst[r * dst_words + w] = src[r * src_words + w];
}

// ---- nan_to_num over a LIST of fp32 tensors in one launch (engine_utils.py:14-18 walks the parameters one torch.nan_to_num at a time: 303 launches per
//      training step here): tensor t = ptrs[t][0 .. sizes[t]), blockIdx.y = t, grid-stride over its elements; only non-finite values are written back.
__global__ __launch_bounds__(256) void nan_to_num_multi_kernel(float *const *__restrict__ ptrs, const long long *__restrict__ sizes, float nan_v, float pos_v,
                                                               float neg_v) {
  float *p = ptrs[blockIdx.y];
  const long long n = sizes[blockIdx.y];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float v = p[i];
    if (v != v) p[i] = nan_v;
    else if (v == __builtin_huge_valf()) p[i] = pos_v;
    else if (v == -__builtin_huge_valf()) p[i] = neg_v;
  }
}

// overlap scores: out[b][j] = clamp(sigmoid(scores[b][1 + j]), 0, 1) for j < n1, scores[b][n1 + 2 + (j - n1)] for the second cloud
// (oneref_predator_coarse_point_matching.py:68-76: the background tokens at 0 and n1 + 1 are dropped)
//   `halves`: the score head ran over the two clouds as ONE batch of 2B (cloud 2 of pair b is batch B + b): scores is (2B, n1 + 1) then (n2 = n1)
template <bool X_BF16>
__global__ __launch_bounds__(256) void overlap_scores_kernel(const void *__restrict__ scores, int n_tot, int n1, int halves, float *__restrict__ out) {
  const int n_out = n_tot - 2, j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n_out) return;
  const size_t src = halves ? ((size_t)(j < n1 ? blockIdx.y : gridDim.y + blockIdx.y) * (n1 + 1) + 1 + (j < n1 ? j : j - n1))
                            : (size_t)blockIdx.y * n_tot + (j < n1 ? 1 + j : 2 + j);
  const float x = X_BF16 ? __uint_as_float((uint32_t) reinterpret_cast<const u16 *>(scores)[src] << 16) : reinterpret_cast<const float *>(scores)[src];
  const float y = 1.f / (1.f + expf(-x));
  out[(size_t)blockIdx.y * n_out + j] = fminf(fmaxf(y, 0.f), 1.f);
}

// y = bf16( (p - t) @ R ) with the operands rounded to bf16 and fp32 accumulation -- what autocast's bf16 bmm computes (Fi:69); p (B,N,3) fp32
__global__ __launch_bounds__(256) void rigid_rows_bf16_kernel(const float *__restrict__ p, int N, const float *__restrict__ t, const float *__restrict__ R,
                                                              u16 *__restrict__ out) {
  const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  auto f2bf_rn = [](float v) { return (u16)(cvt_pk_bf16_f32(v, 0.f) & 0xffffu); };
  auto rb = [&](float v) { return __uint_as_float((uint32_t)f2bf_rn(v) << 16); };
  const float *pp = p + ((size_t)b * N + i) * 3, *tt = t + b * 3, *Rb = R + b * 9;
  const float x0 = rb(pp[0] - tt[0]), x1 = rb(pp[1] - tt[1]), x2 = rb(pp[2] - tt[2]);
  u16 *o = out + ((size_t)b * N + i) * 3;
#pragma unroll
  for (int j = 0; j < 3; ++j) o[j] = f2bf_rn((x0 * rb(Rb[j]) + x1 * rb(Rb[3 + j])) + x2 * rb(Rb[6 + j]));
}

// out[b][c] = sum_j x[b][j][c], x (B, J, C) bf16, fp32 sums in token order (the focused linear attention's k-sum, transformer.py:560-566)
__global__ __launch_bounds__(256) void token_sum_bf16_kernel(const u16 *__restrict__ x, int J, int C, float *__restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const u16 *X = x + (size_t)blockIdx.y * J * C + c;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int j = 0;
  for (; j + 3 < J; j += 4) {   // four independent chains: the loads of a column are C elements apart, latency-bound otherwise
    a0 += __uint_as_float((uint32_t)X[(size_t)j * C] << 16);
    a1 += __uint_as_float((uint32_t)X[(size_t)(j + 1) * C] << 16);
    a2 += __uint_as_float((uint32_t)X[(size_t)(j + 2) * C] << 16);
    a3 += __uint_as_float((uint32_t)X[(size_t)(j + 3) * C] << 16);
  }
  for (; j < J; ++j) a0 += __uint_as_float((uint32_t)X[(size_t)j * C] << 16);
  out[(size_t)blockIdx.y * C + c] = (a0 + a1) + (a2 + a3);
}

// pose score of compute_fine_Rt_overlap / compute_coarse_Rt_overlap's tail (model_utils.py:559-566):
//   ps = sum_i [dis_i < thr] w_i / (sum_i w_i + 1e-8) * mean_i w_i
__global__ __launch_bounds__(256) void pose_score_kernel(const float *__restrict__ dis, const float *__restrict__ w, int N, float thr, float *__restrict__ out) {
  __shared__ float sh[4];
  const float *D = dis + (size_t)blockIdx.x * N, *W = w + (size_t)blockIdx.x * N;
  float s1 = 0.f, s2 = 0.f;
  for (int i = threadIdx.x; i < N; i += 256) {
    const float wi = W[i];
    s1 += D[i] < thr ? wi : 0.f;
    s2 += wi;
  }
  s1 = block256_sum(s1, sh);
  s2 = block256_sum(s2, sh);
  if (threadIdx.x == 0) out[blockIdx.x] = s1 / (s2 + 1e-8f) * (s2 / (float)N);
}

}  // namespace unopose

using namespace unopose;

extern "C" {

int unopose_gather_rows(const void *feats, int B, int N, int row_bytes, const void *idx, int idx_is_i64, int J, int off, const void *alt,
                        long alt_stride_bytes, int prepend, void *out, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(feats && idx && out, "gather_rows: null pointer");
  UNOPOSE_REQUIRE(B >= 1 && B <= 65535 && N >= 1 && J >= 1 && row_bytes >= 4 && row_bytes % 4 == 0 && (off == 0 || off == 1) && (prepend == 0 || prepend == 1),
                  "gather_rows: bad sizes (B=%d N=%d J=%d row_bytes=%d)", B, N, J, row_bytes);
  UNOPOSE_REQUIRE(alt || (off == 0 && prepend == 0), "gather_rows: off / prepend need the alternative row");
  UNOPOSE_REQUIRE(!alt || ((alt_stride_bytes == 0 || alt_stride_bytes >= row_bytes) && alt_stride_bytes % 4 == 0 && alt_stride_bytes / 4 <= 0x7FFFFFFF),
                  "gather_rows: alt rows must be 0 (one row for all) or >= row_bytes apart, a multiple of 4 (got %ld)", alt_stride_bytes);
  const int words = row_bytes / 4, alt_words = (int)(alt_stride_bytes / 4);
  const dim3 grid((unsigned)std::min<long>(((long)(J + prepend) * words + 255) / 256, 4096), B);
  if (idx_is_i64)
    hipLaunchKernelGGL(gather_rows_kernel<long long>, grid, dim3(256), 0, (hipStream_t)stream, (const uint32_t *)feats, (const long long *)idx,
                       (const uint32_t *)alt, alt_words, N, J, words, off, prepend, (uint32_t *)out);
  else
    hipLaunchKernelGGL(gather_rows_kernel<int>, grid, dim3(256), 0, (hipStream_t)stream, (const uint32_t *)feats, (const int *)idx, (const uint32_t *)alt,
                       alt_words, N, J, words, off, prepend, (uint32_t *)out);
  return check_launch("gather_rows");
}

int unopose_patchify_bf16(const float *rgb_a, int na, const float *rgb_b, int nb, int S, int Kp, void *out, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(rgb_a && out && (rgb_b || nb == 0), "patchify_bf16: null pointer");
  UNOPOSE_REQUIRE(na >= 1 && nb >= 0 && S >= 14 && S % 14 == 0 && Kp >= 588 && Kp % 2 == 0, "patchify_bf16: needs S %% 14 == 0 and an even Kp >= 588 (got S=%d Kp=%d)", S, Kp);
  hipLaunchKernelGGL(patchify_kernel<false>, dim3((na + nb) * (S / 14)), dim3(256), 0, (hipStream_t)stream, rgb_a, na, rgb_b, S, Kp, (u16 *)out);
  return check_launch("patchify_bf16");
}

int unopose_patchify_split(const float *rgb_a, int na, const float *rgb_b, int nb, int S, int Kp, void *out, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(rgb_a && out && (rgb_b || nb == 0), "patchify_split: null pointer");
  UNOPOSE_REQUIRE(na >= 1 && nb >= 0 && S >= 14 && S % 14 == 0 && Kp >= 588 && Kp % 32 == 0, "patchify_split: needs S %% 14 == 0 and Kp >= 588, a multiple of 32 (got S=%d Kp=%d)", S, Kp);
  hipLaunchKernelGGL(patchify_kernel<true>, dim3((na + nb) * (S / 14)), dim3(256), 0, (hipStream_t)stream, rgb_a, na, rgb_b, S, Kp, (u16 *)out);
  return check_launch("patchify_split");
}

int unopose_vit_tokens_layernorm(const void *patch, const float *pos, const float *prefix, int npre, int P, int nimg, int C,
                                 const float *ln_w, const float *ln_b, float eps, float *x, void *n1, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(patch && pos && prefix && ln_w && ln_b && x && n1, "vit_tokens_layernorm: null pointer");
  UNOPOSE_REQUIRE(C == 768 && npre >= 0 && P >= 1 && nimg >= 1, "vit_tokens_layernorm: built for C = 768 (got %d)", C);
  const long rows = (long)nimg * (npre + P);
  hipLaunchKernelGGL(vit_tokens_layernorm_kernel<false>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, patch, pos,
                     prefix, npre, P, rows, ln_w, ln_b, eps, x, (u16 *)n1);
  return check_launch("vit_tokens_layernorm");
}

int unopose_vit_tokens_layernorm_f32(const float *patch, const float *pos, const float *prefix, int npre, int P, int nimg, int C,
                                     const float *ln_w, const float *ln_b, float eps, float *x, void *n1_split, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(patch && pos && prefix && ln_w && ln_b && x && n1_split, "vit_tokens_layernorm_f32: null pointer");
  UNOPOSE_REQUIRE(C == 768 && npre >= 0 && P >= 1 && nimg >= 1, "vit_tokens_layernorm_f32: built for C = 768 (got %d)", C);
  const long rows = (long)nimg * (npre + P);
  hipLaunchKernelGGL(vit_tokens_layernorm_kernel<true>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const void *)patch, pos,
                     prefix, npre, P, rows, ln_w, ln_b, eps, x, (u16 *)n1_split);
  return check_launch("vit_tokens_layernorm_f32");
}

int unopose_row_dot(const void *x, int x_bf16, const float *w, float b, long rows, int C, void *out, int out_bf16, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(x && w && out, "row_dot: null pointer");
  UNOPOSE_REQUIRE(C == 256 && rows >= 1, "row_dot: built for C = 256 (got %d)", C);
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (x_bf16 && out_bf16) hipLaunchKernelGGL((row_dot_kernel<true, true>), grid, block, 0, s, x, w, b, rows, out);
  else if (x_bf16) hipLaunchKernelGGL((row_dot_kernel<true, false>), grid, block, 0, s, x, w, b, rows, out);
  else if (out_bf16) hipLaunchKernelGGL((row_dot_kernel<false, true>), grid, block, 0, s, x, w, b, rows, out);
  else hipLaunchKernelGGL((row_dot_kernel<false, false>), grid, block, 0, s, x, w, b, rows, out);
  return check_launch("row_dot");
}

int unopose_normalize_rows_bf16(const void *x, int x_bf16, long rows, int C, float temp, void *out, int out_f32, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(x && out, "normalize_rows_bf16: null pointer");
  UNOPOSE_REQUIRE(C == 256 && rows >= 1 && temp > 0.f, "normalize_rows_bf16: built for C = 256 (got %d)", C);
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  hipStream_t s = (hipStream_t)stream;
  UNOPOSE_REQUIRE(out_f32 >= 0 && out_f32 <= 2, "normalize_rows_bf16: out_f32 is 0 (bf16), 1 (bf16-rounded values as float32) or 2 (unrounded float32)");
  if (out_f32 == 2 && x_bf16) hipLaunchKernelGGL((normalize_rows_kernel<true, 2>), grid, block, 0, s, x, rows, temp, out);
  else if (out_f32 == 2) hipLaunchKernelGGL((normalize_rows_kernel<false, 2>), grid, block, 0, s, x, rows, temp, out);
  else if (x_bf16 && out_f32) hipLaunchKernelGGL((normalize_rows_kernel<true, 1>), grid, block, 0, s, x, rows, temp, out);
  else if (x_bf16) hipLaunchKernelGGL((normalize_rows_kernel<true, 0>), grid, block, 0, s, x, rows, temp, out);
  else if (out_f32) hipLaunchKernelGGL((normalize_rows_kernel<false, 1>), grid, block, 0, s, x, rows, temp, out);
  else hipLaunchKernelGGL((normalize_rows_kernel<false, 0>), grid, block, 0, s, x, rows, temp, out);
  return check_launch("normalize_rows_bf16");
}

int unopose_transpose_pad_bf16(const void *v, long ld, int B, int m, int C, int pad, void *vt, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(v && vt, "transpose_pad_bf16: null pointer");
  UNOPOSE_REQUIRE(B >= 1 && m >= 1 && m <= pad && pad % 2 == 0 && pad <= 1024 && C % 64 == 0 && ld >= C, "transpose_pad_bf16: bad shape (m=%d pad=%d C=%d)", m, pad, C);
  hipLaunchKernelGGL(transpose_pad_kernel, dim3(C / 64, B), dim3(256), (size_t)64 * (pad + 2) * 2, (hipStream_t)stream, (const u16 *)v, ld, m, C,
                     pad, (u16 *)vt);
  return check_launch("transpose_pad_bf16");
}

int unopose_transpose_pad_f32(const float *v, long ld, int B, int m, int C, int pad, float *vt, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(v && vt, "transpose_pad_f32: null pointer");
  UNOPOSE_REQUIRE(B >= 1 && B <= 65535 && m >= 1 && m <= pad && pad <= 1024 && C % 32 == 0 && ld >= C, "transpose_pad_f32: bad shape (m=%d pad=%d C=%d)", m, pad, C);
  hipLaunchKernelGGL(transpose_pad_f32_kernel, dim3(C / 32, B), dim3(256), (size_t)32 * (pad + 1) * 4, (hipStream_t)stream, v, ld, m, C, pad, vt);
  return check_launch("transpose_pad_f32");
}

int unopose_cloud_radius(const float *pts, int B, int N, float *radius, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(pts && radius, "cloud_radius: null pointer");
  UNOPOSE_REQUIRE(B >= 0 && N >= 1, "cloud_radius: bad sizes");
  if (B == 0) return UNOPOSE_OK;
  hipLaunchKernelGGL(cloud_radius_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, pts, N, radius);
  return check_launch("cloud_radius");
}

int unopose_scale_by_radius(const float *x, int B, int n, const float *radius, float eps, int multiply, float *out, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(x && radius && out, "scale_by_radius: null pointer");
  UNOPOSE_REQUIRE(B >= 0 && n >= 1 && B <= 65535, "scale_by_radius: bad sizes");
  if (B == 0) return UNOPOSE_OK;
  hipLaunchKernelGGL(scale_by_radius_kernel, dim3(cdiv(n, 256), B), dim3(256), 0, (hipStream_t)stream, x, n, radius, eps, multiply, out);
  return check_launch("scale_by_radius");
}

int unopose_overlap_scores(const void *scores, int x_bf16, int B, int n_tot, int n1, int halves, float *out, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(scores && out, "overlap_scores: null pointer");
  UNOPOSE_REQUIRE(B >= 0 && B <= 65535 && n1 >= 1 && n_tot >= n1 + 3, "overlap_scores: needs two clouds behind their background tokens (n_tot=%d n1=%d)", n_tot, n1);
  UNOPOSE_REQUIRE(!halves || n_tot == 2 * (n1 + 1), "overlap_scores: the two-halves form needs clouds of one size (n_tot=%d n1=%d)", n_tot, n1);
  if (B == 0) return UNOPOSE_OK;
  const dim3 grid(cdiv(n_tot - 2, 256), B);
  if (x_bf16) hipLaunchKernelGGL(overlap_scores_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, scores, n_tot, n1, halves, out);
  else hipLaunchKernelGGL(overlap_scores_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, scores, n_tot, n1, halves, out);
  return check_launch("overlap_scores");
}

int unopose_copy_rows(const void *src, long src_stride_bytes, void *dst, long dst_stride_bytes, int rows, int row_bytes, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(src && dst, "copy_rows: null pointer");
  UNOPOSE_REQUIRE(rows >= 0 && rows <= 65535 && row_bytes >= 4 && row_bytes % 4 == 0 && src_stride_bytes >= 0 && src_stride_bytes % 4 == 0 &&
                      dst_stride_bytes >= row_bytes && dst_stride_bytes % 4 == 0,
                  "copy_rows: bad sizes (rows=%d row_bytes=%d)", rows, row_bytes);
  if (rows == 0) return UNOPOSE_OK;
  hipLaunchKernelGGL(copy_rows_kernel, dim3(std::min(cdiv(row_bytes / 4, 256), 64), rows), dim3(256), 0, (hipStream_t)stream, (const uint32_t *)src,
                     src_stride_bytes / 4, (uint32_t *)dst, dst_stride_bytes / 4, row_bytes / 4);
  return check_launch("copy_rows");
}

int unopose_rigid_rows_bf16(const float *p, int B, int N, const float *t, const float *R, void *out, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(p && t && R && out, "rigid_rows_bf16: null pointer");
  UNOPOSE_REQUIRE(B >= 0 && B <= 65535 && N >= 1, "rigid_rows_bf16: bad sizes");
  if (B == 0) return UNOPOSE_OK;
  hipLaunchKernelGGL(rigid_rows_bf16_kernel, dim3(cdiv(N, 256), B), dim3(256), 0, (hipStream_t)stream, p, N, t, R, (u16 *)out);
  return check_launch("rigid_rows_bf16");
}

int unopose_token_sum_bf16(const void *x, int B, int J, int C, float *out, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(x && out, "token_sum_bf16: null pointer");
  UNOPOSE_REQUIRE(B >= 0 && B <= 65535 && J >= 1 && C >= 1, "token_sum_bf16: bad sizes");
  if (B == 0) return UNOPOSE_OK;
  hipLaunchKernelGGL(token_sum_bf16_kernel, dim3(cdiv(C, 256), B), dim3(256), 0, (hipStream_t)stream, (const u16 *)x, J, C, out);
  return check_launch("token_sum_bf16");
}

int unopose_nan_to_num_multi(const void *ptrs_dev, const void *sizes_dev, int n_tensors, long max_size, float nan_value, float posinf_value,
                             float neginf_value, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(n_tensors == 0 || (ptrs_dev && sizes_dev), "nan_to_num_multi: null pointer");
  UNOPOSE_REQUIRE(n_tensors >= 0 && n_tensors <= 65535 && max_size >= 0, "nan_to_num_multi: bad sizes (%d tensors)", n_tensors);
  if (n_tensors == 0 || max_size == 0) return UNOPOSE_OK;
  const unsigned gx = (unsigned)std::min<long>((max_size + 255) / 256, 64);
  hipLaunchKernelGGL(nan_to_num_multi_kernel, dim3(gx, n_tensors), dim3(256), 0, (hipStream_t)stream, (float *const *)ptrs_dev,
                     (const long long *)sizes_dev, nan_value, posinf_value, neginf_value);
  return check_launch("nan_to_num_multi");
}

int unopose_pose_score(const float *dis, const float *w, int B, int N, float thr, float *out, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(dis && w && out, "pose_score: null pointer");
  UNOPOSE_REQUIRE(B >= 0 && N >= 1, "pose_score: bad sizes");
  if (B == 0) return UNOPOSE_OK;
  hipLaunchKernelGGL(pose_score_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dis, w, N, thr, out);
  return check_launch("pose_score");
}

int unopose_nearest_partner(const float *a, const float *b, int B, int n, int m, int over_b, float thr, float *dmin, int32_t *arg, uint8_t *any_close,
                            unopose_stream_t stream) {
  UNOPOSE_REQUIRE(a && b && dmin && arg && any_close, "nearest_partner: null pointer");
  UNOPOSE_REQUIRE(B >= 0 && B <= 65535 && n >= 1 && m >= 1, "nearest_partner: bad sizes (B=%d n=%d m=%d)", B, n, m);
  if (B == 0) return UNOPOSE_OK;
  hipLaunchKernelGGL(nearest_partner_kernel, dim3(cdiv(over_b ? n : m, 256), B), dim3(256), 0, (hipStream_t)stream, a, b, n, m, over_b, thr, dmin, arg,
                     any_close);
  return check_launch("nearest_partner");
}

}  // extern "C"
