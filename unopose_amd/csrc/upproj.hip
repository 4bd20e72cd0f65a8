// Pixel features WITHOUT the dense up-projection (gfx950).
//
// ViT_AE (core/unopose/model/oneref_feature_extraction.py:200-236) maps every patch token through Linear 3072 -> 4096,
// pixel-shuffles the result into a (4 side) x (4 side) map of 256-wide cells, upsamples it bilinearly to the crop size and
// gathers Np pixels (get_chosen_pixel_feats, model_utils.py:215-227).  A chosen pixel touches 4 cells of that map; with
// Np = 2048 pixels at most 8192 of the 21 904 cells of a 518 x 518 crop are ever read (27 % on the benchmark's crops), yet the
// dense GEMM computes all of them (2.2 TFLOP and 1.65 ms per step at 64 crops).  Cell (token, s), s = 4 (Y & 3) + (X & 3), is
// row `token` of the activations times the 256-row block s of the weight, so the needed cells form 16 row-gathered GEMMs:
//
//   plan   (needs only the pixel indices: runs on the side stream underneath the ViT)
//     mark     flags[crop][cell] = 1 for the 4 cells of every chosen pixel
//     count    counts[s][crop]   = marked cells of sub-position s in the crop
//     offsets  groups laid out one after another, each padded to whole 256-row tiles; tile_info = {tiles, first tile of g}
//     fill     row_list[compact row] = activation row (crop, token) in (s, crop, token) order; cellmap[crop][cell] = compact row
//   GEMM   csrc/gemm.hip, gathered form: C (compact rows, 256) bf16
//   sample the bilinear blend of the 4 compact rows of each pixel (same arithmetic as bilinear_sample_kernel)
//
// Everything stays on the device: the number of tiles is read by the GEMM from tile_info[0].
#include <algorithm>

#include "common.h"

namespace unopose {

typedef unsigned short u16;

__device__ __forceinline__ float up_bf2f(u16 h) { return __uint_as_float((uint32_t)h << 16); }

// flags <- 0 and row_list <- -1 in ONE launch (two hipMemsetAsync cost two ~50-us runtime fill kernels in front of every plan)
__global__ __launch_bounds__(256) void upproj_clear_kernel(int *__restrict__ flags, long nflags, int *__restrict__ row_list, long nrows) {
  const long n = nflags + nrows, step = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += step) {
    if (i < nflags) flags[i] = 0;
    else row_list[i - nflags] = -1;
  }
}

__global__ __launch_bounds__(256) void upproj_mark_kernel(const long long *__restrict__ choose, int Np, int H, int W, int side,
                                                          int *__restrict__ flags) {
  const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
  if (p >= Np) return;
  const int hw = 4 * side;
  const BilinearTap t = bilinear_tap(choose[(size_t)b * Np + p], H, W, hw);
  int *f = flags + (size_t)b * hw * hw;
  f[t.y0 * hw + t.x0] = 1;
  f[t.y0 * hw + t.x1] = 1;
  f[t.y1 * hw + t.x0] = 1;
  f[t.y1 * hw + t.x1] = 1;
}

__device__ __forceinline__ int up_cell(int t, int s, int side) {  // map pixel of (token t, sub-position s)
  const int ty = t / side, tx = t - ty * side;
  return (4 * ty + (s >> 2)) * (4 * side) + 4 * tx + (s & 3);
}

__global__ __launch_bounds__(256) void upproj_count_kernel(const int *__restrict__ flags, int side, int B2, int *__restrict__ counts) {
  __shared__ int cnt[16];
  const int b = blockIdx.x, hw = 4 * side;
  if (threadIdx.x < 16) cnt[threadIdx.x] = 0;
  __syncthreads();
  const int *f = flags + (size_t)b * hw * hw;
  int local[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) local[s] = 0;
  for (int t = threadIdx.x; t < side * side; t += 256) {
#pragma unroll
    for (int s = 0; s < 16; ++s) local[s] += f[up_cell(t, s, side)];
  }
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const int v = (int)wave_sum_f32((float)local[s]);  // exact: counts < 2^24
    if ((threadIdx.x & 63) == 0) atomicAdd(&cnt[s], v);
  }
  __syncthreads();
  if (threadIdx.x < 16) counts[threadIdx.x * B2 + b] = cnt[threadIdx.x];
}

__global__ __launch_bounds__(64) void upproj_offsets_kernel(const int *__restrict__ counts, int B2, int *__restrict__ offsets,
                                                            int *__restrict__ tile_info) {
  __shared__ int total[16], start[17];
  const int s = threadIdx.x;
  if (s < 16) {
    int t = 0;
    for (int b = 0; b < B2; ++b) t += counts[s * B2 + b];
    total[s] = t;
  }
  __syncthreads();
  if (s == 0) {
    int run = 0;
    for (int g = 0; g < 16; ++g) {
      start[g] = run;
      run += (total[g] + 255) & ~255;
    }
    start[16] = run;
    tile_info[0] = run >> 8;  // <= cap_rows / 256: the host checked the worst case
    for (int g = 0; g <= 16; ++g) tile_info[1 + g] = start[g] >> 8;
  }
  __syncthreads();
  if (s < 16) {
    int run = start[s];
    for (int b = 0; b < B2; ++b) {
      offsets[s * B2 + b] = run;
      run += counts[s * B2 + b];
    }
  }
}

__global__ __launch_bounds__(512) void upproj_fill_kernel(const int *__restrict__ flags, const int *__restrict__ offsets, int side, int B2, int tok_offset,
                                                          int tok_stride, int *__restrict__ row_list, int *__restrict__ cellmap) {
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hw = 4 * side;
  const int *f = flags + (size_t)b * hw * hw;
  int *cm = cellmap + (size_t)b * hw * hw;
  for (int s = wave * 2; s < wave * 2 + 2; ++s) {
    int base = offsets[s * B2 + b];
    for (int t0 = 0; t0 < side * side; t0 += 64) {
      const int t = t0 + lane;
      const int cell = t < side * side ? up_cell(t, s, side) : 0;
      const bool on = t < side * side && f[cell] != 0;
      const unsigned long long m = __ballot(on);
      if (on) {
        const int r = base + __popcll(m & ((1ull << lane) - 1ull));
        row_list[r] = b * tok_stride + tok_offset + t;
        cm[cell] = r;
      }
      base += __popcll(m);
    }
  }
}

// One wavefront per output pixel, 4 channels per lane: the arithmetic of bilinear_sample_kernel<true> on compact rows.
template <bool OUT_BF16>  // (the result rounded to bf16: what an autocast Linear makes of it next -- the fine matcher's in_proj reads it directly)
__global__ __launch_bounds__(256) void bilinear_sample_compact_kernel(const u16 *__restrict__ Cc, const int *__restrict__ cellmap,
                                                                      const long long *__restrict__ choose, int side, int Np, int H,
                                                                      int W, void *__restrict__ outv) {
  const int b = blockIdx.y, lane = threadIdx.x & 63;
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= Np) return;
  const int hw = 4 * side;
  const BilinearTap tp = bilinear_tap(choose[(size_t)b * Np + p], H, W, hw);
  const int *cm = cellmap + (size_t)b * hw * hw;
  const int rows[4] = {cm[tp.y0 * hw + tp.x0], cm[tp.y0 * hw + tp.x1], cm[tp.y1 * hw + tp.x0], cm[tp.y1 * hw + tp.x1]};
  float v[4][4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint2 r = *reinterpret_cast<const uint2 *>(Cc + (size_t)rows[k] * 256 + lane * 4);
    v[k][0] = up_bf2f((u16)(r.x & 0xFFFF));
    v[k][1] = up_bf2f((u16)(r.x >> 16));
    v[k][2] = up_bf2f((u16)(r.y & 0xFFFF));
    v[k][3] = up_bf2f((u16)(r.y >> 16));
  }
  const float ly = tp.ly, lx = tp.lx;
  float4 res;
  float *rp = &res.x;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float top = (1.f - lx) * v[0][c] + lx * v[1][c];
    const float bot = (1.f - lx) * v[2][c] + lx * v[3][c];
    rp[c] = (1.f - ly) * top + ly * bot;
  }
  if (OUT_BF16)
    *reinterpret_cast<uint2 *>(reinterpret_cast<u16 *>(outv) + ((size_t)b * Np + p) * 256 + lane * 4) =
        make_uint2(cvt_pk_bf16_f32(res.x, res.y), cvt_pk_bf16_f32(res.z, res.w));
  else
    *reinterpret_cast<float4 *>(reinterpret_cast<float *>(outv) + ((size_t)b * Np + p) * 256 + lane * 4) = res;
}

}  // namespace unopose

using namespace unopose;

extern "C" {

int unopose_upproj_plan(const long long *choose, int B2, int Np, int H, int W, int side, int tok_offset, int tok_stride, int cap_rows,
                        int *ws, int *row_list, int *cellmap, int *tile_info, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(choose && ws && row_list && cellmap && tile_info, "upproj_plan: null pointer");
  UNOPOSE_REQUIRE(B2 >= 1 && B2 <= 65535 && Np >= 1 && H >= 1 && W >= 1 && side >= 1 && side <= 256 && tok_offset >= 0 &&
                      tok_stride >= tok_offset + side * side && cap_rows >= 0 && cap_rows % 256 == 0,
                  "upproj_plan: bad sizes");
  UNOPOSE_REQUIRE((long)B2 * tok_stride < (1L << 31), "upproj_plan: too many activation rows");
  {
    const long per_crop = std::min<long>(4L * Np, 16L * side * side);
    UNOPOSE_REQUIRE(cap_rows >= B2 * per_crop + 16 * 256, "upproj_plan: cap_rows %d below the worst case %ld (B2 * min(4 Np, cells) + 16 * 256)",
                    cap_rows, B2 * per_crop + 16 * 256);
  }
  hipStream_t s = (hipStream_t)stream;
  const size_t cells = (size_t)B2 * 16 * side * side;
  int *flags = ws, *counts = ws + cells, *offsets = counts + 16 * B2;  // ws: cells + 32 B2 ints
  hipLaunchKernelGGL(upproj_clear_kernel, dim3((unsigned)std::min<long>(((long)cells + cap_rows + 255) / 256, 2048)), dim3(256), 0, s, flags, (long)cells,
                     row_list, (long)cap_rows);
  hipLaunchKernelGGL(upproj_mark_kernel, dim3(cdiv(Np, 256), B2), dim3(256), 0, s, choose, Np, H, W, side, flags);
  hipLaunchKernelGGL(upproj_count_kernel, dim3(B2), dim3(256), 0, s, flags, side, B2, counts);
  hipLaunchKernelGGL(upproj_offsets_kernel, dim3(1), dim3(64), 0, s, counts, B2, offsets, tile_info);
  hipLaunchKernelGGL(upproj_fill_kernel, dim3(B2), dim3(512), 0, s, flags, offsets, side, B2, tok_offset, tok_stride, row_list,
                     cellmap);
  return check_launch("upproj_plan");
}

int unopose_bilinear_sample_compact(const void *Cc, const int *cellmap, const long long *choose, int B2, int side, int Np, int H, int W,
                                    void *out, int out_bf16, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(Cc && cellmap && choose && out, "bilinear_sample_compact: null pointer");
  UNOPOSE_REQUIRE(B2 >= 0 && B2 <= 65535 && side >= 1 && Np >= 0 && H >= 1 && W >= 1, "bilinear_sample_compact: bad sizes");
  if (B2 == 0 || Np == 0) return UNOPOSE_OK;
  if (out_bf16)
    hipLaunchKernelGGL(bilinear_sample_compact_kernel<true>, dim3(cdiv(Np, 4), B2), dim3(256), 0, (hipStream_t)stream, (const u16 *)Cc, cellmap,
                       choose, side, Np, H, W, out);
  else
    hipLaunchKernelGGL(bilinear_sample_compact_kernel<false>, dim3(cdiv(Np, 4), B2), dim3(256), 0, (hipStream_t)stream, (const u16 *)Cc, cellmap,
                       choose, side, Np, H, W, out);
  return check_launch("bilinear_sample_compact");
}

}  // extern "C"
