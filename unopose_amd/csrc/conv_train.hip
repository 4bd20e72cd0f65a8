// 1 x 1 convolutions of the positional encoding's SharedMLP under autograd (C ABI part 3: training path).
//
// Replaces, per layer of core/unopose/model/pointnet2/pytorch_utils.py:25-132 under train(), nn.Conv2d(cin, cout, 1, bias=False) and
// its two gradients (MIOpen: NCHW <-> NHWC transposes + Sp3AsmConv forward, library GEMMs / igemm_wrw backward: 44 ms of the 188 ms
// step at BASELINE configs[3]).  The tensors are (B, C, N, S) fp32 = channel-major slabs of L = N * S contiguous positions, the
// channel counts are 6 / 32 / 64 / 128: tiny matrices against streams of up to 10.7 GB, so all three products are HBM-bound and
// the arithmetic runs on the fp32 matrix instruction (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulation -- the
// precision of the convolution it replaces, no operand splitting):
//   forward / input gradient   y[b, m, l] = sum_k W[m, k] x[b, k, l]      (the input gradient is the same kernel on W^T)
//        lanes run along l in BOTH the B operand (k = ci row, n = position) and the D tile (column = position): every load and
//        store is a run of 256 contiguous bytes per channel row; W^T sits in LDS ([k][m]: conflict-free A reads);
//   weight gradient            dW[m, k] = sum_{b, l} dy[b, m, l] x[b, k, l]
//        the reduction runs along l, so both operands need lanes along CHANNELS: a workgroup stages a (channels x 64 positions)
//        tile of both tensors in LDS with coalesced loads and the waves read it back transposed (row stride 66 floats: the 32 rows
//        of a fragment read fall on 32 different banks); per-workgroup partial sums, combined in double by a second kernel
//        (deterministic: no atomics).
#include "common.h"

namespace unopose {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------------------------- forward / dgrad
// K: input channels rounded up to an even number of MFMA k-steps (rows >= cin read as zero); M: output channels (multiple of 32).
// A wave owns 64 consecutive positions of one slab: two 32-column tiles (even / odd positions, so a lane's pair is one 8-byte access).
template <int K, int M>
__global__ __launch_bounds__(256, 2) void conv1x1_f32_kernel(const float *__restrict__ x, const float *__restrict__ w, int cin, int Bn, long L,
                                                             float *__restrict__ y) {
  __shared__ float Wt[K * M];  // [k][m]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (uniform: tile addresses live in SGPRs)
  for (int e = tid; e < K * M; e += 256) {
    const int k = e / M, m = e - k * M;
    Wt[e] = k < cin ? w[(size_t)m * cin + k] : 0.f;
  }
  __syncthreads();
  const int col = lane & 31, kh = lane >> 5;
  const long tiles_per_slab = L >> 6, total = (long)Bn * tiles_per_slab;
  // the input rows arrive in chunks of KC channels, the next chunk (of this tile or of the wave's next tile) in flight under the
  // products of the current one; the chunk loop stays rolled (two chunks per trip: the buffers alternate) so that the compiler
  // does not hoist a whole tile's LDS reads above the products
  constexpr int KC = K < 32 ? K : 16, NCH = K / KC;
  static_assert(NCH == 1 || NCH % 2 == 0, "chunk pairs");
  float2 bv[2][KC / 2];
  auto fetch = [&](long tile, int ch, float2 (&dst)[KC / 2]) {
    const long b = tile / tiles_per_slab, l0 = (tile - b * tiles_per_slab) << 6;
    const float *xb = x + ((size_t)b * cin + ch * KC) * L + l0;  // uniform
    const int lo = kh * (int)L + 2 * col;                          // this lane's element offset inside the chunk's first row pair
#pragma unroll
    for (int s = 0; s < KC / 2; ++s) {
      const int k = ch * KC + 2 * s + kh;
      dst[s] = k < cin ? *reinterpret_cast<const float2 *>(xb + (size_t)(2 * s) * L + lo) : make_float2(0.f, 0.f);
    }
  };
  const long first = (long)blockIdx.x * 4 + wave, stride = (long)gridDim.x * 4;
  if (first < total) fetch(first, 0, bv[0]);
  for (long tile = first; tile < total; tile += stride) {
    const long b = tile / tiles_per_slab, l0 = (tile - b * tiles_per_slab) << 6;
    f32x16 acc[M / 32][2];
#pragma unroll
    for (int m = 0; m < M / 32; ++m)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;
    auto products = [&](int ch, const float2 (&cur)[KC / 2]) {
      const float *wr = &Wt[(ch * KC + kh) * M + col];
#pragma unroll
      for (int s = 0; s < KC / 2; ++s) {
#pragma unroll
        for (int m = 0; m < M / 32; ++m) {
          const float a = wr[2 * s * M + m * 32];
          acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, cur[s].x, acc[m][0], 0, 0, 0);
          acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, cur[s].y, acc[m][1], 0, 0, 0);
        }
      }
    };
    if (NCH == 1) {
      float2 cur[KC / 2];
#pragma unroll
      for (int s = 0; s < KC / 2; ++s) cur[s] = bv[0][s];
      if (tile + stride < total) fetch(tile + stride, 0, bv[0]);
      products(0, cur);
    } else {
#pragma unroll 1
      for (int ch = 0; ch < NCH; ch += 2) {
        fetch(tile, ch + 1, bv[1]);
        products(ch, bv[0]);
        if (ch + 2 < NCH)
          fetch(tile, ch + 2, bv[0]);
        else if (tile + stride < total)
          fetch(tile + stride, 0, bv[0]);
        products(ch + 1, bv[1]);
      }
    }
    float *yb = y + (size_t)b * M * L + l0;     // uniform
    const int so = 4 * kh * (int)L + 2 * col;  // lane part of the D row (r & 3) + 8 (r >> 2) + 4 kh, column pair 2 col
#pragma unroll
    for (int m = 0; m < M / 32; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        *reinterpret_cast<float2 *>(yb + (size_t)(m * 32 + (r & 3) + 8 * (r >> 2)) * L + so) = make_float2(acc[m][0][r], acc[m][1][r]);
  }
}

// ---------------------------------------------------------------------------------------------------------------- weight gradient
constexpr int WG_LD = 66;  // LDS row stride in floats

// CO: channels of dy (multiple of 32), CIP: channels of x rounded up to 32.  part[wg][CO][CIP] = this workgroup's share of dW.
template <int CO, int CIP>
__global__ __launch_bounds__(256, 2) void conv1x1_wgrad_kernel(const float *__restrict__ dy, const float *__restrict__ x, int cin, int Bn, long L,
                                                               float *__restrict__ part) {
  constexpr int ROWS = CO + CIP, NT = CIP / 32, PAIRS = (CO / 32) * NT, PPW = (PAIRS + 3) / 4, NLD = ROWS * 16 / 256;
  static_assert(ROWS * 16 % 256 == 0, "rows x 16 float4 must fill whole rounds of the workgroup");
  __shared__ float tile[ROWS * WG_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), col = lane & 31, kh = lane >> 5;
  const long chunks_per_slab = L >> 6, total = (long)Bn * chunks_per_slab;
  const long per = (total + gridDim.x - 1) / gridDim.x, c0 = (long)blockIdx.x * per, c1 = min(total, c0 + per);
  f32x16 acc[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float4 st[NLD];
  auto fetch = [&](long c) {
    const long b = c / chunks_per_slab, l0 = (c - b * chunks_per_slab) << 6;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int e = i * 256 + tid, row = e >> 4, c4 = e & 15;
      if (row < CO)
        st[i] = *reinterpret_cast<const float4 *>(dy + ((size_t)b * CO + row) * L + l0 + 4 * c4);
      else if (row - CO < cin)
        st[i] = *reinterpret_cast<const float4 *>(x + ((size_t)b * cin + (row - CO)) * L + l0 + 4 * c4);
      else
        st[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  if (c0 < c1) fetch(c0);
  for (long c = c0; c < c1; ++c) {
    __syncthreads();  // the previous chunk's fragment reads are done
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int e = i * 256 + tid, row = e >> 4, c4 = e & 15;
      float2 *d = reinterpret_cast<float2 *>(&tile[row * WG_LD + 4 * c4]);
      d[0] = make_float2(st[i].x, st[i].y);
      d[1] = make_float2(st[i].z, st[i].w);
    }
    __syncthreads();
    if (c + 1 < c1) fetch(c + 1);  // the next chunk's loads fly under this chunk's products
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int p = wave + 4 * i;
      if (p < PAIRS) {
        const int m = p / NT, n = p - m * NT;
        const float *ar = &tile[(m * 32 + col) * WG_LD + kh], *br = &tile[(CO + n * 32 + col) * WG_LD + kh];
#pragma unroll
        for (int s = 0; s < 32; ++s) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[2 * s], br[2 * s], acc[i], 0, 0, 0);
      }
    }
  }
  float *out = part + (size_t)blockIdx.x * CO * CIP;
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int p = wave + 4 * i;
    if (p < PAIRS) {
      const int m = p / NT, n = p - m * NT;
#pragma unroll
      for (int r = 0; r < 16; ++r) out[(size_t)(m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * CIP + n * 32 + col] = acc[i][r];
    }
  }
}

// dW[m][k] = sum over workgroups of part[wg][m][k] (double), k < cin.  64 outputs per workgroup, the partials of an output split over
// the four waves (wave q takes wg = q, q + 4, ...: four independent load streams per output instead of one chain of nwg dependent
// loads), the four sums combined in a fixed order: deterministic.
__global__ __launch_bounds__(256) void conv1x1_wgrad_reduce_kernel(const float *__restrict__ part, int nwg, int CO, int CIP, int cin,
                                                                   float *__restrict__ dw) {
  __shared__ double red[4][64];
  const int o = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + o;
  const bool ok = e < CO * cin;
  const int m = ok ? e / cin : 0, k = ok ? e - m * cin : 0;
  const float *p = part + (size_t)m * CIP + k;
  const size_t stride = (size_t)CO * CIP;
  double s0 = 0.0, s1 = 0.0;
  int g = q;
  for (; g + 4 < nwg; g += 8) {
    s0 += (double)p[(size_t)g * stride];
    s1 += (double)p[(size_t)(g + 4) * stride];
  }
  if (g < nwg) s0 += (double)p[(size_t)g * stride];
  red[q][o] = s0 + s1;
  __syncthreads();
  if (q == 0 && ok) dw[e] = (float)((red[0][o] + red[1][o]) + (red[2][o] + red[3][o]));
}

// ---------------------------------------------------------------------------------------------------------------- nn.Linear weight gradient
// dW[n][k] = sum_r g[r][n] x[r][k] for row-major g (rows, N) and x (rows, K): the reduction runs along the ROWS and the channels are
// contiguous, i.e. both operands of v_mfma_f32_32x32x2_f32 have their lanes along contiguous memory as they are (A: lane = column of g,
// k-step = row; B: lane = column of x) -- no LDS, no transposes.  A workgroup owns a 128 x 128 tile of dW over one slice of the rows, a wave
// 64 x 64 of it (even / odd columns from one 8-byte load per lane); per-slice partials, reduced in double by conv1x1_wgrad_reduce_kernel.
__global__ __launch_bounds__(256, 2) void linear_wgrad_f32_kernel(const float *__restrict__ g, const float *__restrict__ x, long rows, int N, int K,
                                                                  float *__restrict__ part) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), col = lane & 31, kh = lane >> 5;
  const int tiles_k = K >> 7, tn = blockIdx.x / tiles_k, tk = blockIdx.x - tn * tiles_k;
  const int n0 = tn * 128 + (wave >> 1) * 64, k0 = tk * 128 + (wave & 1) * 64;
  long per = (rows + gridDim.y - 1) / gridDim.y;
  per = (per + 1) & ~1L;
  const long r0 = (long)blockIdx.y * per, r1 = min(rows, r0 + per);
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const float *gp = g + (size_t)(r0 + kh) * N + n0 + 2 * col, *xp = x + (size_t)(r0 + kh) * K + k0 + 2 * col;
  long r = r0;
#pragma unroll 1
  for (; r + 16 <= r1; r += 16) {  // eight row pairs per trip: sixteen loads in flight
    float2 a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a[u] = *reinterpret_cast<const float2 *>(gp + (size_t)(2 * u) * N);
      b[u] = *reinterpret_cast<const float2 *>(xp + (size_t)(2 * u) * K);
    }
    gp += (size_t)16 * N, xp += (size_t)16 * K;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, b[u].x, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, b[u].y, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, b[u].x, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, b[u].y, acc[1][1], 0, 0, 0);
    }
  }
  for (; r < r1; r += 2) {  // the last pairs; a row past the slice contributes zeros
    float2 a = make_float2(0.f, 0.f), b = make_float2(0.f, 0.f);
    if (r + kh < r1) a = *reinterpret_cast<const float2 *>(gp), b = *reinterpret_cast<const float2 *>(xp);
    gp += (size_t)2 * N, xp += (size_t)2 * K;
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.y, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.x, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[1][1], 0, 0, 0);
  }
  // D[m][n]: m = the A lane's column of g (n0 + 2 m + i), n = the B lane's column of x (k0 + 2 col + j)
  float *out = part + (size_t)blockIdx.y * N * K;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      const int m = (rr & 3) + 8 * (rr >> 2) + 4 * kh;
      *reinterpret_cast<float2 *>(out + (size_t)(n0 + 2 * m + i) * K + k0 + 2 * col) = make_float2(acc[i][0][rr], acc[i][1][rr]);
    }
}

static int conv_grid(long units, int per_block) {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) cus = 256;
  const long want = (units + per_block - 1) / per_block, cap = 2L * cus < 512 ? 2L * cus : 512;  // (512: unopose_conv1x1_train_wgrad_blocks)
  return (int)(want < cap ? (want < 1 ? 1 : want) : cap);
}

}  // namespace unopose

using namespace unopose;

extern "C" {

int unopose_conv1x1_train_wgrad_blocks(void) { return 512; }  // upper bound of the workgroup count (workspace = blocks * cout * 128 floats)

int unopose_conv1x1_train_forward(const float *x, int B, int cin, long L, const float *w, int cout, float *y, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(x && w && y, "conv1x1_train_forward: null pointer");
  UNOPOSE_REQUIRE(B >= 1 && L >= 64 && L % 64 == 0 && L < (1L << 28), "conv1x1_train_forward: slab length must be a multiple of 64 below 2^28 (B=%d L=%ld)", B, L);
  hipStream_t s = (hipStream_t)stream;
  const int grid = conv_grid((long)B * (L >> 6), 4);
#define UNOPOSE_CONV_CASE(KP, MM)                                                                                   \
  if (cin <= KP && cout == MM) {                                                                                    \
    hipLaunchKernelGGL((conv1x1_f32_kernel<KP, MM>), dim3(grid), dim3(256), 0, s, x, w, cin, B, L, y);              \
    return check_launch("conv1x1_train_forward");                                                                   \
  }
  // (per output width the tightest K first: a 32 -> 32 layer must not run zero-padded to K = 64)
  UNOPOSE_CONV_CASE(8, 32)
  UNOPOSE_CONV_CASE(32, 32)
  UNOPOSE_CONV_CASE(64, 32)
  UNOPOSE_CONV_CASE(32, 64)
  UNOPOSE_CONV_CASE(64, 64)
  UNOPOSE_CONV_CASE(128, 64)
  UNOPOSE_CONV_CASE(64, 128)
  UNOPOSE_CONV_CASE(128, 128)
#undef UNOPOSE_CONV_CASE
  UNOPOSE_REQUIRE(false, "conv1x1_train_forward: no kernel for %d -> %d channels (built: <=8->32, <=32->{32,64}, <=64->{32,64,128}, <=128->{64,128})", cin,
                  cout);
  return UNOPOSE_EINVAL;
}

int unopose_conv1x1_train_wgrad(const float *dy, const float *x, int B, int cout, int cin, long L, float *workspace, float *dw,
                                unopose_stream_t stream) {
  UNOPOSE_REQUIRE(dy && x && workspace && dw, "conv1x1_train_wgrad: null pointer");
  UNOPOSE_REQUIRE(B >= 1 && L >= 64 && L % 64 == 0 && L < (1L << 28), "conv1x1_train_wgrad: slab length must be a multiple of 64 below 2^28 (B=%d L=%ld)", B, L);
  hipStream_t s = (hipStream_t)stream;
  const int grid = conv_grid((long)B * (L >> 6), 8);
#define UNOPOSE_WGRAD_CASE(CO, CIP)                                                                                                      \
  if (cout == CO && cin <= CIP) {                                                                                                        \
    hipLaunchKernelGGL((conv1x1_wgrad_kernel<CO, CIP>), dim3(grid), dim3(256), 0, s, dy, x, cin, B, L, workspace);                        \
    if (check_launch("conv1x1_train_wgrad")) return UNOPOSE_ELAUNCH;                                                                     \
    hipLaunchKernelGGL(conv1x1_wgrad_reduce_kernel, dim3(cdiv(CO * cin, 64)), dim3(256), 0, s, (const float *)workspace, grid, CO, CIP, cin, dw); \
    return check_launch("conv1x1_train_wgrad_reduce");                                                                                   \
  }
  UNOPOSE_WGRAD_CASE(32, 32)
  UNOPOSE_WGRAD_CASE(64, 32)
  UNOPOSE_WGRAD_CASE(64, 64)
  UNOPOSE_WGRAD_CASE(128, 64)
  UNOPOSE_WGRAD_CASE(128, 128)
#undef UNOPOSE_WGRAD_CASE
  UNOPOSE_REQUIRE(false, "conv1x1_train_wgrad: no kernel for %d x %d weights", cout, cin);
  return UNOPOSE_EINVAL;
}

int unopose_linear_wgrad_f32_splits(long rows, int N, int K) {  // row slices = workspace / (N * K) floats
  const long tiles = (long)(N >> 7) * (K >> 7);
  long sp = tiles > 0 ? (1024 + tiles - 1) / tiles : 1;  // about four workgroups per CU
  const long cap = rows / 512 > 0 ? rows / 512 : 1;  // at least 512 rows per slice
  return (int)(sp < cap ? (sp < 1 ? 1 : sp) : cap);
}

int unopose_linear_wgrad_f32(const float *g, const float *x, long rows, int N, int K, float *workspace, float *dw, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(g && x && workspace && dw, "linear_wgrad_f32: null pointer");
  UNOPOSE_REQUIRE(rows >= 1 && N >= 128 && K >= 128 && N % 128 == 0 && K % 128 == 0 && (long)N * K < (1L << 31),
                  "linear_wgrad_f32: features must be multiples of 128 (rows=%ld N=%d K=%d)", rows, N, K);
  hipStream_t s = (hipStream_t)stream;
  const int splits = unopose_linear_wgrad_f32_splits(rows, N, K);
  hipLaunchKernelGGL(linear_wgrad_f32_kernel, dim3((N >> 7) * (K >> 7), splits), dim3(256), 0, s, g, x, rows, N, K, workspace);
  if (check_launch("linear_wgrad_f32")) return UNOPOSE_ELAUNCH;
  hipLaunchKernelGGL(conv1x1_wgrad_reduce_kernel, dim3(cdiv(N * K, 64)), dim3(256), 0, s, (const float *)workspace, splits, N, K, K, dw);
  return check_launch("linear_wgrad_f32_reduce");
}

}  // extern "C"
