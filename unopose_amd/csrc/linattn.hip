// Focused linear attention core of the dense (2048-token) layers for gfx950 (C ABI part 2).
//
// Replaces the element-wise chain + two einsums of LinearAttention.forward
// (core/unopose/model/transformer.py:533-568) on the dense side:
//   q = relu(q) + 1e-6; q /= softplus(scale); q <- q^3 * |q| / |q^3|      ("focusing", per token)
//   z = 1 / (q_h . sum_j k_j,h + 1e-6);  x_h = (q_h kv_h) * z            (per head, kv_h = k_h^T v_h, 64x64)
// The reference makes ~12 passes over the (B,2048,256) tensor; here one wavefront owns 32 tokens, does
// the focusing in registers in MFMA A-operand layout (so the result feeds v_mfma_f32_32x32x16_bf16
// directly), contracts with the 64x64 kv of each head and scales by z in the epilogue.
// mode 1 writes only the focused features (used for the 196 sparse keys, whose kv / k-sum are tiny
// torch contractions).
#include "common.h"

namespace unopose {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
__device__ __forceinline__ u16 la_f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (u16)(u >> 16);
}
__device__ __forceinline__ float la_bf2f(u16 h) { return __uint_as_float(((uint32_t)h) << 16); }

// The focusing of one token row held in MFMA A-operand layout: the lane pair (lane, lane ^ 32) owns the row, lane half `hb` the 8-channel
// runs [16 ks + 8 hb, +8).  q <- ((relu(x) + 1e-6) / softplus(scale))^3, returns |q_before| / |q_after| (the row's rescale factor).
__device__ __forceinline__ float la_focus(const u16 *xr, const float *__restrict__ inv_sp, int hb, float (&q)[16][8]) {
  float n1 = 0.f;
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    union { bf16x8 v; u16 u[8]; } f;
    f.v = *reinterpret_cast<const bf16x8 *>(xr + ks * 16);
    const float4 s0 = *reinterpret_cast<const float4 *>(inv_sp + ks * 16 + hb * 8);
    const float4 s1 = *reinterpret_cast<const float4 *>(inv_sp + ks * 16 + hb * 8 + 4);
    const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = (fmaxf(la_bf2f(f.u[e]), 0.f) + 1e-6f) * sc[e];
      q[ks][e] = v;
      n1 += v * v;
    }
  }
  n1 += __shfl_xor(n1, 32);
  float n3 = 0.f;
#pragma unroll
  for (int ks = 0; ks < 16; ++ks)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float p = q[ks][e];
      p = p * p * p;  // focusing_factor = 3 (the only value the kernel is built for; checked by the host)
      q[ks][e] = p;
      n3 += p * p;
    }
  n3 += __shfl_xor(n3, 32);
  return sqrtf(n1) / sqrtf(n3);
}

// x: (B,N,256) bf16 projected q (or k); inv_sp: (256) 1/softplus(scale); kvt: (B,4,64 d,64 c) bf16;
// ksum: (B,256) fp32; out (B,N,256) bf16.
template <int MODE>
__global__ __launch_bounds__(256) void linear_attn_kernel(const u16 *__restrict__ x, const float *__restrict__ inv_sp,
                                                          const u16 *__restrict__ kvt, const float *__restrict__ ksum,
                                                          int N, int focus, u16 *__restrict__ out) {
  __shared__ float zl[4][32][4];
  // per-wave I/O tile: the wave's 32 token rows (16 KiB, contiguous in memory) are moved with fully
  // coalesced 16-byte accesses and re-read / re-written in MFMA fragment order through LDS
  __shared__ __attribute__((aligned(16))) u16 tile[4][32][256 + 8];
  const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t0 = (blockIdx.x * 4 + wave) * 32;
  if (t0 >= N) return;
  const int col = lane & 31, hb = lane >> 5;
  {
    const u16 *src = x + ((size_t)b * N + t0) * 256;
    const int nrows = min(32, N - t0);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int e = i * 64 + lane, r = e >> 5, c8 = e & 31;  // row, 16-byte column
      uint4 v = make_uint4(0, 0, 0, 0);
      if (r < nrows) v = *reinterpret_cast<const uint4 *>(src + (size_t)r * 256 + c8 * 8);
      *reinterpret_cast<uint4 *>(&tile[wave][r][c8 * 8]) = v;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const u16 *xr = &tile[wave][col][hb * 8];
  float q[16][8];
  const float fac = la_focus(xr, inv_sp, hb, q);
  bf16x8 qa[16];
  float zp[4] = {0.f, 0.f, 0.f, 0.f};
  const float *ks_b = ksum + (size_t)b * 256 + hb * 8;
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    union { bf16x8 v; u16 u[8]; } f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = q[ks][e] * fac;
      f.u[e] = la_f2bf(v);
      if (MODE == 0) zp[ks >> 2] += v * ks_b[ks * 16 + e];
    }
    qa[ks] = f.v;
  }
  auto store_tile = [&]() {  // tile[wave] -> out rows t0.. (coalesced 16-byte stores)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    u16 *dst = out + ((size_t)b * N + t0) * 256;
    const int nrows = min(32, N - t0);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int e = i * 64 + lane, r = e >> 5, c8 = e & 31;
      if (r < nrows)
        *reinterpret_cast<uint4 *>(dst + (size_t)r * 256 + c8 * 8) = *reinterpret_cast<const uint4 *>(&tile[wave][r][c8 * 8]);
    }
  };
  if (MODE == 1) {  // focused features only
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) *reinterpret_cast<bf16x8 *>(&tile[wave][col][ks * 16 + hb * 8]) = qa[ks];
    store_tile();
    return;
  }
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    zp[h] += __shfl_xor(zp[h], 32);
    if (hb == 0) zl[wave][col][h] = 1.f / (zp[h] + 1e-6f);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // x_h = q_h kv_h : A = q (rows = tokens), B = kv (B[k=c][n=d] read from kvt[d][c])
#pragma unroll
  for (int h = 0; h < 4; ++h) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const u16 *kp = kvt + (((size_t)b * 4 + h) * 64 + nt * 32 + col) * 64 + hb * 8;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 bv = *reinterpret_cast<const bf16x8 *>(kp + ks * 16);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[h * 4 + ks], bv, acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hb;
        tile[wave][row][h * 64 + nt * 32 + col] = la_f2bf(acc[r] * zl[wave][row][h]);
      }
    }
  }
  store_tile();
}

// The key / value state of one layer in ONE launch: ykv (B,J,512) bf16 = [k projection | v] as the fused projection GEMM leaves it ->
// kvt (B,4,64 d,64 c) bf16 = sum_j v[j,h,d] kf[j,h,c] and ksum (B,256) fp32 = sum_j kf[j], kf = the focused keys rounded to bf16 (the
// values the MODE 1 kernel writes; here they never leave LDS).  One workgroup per pair; rounds of 128 tokens: wave w focuses rows
// [32 w, +32) of the round in place in LDS, then wave h contracts head h over the round's tokens on the matrix cores (k = tokens: the
// operands are gathered down LDS columns, 8 two-byte reads per fragment -- 100 KB per pair, latency does not matter here).
// A pair's rows are [j0, j0 + J) of its Jrows rows in memory (the sparse-to-dense block projects the background-token row along
// with the 196 tokens rather than copying them out: j0 = 1).
__global__ __launch_bounds__(256) void linear_attn_kv_state_kernel(const u16 *__restrict__ ykv, const float *__restrict__ inv_sp, int J,
                                                                   int Jrows, int j0, u16 *__restrict__ kvt, float *__restrict__ ksum) {
  __shared__ __attribute__((aligned(16))) u16 kf_s[128][256 + 8];
  __shared__ __attribute__((aligned(16))) u16 v_s[128][256 + 8];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, hb = lane >> 5;
  f32x16 acc[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
  float csum = 0.f;  // this thread's channel (= tid) of ksum
  for (int r0 = 0; r0 < J; r0 += 128) {
    {  // phase A: wave's 32 rows of the round
      const int t0 = r0 + wave * 32;
      const int nrows = max(0, min(32, J - t0));
      const u16 *src = ykv + ((size_t)b * Jrows + j0 + t0) * 512;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int e = i * 64 + lane, r = e >> 5, c8 = e & 31;  // row, 16-byte column
        uint4 kq = make_uint4(0, 0, 0, 0), vq = make_uint4(0, 0, 0, 0);
        if (r < nrows) {
          kq = *reinterpret_cast<const uint4 *>(src + (size_t)r * 512 + c8 * 8);
          vq = *reinterpret_cast<const uint4 *>(src + (size_t)r * 512 + 256 + c8 * 8);
        }
        *reinterpret_cast<uint4 *>(&kf_s[wave * 32 + r][c8 * 8]) = kq;
        *reinterpret_cast<uint4 *>(&v_s[wave * 32 + r][c8 * 8]) = vq;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      u16 *xr = &kf_s[wave * 32 + col][hb * 8];
      float q[16][8];
      const float fac = la_focus(xr, inv_sp, hb, q);
      const bool live = col < nrows;  // rows past the cloud contribute nothing (their focusing is 0 / 0)
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        union { bf16x8 v; u16 u[8]; } f;
#pragma unroll
        for (int e = 0; e < 8; ++e) f.u[e] = live ? la_f2bf(q[ks][e] * fac) : (u16)0;
        *reinterpret_cast<bf16x8 *>(xr + ks * 16) = f.v;  // the positions this lane read: no other lane touches them
      }
    }
    __syncthreads();
    const int valid = min(128, J - r0);
#pragma unroll 8
    for (int t = 0; t < valid; ++t) csum += la_bf2f(kf_s[t][tid]);
    const int nk = (valid + 15) >> 4;
#pragma unroll 2
    for (int kk = 0; kk < nk; ++kk) {
      bf16x8 av[2], bv[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        union { bf16x8 v; u16 u[8]; } fa, fb;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          fa.u[e] = v_s[kk * 16 + hb * 8 + e][wave * 64 + i * 32 + col];
          fb.u[e] = kf_s[kk * 16 + hb * 8 + e][wave * 64 + i * 32 + col];
        }
        av[i] = fa.v;
        bv[i] = fb.v;
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[mt], bv[nt], acc[mt][nt], 0, 0, 0);
    }
    __syncthreads();
  }
  ksum[(size_t)b * 256 + tid] = csum;
  u16 *dst = kvt + ((size_t)b * 4 + wave) * 64 * 64;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hb;
        dst[d * 64 + nt * 32 + col] = la_f2bf(acc[mt][nt][r]);
      }
}

// ---- fp32-class variant (the reference's default precision: no autocast) ------------------------------------
// Same scheme on fp32 data: focusing in fp32 registers, the per-head 64x64 contraction as hi/lo-split bf16
// MFMAs (x = xh + xl, 3 MFMAs per product, fp32 accumulation: ~2^-16 relative error, as csrc/attn_f32.hip),
// fp32 in / out.  I/O goes straight between global memory and fragment registers (32-byte pieces per
// lane): this is the parity configuration, not the benched one.
__device__ __forceinline__ void la_split(const float (&v)[8], bf16x8 &hi, bf16x8 &lo) {
  union { bf16x8 v; uint32_t u[4]; } h, l;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    h.u[e] = cvt_pk_bf16_f32(v[2 * e], v[2 * e + 1]);
    const float r0 = v[2 * e] - __uint_as_float(h.u[e] << 16);
    const float r1 = v[2 * e + 1] - __uint_as_float(h.u[e] & 0xFFFF0000u);
    l.u[e] = cvt_pk_bf16_f32(r0, r1);
  }
  hi = h.v;
  lo = l.v;
}

// x: (B,N,256) fp32 projected q (or k); kvt: (B,4,64 d,64 c) fp32; ksum: (B,256) fp32; out (B,N,256) fp32.
template <int MODE>
__global__ __launch_bounds__(256) void linear_attn_f32_kernel(const float *__restrict__ x, const float *__restrict__ inv_sp,
                                                              const float *__restrict__ kvt, const float *__restrict__ ksum,
                                                              int N, float *__restrict__ out) {
  const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t0 = (blockIdx.x * 4 + wave) * 32;
  if (t0 >= N) return;
  const int col = lane & 31, hb = lane >> 5;
  const int trow = min(t0 + col, N - 1);  // ragged tail: clamp the load, mask the store
  const float *xr = x + ((size_t)b * N + trow) * 256 + hb * 8;
  float q[16][8];
  float n1 = 0.f;
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    const float4 a0 = *reinterpret_cast<const float4 *>(xr + ks * 16);
    const float4 a1 = *reinterpret_cast<const float4 *>(xr + ks * 16 + 4);
    const float4 s0 = *reinterpret_cast<const float4 *>(inv_sp + ks * 16 + hb * 8);
    const float4 s1 = *reinterpret_cast<const float4 *>(inv_sp + ks * 16 + hb * 8 + 4);
    const float xv[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = (fmaxf(xv[e], 0.f) + 1e-6f) * sc[e];
      q[ks][e] = v;
      n1 += v * v;
    }
  }
  n1 += __shfl_xor(n1, 32);
  float n3 = 0.f;
#pragma unroll
  for (int ks = 0; ks < 16; ++ks)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float p = q[ks][e];
      p = p * p * p;
      q[ks][e] = p;
      n3 += p * p;
    }
  n3 += __shfl_xor(n3, 32);
  const float fac = sqrtf(n1) / sqrtf(n3);
  float zp[4] = {0.f, 0.f, 0.f, 0.f};
  const float *ks_b = MODE == 0 ? ksum + (size_t)b * 256 + hb * 8 : nullptr;
#pragma unroll
  for (int ks = 0; ks < 16; ++ks)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      q[ks][e] *= fac;
      if (MODE == 0) zp[ks >> 2] += q[ks][e] * ks_b[ks * 16 + e];
    }
  if (MODE == 1) {  // focused features only
    if (t0 + col < N) {
      float *dst = out + ((size_t)b * N + t0 + col) * 256 + hb * 8;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        *reinterpret_cast<float4 *>(dst + ks * 16) = make_float4(q[ks][0], q[ks][1], q[ks][2], q[ks][3]);
        *reinterpret_cast<float4 *>(dst + ks * 16 + 4) = make_float4(q[ks][4], q[ks][5], q[ks][6], q[ks][7]);
      }
    }
    return;
  }
  __shared__ float zl[4][32][4];
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    zp[h] += __shfl_xor(zp[h], 32);
    if (hb == 0) zl[wave][col][h] = 1.f / (zp[h] + 1e-6f);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    bf16x8 qh[4], ql[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) la_split(q[h * 4 + ks], qh[ks], ql[ks]);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float *kp = kvt + (((size_t)b * 4 + h) * 64 + nt * 32 + col) * 64 + hb * 8;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const float4 b0 = *reinterpret_cast<const float4 *>(kp + ks * 16);
        const float4 b1 = *reinterpret_cast<const float4 *>(kp + ks * 16 + 4);
        const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        bf16x8 bh, bl;
        la_split(bv, bh, bl);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ql[ks], bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qh[ks], bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qh[ks], bh, acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hb;
        if (t0 + row < N) out[((size_t)b * N + t0 + row) * 256 + h * 64 + nt * 32 + col] = acc[r] * zl[wave][row][h];
      }
    }
  }
}

}  // namespace unopose

using namespace unopose;

extern "C" {

int unopose_linear_attention(const void *x, const float *inv_softplus_scale, const void *kvt, const float *ksum, int B,
                             int N, int focus, int mode, void *out, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(x && inv_softplus_scale && out && (mode == 1 || (kvt && ksum)), "linear_attention: null pointer");
  UNOPOSE_REQUIRE(B >= 0 && N >= 1 && B <= 65535, "linear_attention: bad sizes");
  UNOPOSE_REQUIRE(focus == 3, "linear_attention: built for focusing_factor = 3 (got %d)", focus);
  if (B == 0) return UNOPOSE_OK;
  dim3 grid(cdiv(N, 128), B);
  hipStream_t s = (hipStream_t)stream;
  if (mode == 1)
    hipLaunchKernelGGL(linear_attn_kernel<1>, grid, dim3(256), 0, s, (const u16 *)x, inv_softplus_scale,
                       (const u16 *)nullptr, (const float *)nullptr, N, focus, (u16 *)out);
  else
    hipLaunchKernelGGL(linear_attn_kernel<0>, grid, dim3(256), 0, s, (const u16 *)x, inv_softplus_scale,
                       (const u16 *)kvt, ksum, N, focus, (u16 *)out);
  return check_launch("linear_attention");
}

int unopose_linear_attention_kv_state(const void *ykv, const float *inv_softplus_scale, int B, int J, int rows_per_pair, int first_row,
                                      int focus, void *kvt, float *ksum, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(ykv && inv_softplus_scale && kvt && ksum, "linear_attention_kv_state: null pointer");
  UNOPOSE_REQUIRE(B >= 0 && J >= 1 && first_row >= 0 && rows_per_pair >= first_row + J, "linear_attention_kv_state: bad sizes");
  UNOPOSE_REQUIRE(focus == 3, "linear_attention_kv_state: built for focusing_factor = 3 (got %d)", focus);
  if (B == 0) return UNOPOSE_OK;
  hipLaunchKernelGGL(linear_attn_kv_state_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, (const u16 *)ykv, inv_softplus_scale, J,
                     rows_per_pair, first_row, (u16 *)kvt, ksum);
  return check_launch("linear_attention_kv_state");
}

int unopose_linear_attention_f32(const float *x, const float *inv_softplus_scale, const float *kvt, const float *ksum, int B,
                                 int N, int focus, int mode, float *out, unopose_stream_t stream) {
  UNOPOSE_REQUIRE(x && inv_softplus_scale && out && (mode == 1 || (kvt && ksum)), "linear_attention_f32: null pointer");
  UNOPOSE_REQUIRE(B >= 0 && N >= 1 && B <= 65535, "linear_attention_f32: bad sizes");
  UNOPOSE_REQUIRE(focus == 3, "linear_attention_f32: built for focusing_factor = 3 (got %d)", focus);
  if (B == 0) return UNOPOSE_OK;
  dim3 grid(cdiv(N, 128), B);
  hipStream_t s = (hipStream_t)stream;
  if (mode == 1)
    hipLaunchKernelGGL(linear_attn_f32_kernel<1>, grid, dim3(256), 0, s, x, inv_softplus_scale, (const float *)nullptr,
                       (const float *)nullptr, N, out);
  else
    hipLaunchKernelGGL(linear_attn_f32_kernel<0>, grid, dim3(256), 0, s, x, inv_softplus_scale, kvt, ksum, N, out);
  return check_launch("linear_attention_f32");
}

}  // extern "C"
