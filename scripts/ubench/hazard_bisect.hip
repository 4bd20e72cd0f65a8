// Which instruction pattern of the frame kernels changes results beside an MFMA-only neighbour?  (DESIGN.md section 7, round 3:
// scripts/ubench/coresidency_matrix.py showed query_lrf_group differing in 16 of 20 runs beside agg_mfma -- a kernel that touches
// neither LDS nor memory.)  Each micro-victim below isolates ONE pattern of csrc/geom.hip; all are deterministic functions of
// (blockIdx, lane, iteration), so any run-to-run difference is a fault.  Driven by scripts/ubench/hazard_bisect.py.
//   0 vote        v_cmp -> SGPR pair -> s_bcnt1 -> scalar accumulate (the sign vote)
//   1 wavesum     DPP row_shr adds + 4 v_readlane + 3 adds (wave_sum_f32)
//   2 division    IEEE fp32 division chain (v_div_scale / v_div_fmas / v_div_fixup, denorm-mode s_setreg around it)
//   3 sqrt        sqrtf chain
//   4 ldslist     ballot + mbcnt compaction into a per-wave LDS list, wave_barrier, cross-lane read-back
//   5 eig         eig_sym3 (cyclic Jacobi, csrc/jacobi3.h) on per-lane matrices
//   6 readlane    v_readlane of a just-written VGPR -> v_mov back -> add
//   7 cmpmask     v_cmp -> SGPR mask -> v_cndmask with that mask (VALU reads a VALU-written SGPR)
#include "../../unopose_amd/csrc/common.h"
#include "../../unopose_amd/csrc/jacobi3.h"
using namespace unopose;

__device__ __forceinline__ float seedf(int i, int lane, int blk) { return (float)(((i * 2654435761u) ^ (lane * 40503u) ^ (blk * 9973u)) & 0xffff) * (1.f / 65536.f) - 0.5f; }

__global__ __launch_bounds__(256) void hv_vote(float *out, int iters) {
  const int lane = threadIdx.x & 63;
  int vote = 0;
  for (int i = 0; i < iters; ++i) {
    const float pr = seedf(i, lane, blockIdx.x);
    vote += __builtin_popcountll(__ballot(pr > 1e-3f)) - __builtin_popcountll(__ballot(pr < -1e-3f));
  }
  out[blockIdx.x * 256 + threadIdx.x] = (float)vote;
}
__global__ __launch_bounds__(256) void hv_wavesum(float *out, int iters) {
  const int lane = threadIdx.x & 63;
  float acc = 0.f;
  for (int i = 0; i < iters; ++i) acc = acc * 0.5f + wave_sum_f32(seedf(i, lane, blockIdx.x));
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void hv_division(float *out, int iters) {
  const int lane = threadIdx.x & 63;
  float acc = 1.f;
  for (int i = 0; i < iters; ++i) acc = (acc + seedf(i, lane, blockIdx.x)) / (1.5f + seedf(i + 7, lane, blockIdx.x));
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void hv_sqrt(float *out, int iters) {
  const int lane = threadIdx.x & 63;
  float acc = 1.f;
  for (int i = 0; i < iters; ++i) acc = sqrtf(acc + 1.f + seedf(i, lane, blockIdx.x));
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void hv_ldslist(float *out, int iters) {
  __shared__ int list[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int acc = 0;
  for (int i = 0; i < iters; ++i) {
    const bool hit = seedf(i, lane, blockIdx.x) > 0.f;
    const unsigned long long mask = __ballot(hit);
    const int pre = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    const int cnt = __builtin_popcountll(mask);
    if (hit) list[wave][pre] = lane + i;
    for (int l = cnt + lane; l < 64; l += 64) list[wave][l] = -1;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    acc = acc * 3 + list[wave][(lane * 7 + i) & 63];
    __builtin_amdgcn_wave_barrier();
  }
  out[blockIdx.x * 256 + threadIdx.x] = (float)(acc & 0xfffff);
}
__global__ __launch_bounds__(256) void hv_eig(float *out, int iters) {
  const int lane = threadIdx.x & 63;
  float acc = 0.f;
  for (int i = 0; i < iters; ++i) {
    const float a = seedf(i, lane, blockIdx.x), b = seedf(i + 1, lane, blockIdx.x), c = seedf(i + 2, lane, blockIdx.x);
    Vec3 e0, e1, e2;
    float l0, l1, l2;
    eig_sym3(1.f + a * a, a * b, a * c, 1.f + b * b, b * c, 1.f + c * c, e0, e1, e2, l0, l1, l2);
    acc += l0 + 2.f * l1 + 3.f * l2 + e2.x;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void hv_readlane(float *out, int iters) {
  const int lane = threadIdx.x & 63;
  float acc = 0.f;
  for (int i = 0; i < iters; ++i) {
    const float v = seedf(i, lane, blockIdx.x) + acc * 0.25f;
    const float r = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 47));
    acc = v + r;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void hv_cmpmask(float *out, int iters) {
  const int lane = threadIdx.x & 63;
  float acc = 0.f;
  for (int i = 0; i < iters; ++i) {
    const float a = seedf(i, lane, blockIdx.x), b = seedf(i + 3, lane, blockIdx.x);
    acc += (a > b) ? a * 1.5f : b * 0.75f;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
// 8 pkf32: packed fp32 multiply / add (v_pk_mul_f32, v_pk_add_f32) feeding ordinary VALU ops, the pattern -O3's SLP vectoriser
// puts all over the frame kernels (97 packed ops in query_lrf_group; none at -O1, which is bit-stable beside MFMA neighbours)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void hv_pkf32(float *out, int iters) {
  const int lane = threadIdx.x & 63;
  f32x2 acc = {0.f, 1.f};
  float s = 0.f;
  for (int i = 0; i < iters; ++i) {
    const f32x2 a = {seedf(i, lane, blockIdx.x), seedf(i + 1, lane, blockIdx.x)};
    const f32x2 b = {seedf(i + 2, lane, blockIdx.x), seedf(i + 3, lane, blockIdx.x)};
    f32x2 p = a * b;            // v_pk_mul_f32
    p = p + acc * 0.5f;         // v_pk_mul_f32 + v_pk_add_f32
    s = s * 0.25f + (p.x - p.y);  // scalar VALU consumer right behind the packed result
    acc = p;
  }
  out[blockIdx.x * 256 + threadIdx.x] = s + acc.x;
}
// 9 pkmul_opsel: the minimal reproducer asm_var.py arrived at -- v_pk_mul_f32 with an op_sel modifier (the lo half reads the HI
// element of a source pair).  out[..] = number of iterations in which the packed result differs from the same two products
// computed with v_mul_f32 (self-checking: no reference run needed; must be 0).
__global__ __launch_bounds__(256) void hv_pkmul_opsel(float *out, int iters) {
  const int lane = threadIdx.x & 63;
  int bad = 0;
  float carry = 0.f;
  for (int i = 0; i < iters; ++i) {
    f32x2 a = {seedf(i, lane, blockIdx.x) + carry, seedf(i + 1, lane, blockIdx.x)};
    f32x2 b = {seedf(i + 2, lane, blockIdx.x), seedf(i + 3, lane, blockIdx.x) - carry};
    f32x2 p;
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(p) : "v"(a), "v"(b));  // p.lo = a.lo * b.hi, p.hi = a.hi * b.lo
    float w0, w1;
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(w0) : "v"(a.x), "v"(b.y));
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(w1) : "v"(a.y), "v"(b.x));
    bad += (__float_as_uint(p.x) != __float_as_uint(w0)) || (__float_as_uint(p.y) != __float_as_uint(w1));
    carry = w0 * 0.125f;
  }
  out[blockIdx.x * 256 + threadIdx.x] = (float)bad;
}
extern "C" int hazard_victim_launch(int which, void *out, int blocks, int iters, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  float *o = (float *)out;
  switch (which) {
    case 0: hipLaunchKernelGGL(hv_vote, dim3(blocks), dim3(256), 0, s, o, iters); break;
    case 1: hipLaunchKernelGGL(hv_wavesum, dim3(blocks), dim3(256), 0, s, o, iters); break;
    case 2: hipLaunchKernelGGL(hv_division, dim3(blocks), dim3(256), 0, s, o, iters); break;
    case 3: hipLaunchKernelGGL(hv_sqrt, dim3(blocks), dim3(256), 0, s, o, iters); break;
    case 4: hipLaunchKernelGGL(hv_ldslist, dim3(blocks), dim3(256), 0, s, o, iters); break;
    case 5: hipLaunchKernelGGL(hv_eig, dim3(blocks), dim3(256), 0, s, o, iters); break;
    case 6: hipLaunchKernelGGL(hv_readlane, dim3(blocks), dim3(256), 0, s, o, iters); break;
    case 9: hipLaunchKernelGGL(hv_pkmul_opsel, dim3(blocks), dim3(256), 0, s, o, iters); break;
    case 8: hipLaunchKernelGGL(hv_pkf32, dim3(blocks), dim3(256), 0, s, o, iters); break;
    default: hipLaunchKernelGGL(hv_cmpmask, dim3(blocks), dim3(256), 0, s, o, iters); break;
  }
  return (int)hipGetLastError();
}
namespace unopose { void set_error(const char *, ...) {} }
