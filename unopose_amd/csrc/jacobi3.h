// 3x3 Jacobi eigen / SVD in registers (fp32), one problem per calling lane.
// Fixed sweep counts -> uniform control flow across a wavefront.
#pragma once
#include "common.h"

namespace unopose {

struct Vec3 {
  float x, y, z;
};
__device__ __forceinline__ Vec3 v3(float x, float y, float z) { return Vec3{x, y, z}; }
__device__ __forceinline__ float dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ Vec3 cross(Vec3 a, Vec3 b) {
  return Vec3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ Vec3 scale(Vec3 a, float s) { return Vec3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ Vec3 sub(Vec3 a, Vec3 b) { return Vec3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ Vec3 add(Vec3 a, Vec3 b) { return Vec3{a.x + b.x, a.y + b.y, a.z + b.z}; }

// Givens pair (c, s) that annihilates the off-diagonal `apq` of [[app, apq],[apq, aqq]].
// Hardware reciprocal / sqrt / rsqrt (v_rcp_f32, v_sqrt_f32, v_rsq_f32: ~1 ulp, one instruction each)
// instead of the ~10-instruction IEEE sequences: a Jacobi rotation only has to be orthogonal to working
// precision -- (c, s) is renormalised by construction (c = rsq(1+t^2), s = t c) -- and the dependent
// chain of 18 rotations is the latency floor of every per-point frame.
__device__ __forceinline__ void sym_schur2(float app, float apq, float aqq, float &c, float &s) {
  if (fabsf(apq) > 1e-30f) {
    const float tau = (aqq - app) * __builtin_amdgcn_rcpf(2.f * apq);
    const float t = (tau >= 0.f ? 1.f : -1.f) * __builtin_amdgcn_rcpf(fabsf(tau) + __builtin_amdgcn_sqrtf(1.f + tau * tau));
    c = __builtin_amdgcn_rsqf(1.f + t * t);
    s = t * c;
  } else {
    c = 1.f;
    s = 0.f;
  }
}

// Eigen-decomposition of the symmetric matrix [[a00,a01,a02],[a01,a11,a12],[a02,a12,a22]]
// by cyclic Jacobi.  Returns the eigenvector of the SMALLEST eigenvalue (= last
// right-singular vector of a PSD matrix, what LRF needs: pointnet2_utils.py:446,
// model_utils.py:787) and optionally all three, sorted by descending eigenvalue.
__device__ __forceinline__ void eig_sym3(float a00, float a01, float a02, float a11, float a12, float a22, Vec3 &e0,
                                         Vec3 &e1, Vec3 &e2, float &l0, float &l1, float &l2) {
  // V = I (columns are eigenvectors)
  float v00 = 1, v01 = 0, v02 = 0, v10 = 0, v11 = 1, v12 = 0, v20 = 0, v21 = 0, v22 = 1;
#pragma unroll 1
  for (int sweep = 0; sweep < 6; ++sweep) {
    // converged when the off-diagonal mass is below fp32 resolution of the diagonal (typically after
    // 3-4 sweeps; every caller evaluates this on wave-uniform data, so the exit is not divergent)
    const float off = a01 * a01 + a02 * a02 + a12 * a12;
    const float dia = a00 * a00 + a11 * a11 + a22 * a22;
    if (off <= 1e-16f * dia) break;
    float c, s;
    // (p,q) = (0,1)
    sym_schur2(a00, a01, a11, c, s);
    {
      const float n00 = c * c * a00 - 2.f * s * c * a01 + s * s * a11;
      const float n11 = s * s * a00 + 2.f * s * c * a01 + c * c * a11;
      const float n02 = c * a02 - s * a12, n12 = s * a02 + c * a12;
      a00 = n00; a11 = n11; a01 = 0.f; a02 = n02; a12 = n12;
      float t;
      t = c * v00 - s * v01; v01 = s * v00 + c * v01; v00 = t;
      t = c * v10 - s * v11; v11 = s * v10 + c * v11; v10 = t;
      t = c * v20 - s * v21; v21 = s * v20 + c * v21; v20 = t;
    }
    // (0,2)
    sym_schur2(a00, a02, a22, c, s);
    {
      const float n00 = c * c * a00 - 2.f * s * c * a02 + s * s * a22;
      const float n22 = s * s * a00 + 2.f * s * c * a02 + c * c * a22;
      const float n01 = c * a01 - s * a12, n12 = s * a01 + c * a12;
      a00 = n00; a22 = n22; a02 = 0.f; a01 = n01; a12 = n12;
      float t;
      t = c * v00 - s * v02; v02 = s * v00 + c * v02; v00 = t;
      t = c * v10 - s * v12; v12 = s * v10 + c * v12; v10 = t;
      t = c * v20 - s * v22; v22 = s * v20 + c * v22; v20 = t;
    }
    // (1,2)
    sym_schur2(a11, a12, a22, c, s);
    {
      const float n11 = c * c * a11 - 2.f * s * c * a12 + s * s * a22;
      const float n22 = s * s * a11 + 2.f * s * c * a12 + c * c * a22;
      const float n01 = c * a01 - s * a02, n02 = s * a01 + c * a02;
      a11 = n11; a22 = n22; a12 = 0.f; a01 = n01; a02 = n02;
      float t;
      t = c * v01 - s * v02; v02 = s * v01 + c * v02; v01 = t;
      t = c * v11 - s * v12; v12 = s * v11 + c * v12; v11 = t;
      t = c * v21 - s * v22; v22 = s * v21 + c * v22; v21 = t;
    }
  }
  Vec3 c0 = v3(v00, v10, v20), c1 = v3(v01, v11, v21), c2 = v3(v02, v12, v22);
  float d0 = a00, d1 = a11, d2 = a22;
  // sort descending (3-element network)
  if (d0 < d1) { float t = d0; d0 = d1; d1 = t; Vec3 tv = c0; c0 = c1; c1 = tv; }
  if (d1 < d2) { float t = d1; d1 = d2; d2 = t; Vec3 tv = c1; c1 = c2; c2 = tv; }
  if (d0 < d1) { float t = d0; d0 = d1; d1 = t; Vec3 tv = c0; c0 = c1; c1 = tv; }
  e0 = c0; e1 = c1; e2 = c2;
  l0 = d0; l1 = d1; l2 = d2;
}

// Kabsch rotation from the 3x3 cross-covariance H (row-major h[i][j] = sum w s_i r_j):
//   R = V diag(1,1,sign det(V U^T)) U^T  with  H = U S V^T   (model_utils.py:729-734)
// computed by one-sided (Hestenes) Jacobi on the columns of H.  Equivalent closed
// form used here: R = v1 u1^T + v2 u2^T + (v1 x v2)(u1 x u2)^T, which needs only the
// two leading singular pairs, so rank-2 H (three-point hypotheses) is exact.
// Rank-1/0 H: the reference's answer is LAPACK-implementation-defined; we return a
// deterministic valid rotation (identity for H = 0).
__device__ __forceinline__ void kabsch_from_H(const float h[9], float R[9]) {
  // G = H (columns g0,g1,g2), V = I
  Vec3 g0 = v3(h[0], h[3], h[6]), g1 = v3(h[1], h[4], h[7]), g2 = v3(h[2], h[5], h[8]);
  Vec3 w0 = v3(1, 0, 0), w1 = v3(0, 1, 0), w2 = v3(0, 0, 1);
#define UNOPOSE_HJ_ROT(ga, gb, wa, wb)                                            \
  {                                                                               \
    const float alpha = dot(ga, ga), beta = dot(gb, gb), gamma = dot(ga, gb);     \
    if (fabsf(gamma) > 1e-12f * sqrtf(alpha * beta) && fabsf(gamma) > 1e-37f) {   \
      const float zeta = (beta - alpha) / (2.f * gamma);                          \
      const float t = (zeta >= 0.f ? 1.f : -1.f) / (fabsf(zeta) + sqrtf(1.f + zeta * zeta)); \
      const float c = 1.f / sqrtf(1.f + t * t), s = c * t;                        \
      Vec3 na = sub(scale(ga, c), scale(gb, s));                                  \
      gb = add(scale(ga, s), scale(gb, c));                                       \
      ga = na;                                                                    \
      Vec3 nw = sub(scale(wa, c), scale(wb, s));                                  \
      wb = add(scale(wa, s), scale(wb, c));                                       \
      wa = nw;                                                                    \
    }                                                                             \
  }
#pragma unroll 1
  for (int sweep = 0; sweep < 8; ++sweep) {
    UNOPOSE_HJ_ROT(g0, g1, w0, w1)
    UNOPOSE_HJ_ROT(g0, g2, w0, w2)
    UNOPOSE_HJ_ROT(g1, g2, w1, w2)
  }
#undef UNOPOSE_HJ_ROT
  float s0 = dot(g0, g0), s1 = dot(g1, g1), s2 = dot(g2, g2);
  if (s0 < s1) { float t = s0; s0 = s1; s1 = t; Vec3 tv = g0; g0 = g1; g1 = tv; tv = w0; w0 = w1; w1 = tv; }
  if (s1 < s2) { float t = s1; s1 = s2; s2 = t; Vec3 tv = g1; g1 = g2; g2 = tv; tv = w1; w1 = w2; w2 = tv; }
  if (s0 < s1) { float t = s0; s0 = s1; s1 = t; Vec3 tv = g0; g0 = g1; g1 = tv; tv = w0; w0 = w1; w1 = tv; }
  Vec3 u0, u1;
  if (s0 > 0.f) {
    u0 = scale(g0, 1.f / sqrtf(s0));
  } else {  // H == 0: U = V = I
    u0 = v3(1, 0, 0); w0 = v3(1, 0, 0); w1 = v3(0, 1, 0);
    g1 = v3(0, 0, 0);
  }
  Vec3 g1o = sub(g1, scale(u0, dot(g1, u0)));
  const float n1 = dot(g1o, g1o);
  if (n1 > 1e-24f * s0 && n1 > 0.f) {
    u1 = scale(g1o, 1.f / sqrtf(n1));
  } else {  // rank <= 1: any unit vector orthogonal to u0 (deterministic choice)
    Vec3 a = fabsf(u0.x) < 0.9f ? v3(1, 0, 0) : v3(0, 1, 0);
    if (s0 <= 0.f) a = v3(0, 1, 0);
    Vec3 p = sub(a, scale(u0, dot(a, u0)));
    u1 = scale(p, 1.f / sqrtf(dot(p, p)));
  }
  const Vec3 u2 = cross(u0, u1);
  const Vec3 w2r = cross(w0, w1);
  // R[i][j] = sum_k v_k[i] * u_k[j]
  R[0] = w0.x * u0.x + w1.x * u1.x + w2r.x * u2.x;
  R[1] = w0.x * u0.y + w1.x * u1.y + w2r.x * u2.y;
  R[2] = w0.x * u0.z + w1.x * u1.z + w2r.x * u2.z;
  R[3] = w0.y * u0.x + w1.y * u1.x + w2r.y * u2.x;
  R[4] = w0.y * u0.y + w1.y * u1.y + w2r.y * u2.y;
  R[5] = w0.y * u0.z + w1.y * u1.z + w2r.y * u2.z;
  R[6] = w0.z * u0.x + w1.z * u1.x + w2r.z * u2.x;
  R[7] = w0.z * u0.y + w1.z * u1.y + w2r.z * u2.y;
  R[8] = w0.z * u0.z + w1.z * u1.z + w2r.z * u2.z;
}

}  // namespace unopose
