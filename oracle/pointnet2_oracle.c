/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not shipped, not measured as product.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.
 *
 * Plain-C host restatement of the nine PointNet2 `_ext` operators of the
 * reference (core/unopose/model/pointnet2/_ext_src/src/{sampling,ball_query,group_points,interpolate}_gpu.cu).  The reference
 * kernels are CUDA-only (every CPU branch is TORCH_CHECK(false, "CPU not
 * supported"), e.g. sampling.cpp:38-40) and the reference ships no golden
 * vectors for them (its only test is a loose gradcheck, pointnet2_test.py:20-33),
 * so THIS FILE IS "PARITY UNPINNED" AGAINST A CUDA RUN.  What pins it instead:
 *   - furthest_point_sampling emulates the thread block literally (per-thread
 *     strided scan + shared-memory tree with the __update tie rule) and is
 *     cross-checked in tests/ against the closed-form rule the emulation revealed:
 *     "argmax d, ties -> min (bitreverse_{log2 bs}(k mod bs), k)" -- the tree keeps the
 *     lower SLOT at every level, which is bit-reversed order, not the plain
 *     "lowest k mod bs" SURVEY.md App-B.1 assumed (tests/helpers.py::fps_closed_form);
 *   - the Python callers of these ops (QueryAndLRFGroup, sample_pts_feats, ...)
 *     are run from /root/reference with `_ext` bound to this library when the
 *     golden fixtures under tests/golden/ are generated.
 *
 * Build: gcc -O2 -ffp-contract=off (NO fused multiply-add: the HIP kernels are
 * built the same way so fp32 distance compares are bit-identical).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* cuda_utils.h:18-24  opt_n_threads: largest power of two <= work_size, in [1,512] */
int oracle_opt_n_threads(int work_size) {
  int pow_2 = (int)(log((double)work_size) / log(2.0));
  int t = 1 << pow_2;
  if (t > 512) t = 512;
  if (t < 1) t = 1;
  return t;
}

/* sampling_gpu.cu:74-178 (kernel), sampling.cpp:70-91 (host: idxs zeros, temp 1e10).
 * Literal block emulation: `bs` threads, thread tid scans k = tid, tid+bs, ...
 * keeping the first strict maximum, then the shared-memory tree
 * (tid, tid+s) for s = bs/2 .. 1 with __update (sampling_gpu.cu:64-70):
 *   dists[i1] = max(v1, v2); dists_i[i1] = v2 > v1 ? i2 : i1.            */
void oracle_furthest_point_sampling(int b, int n, int m, const float *dataset,
                                    float *temp, int *idxs) {
  if (m <= 0) return;
  const int bs = oracle_opt_n_threads(n);
  float *dists = (float *)malloc(sizeof(float) * bs);
  int *dists_i = (int *)malloc(sizeof(int) * bs);
  for (int bi = 0; bi < b; ++bi) {
    const float *pts = dataset + (size_t)bi * n * 3;
    float *tmp = temp + (size_t)bi * n;
    int *out = idxs + (size_t)bi * m;
    int old = 0;
    out[0] = old;
    for (int j = 1; j < m; ++j) {
      const float x1 = pts[old * 3 + 0], y1 = pts[old * 3 + 1], z1 = pts[old * 3 + 2];
      for (int tid = 0; tid < bs; ++tid) {
        int besti = 0;
        float best = -1.0f;
        for (int k = tid; k < n; k += bs) {
          const float x2 = pts[k * 3 + 0], y2 = pts[k * 3 + 1], z2 = pts[k * 3 + 2];
          const float d = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) + (z2 - z1) * (z2 - z1);
          const float d2 = fminf(d, tmp[k]);
          tmp[k] = d2;
          besti = d2 > best ? k : besti;
          best = d2 > best ? d2 : best;
        }
        dists[tid] = best;
        dists_i[tid] = besti;
      }
      for (int s = bs / 2; s >= 1; s >>= 1) {
        for (int tid = 0; tid < s; ++tid) {
          const float v1 = dists[tid], v2 = dists[tid + s];
          const int i1 = dists_i[tid], i2 = dists_i[tid + s];
          dists[tid] = fmaxf(v1, v2);
          dists_i[tid] = v2 > v1 ? i2 : i1;
        }
      }
      old = dists_i[0];
      out[j] = old;
    }
  }
  free(dists);
  free(dists_i);
}

/* sampling_gpu.cu:13-25  out[b,c,j] = points[b,c,idx[b,j]] */
void oracle_gather_points(int b, int c, int n, int m, const float *points,
                          const int *idx, float *out) {
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < m; ++j) {
        int a = idx[(size_t)i * m + j];
        out[((size_t)i * c + l) * m + j] = points[((size_t)i * c + l) * n + a];
      }
}

/* sampling_gpu.cu:39-52  scatter-add (atomicAdd order is unspecified on the GPU;
 * here: ascending j) */
void oracle_gather_points_grad(int b, int c, int n, int m, const float *grad_out,
                               const int *idx, float *grad_points) {
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < m; ++j) {
        int a = idx[(size_t)i * m + j];
        grad_points[((size_t)i * c + l) * n + a] += grad_out[((size_t)i * c + l) * m + j];
      }
}

/* ball_query_gpu.cu:14-49; idx zero-initialised by the host (ball_query.cpp:24-26).
 * First `nsample` k (ascending) with d2 < radius^2 (strict); on the first hit the
 * whole row is filled with k.                                                  */
void oracle_ball_query(int b, int n, int m, float radius, int nsample,
                       const float *new_xyz, const float *xyz, int *idx) {
  const float radius2 = radius * radius;
  for (int bi = 0; bi < b; ++bi) {
    const float *P = xyz + (size_t)bi * n * 3;
    const float *Q = new_xyz + (size_t)bi * m * 3;
    int *I = idx + (size_t)bi * m * nsample;
    for (int j = 0; j < m; ++j) {
      const float new_x = Q[j * 3 + 0], new_y = Q[j * 3 + 1], new_z = Q[j * 3 + 2];
      for (int k = 0, cnt = 0; k < n && cnt < nsample; ++k) {
        const float x = P[k * 3 + 0], y = P[k * 3 + 1], z = P[k * 3 + 2];
        const float d2 = (new_x - x) * (new_x - x) + (new_y - y) * (new_y - y) +
                         (new_z - z) * (new_z - z);
        if (d2 < radius2) {
          if (cnt == 0)
            for (int l = 0; l < nsample; ++l) I[(size_t)j * nsample + l] = k;
          I[(size_t)j * nsample + cnt] = k;
          ++cnt;
        }
      }
    }
  }
}

/* group_points_gpu.cu:13-33  out[b,c,j,k] = points[b,c,idx[b,j,k]] */
void oracle_group_points(int b, int c, int n, int npoints, int nsample,
                         const float *points, const int *idx, float *out) {
  for (int bi = 0; bi < b; ++bi)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < npoints; ++j)
        for (int k = 0; k < nsample; ++k) {
          int ii = idx[((size_t)bi * npoints + j) * nsample + k];
          out[(((size_t)bi * c + l) * npoints + j) * nsample + k] =
              points[((size_t)bi * c + l) * n + ii];
        }
}

/* group_points_gpu.cu:48-69 */
void oracle_group_points_grad(int b, int c, int n, int npoints, int nsample,
                              const float *grad_out, const int *idx, float *grad_points) {
  for (int bi = 0; bi < b; ++bi)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < npoints; ++j)
        for (int k = 0; k < nsample; ++k) {
          int ii = idx[((size_t)bi * npoints + j) * nsample + k];
          grad_points[((size_t)bi * c + l) * n + ii] +=
              grad_out[(((size_t)bi * c + l) * npoints + j) * nsample + k];
        }
}

/* interpolate_gpu.cu:14-64  three nearest neighbours; `double` running bests,
 * float d, strict <, outputs rounded back to float                            */
void oracle_three_nn(int b, int n, int m, const float *unknown, const float *known,
                     float *dist2, int *idx) {
  for (int bi = 0; bi < b; ++bi) {
    const float *U = unknown + (size_t)bi * n * 3;
    const float *K = known + (size_t)bi * m * 3;
    for (int j = 0; j < n; ++j) {
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
      float *D = dist2 + ((size_t)bi * n + j) * 3;
      int *I = idx + ((size_t)bi * n + j) * 3;
      D[0] = (float)best1; D[1] = (float)best2; D[2] = (float)best3;
      I[0] = besti1; I[1] = besti2; I[2] = besti3;
    }
  }
}

/* interpolate_gpu.cu:77-106 */
void oracle_three_interpolate(int b, int c, int m, int n, const float *points,
                              const int *idx, const float *weight, float *out) {
  for (int bi = 0; bi < b; ++bi)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < n; ++j) {
        const float *W = weight + ((size_t)bi * n + j) * 3;
        const int *I = idx + ((size_t)bi * n + j) * 3;
        const float *P = points + ((size_t)bi * c + l) * m;
        out[((size_t)bi * c + l) * n + j] = P[I[0]] * W[0] + P[I[1]] * W[1] + P[I[2]] * W[2];
      }
}

/* interpolate_gpu.cu:121-148 */
void oracle_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                                   const int *idx, const float *weight, float *grad_points) {
  for (int bi = 0; bi < b; ++bi)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < n; ++j) {
        const float *W = weight + ((size_t)bi * n + j) * 3;
        const int *I = idx + ((size_t)bi * n + j) * 3;
        float *G = grad_points + ((size_t)bi * c + l) * m;
        const float g = grad_out[((size_t)bi * c + l) * n + j];
        G[I[0]] += g * W[0];
        G[I[1]] += g * W[1];
        G[I[2]] += g * W[2];
      }
}
