// gs_icp.hip — API-level pieces of K4: gauss_newton_solve rows, solve_linear_system, se3_exp,
// transform_pointcloud (odometry/icputils.py:22-232, geometry/se3utils.py:77-115,
// geometry/geometryutils.py:737-794).  The device-resident LM loop is in gs_icp_loop.hip.
#include "gs_icp_math.h"
#include "gs_knn.h"

__global__ void __launch_bounds__(256) gs_gn_rows_kernel(
    const float* __restrict__ src, int64_t n_src, const float* __restrict__ tgt,
    const float* __restrict__ tn, int64_t n_tgt, const unsigned long long* __restrict__ best, float dist_thresh,
    float* __restrict__ A, float* __restrict__ b, int64_t* __restrict__ idx, uint8_t* __restrict__ keep) {
  const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= n_src) return;
  const unsigned long long bb = best[s];
  int64_t j = (int64_t)(bb & 0xffffffffull);
  if (j >= n_tgt) j = 0;  // only when every distance was NaN
  const float d2 = __uint_as_float((uint32_t)(bb >> 32));
  float a[6], r;
  gn_row(src[3 * s], src[3 * s + 1], src[3 * s + 2], tgt, tn, j, a, r);
#pragma unroll
  for (int k = 0; k < 6; ++k) A[6 * s + k] = a[k];
  b[s] = r;
  idx[s] = j;
  if (keep) keep[s] = (dist_thresh < 0.0f) ? 1 : (d2 < dist_thresh ? 1 : 0);
}

extern "C" int gs_gauss_newton_rows_f32(const float* src, int64_t n_src, const float* tgt,
                                        const float* tgt_normals, int64_t n_tgt, float dist_thresh,
                                        float* A, float* b, int64_t* idx, uint8_t* keep,
                                        uint64_t* best_scratch, void* stream) {
  GS_REQUIRE(n_src > 0 && n_tgt > 0, "empty point set");
  GS_REQUIRE(src && tgt && tgt_normals && A && b && idx && best_scratch, "NULL pointer");
  hipStream_t st = gs_stream(stream);
  GS_HIP(hipMemsetAsync(best_scratch, 0xff, 8 * (size_t)n_src, st));
  gs_knn_brute_launch(src, nullptr, nullptr, n_src, tgt, n_tgt, reinterpret_cast<unsigned long long*>(best_scratch), st);
  hipLaunchKernelGGL(gs_gn_rows_kernel, dim3((unsigned)gs_ceil_div(n_src, 256)), dim3(256), 0, st, src,
                     n_src, tgt, tgt_normals, n_tgt, reinterpret_cast<const unsigned long long*>(best_scratch),
                     dist_thresh, A, b, idx, keep);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

__global__ void gs_se3_exp_kernel(const float* __restrict__ xi6, float* __restrict__ T16) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float xi[6], T[16];
    for (int i = 0; i < 6; ++i) xi[i] = xi6[i];
    gs_se3_exp_dev(xi, T);
    for (int i = 0; i < 16; ++i) T16[i] = T[i];
  }
}
extern "C" int gs_se3_exp_f32(const float* xi6, float* T16, void* stream) {
  GS_REQUIRE(xi6 && T16, "NULL pointer");
  hipLaunchKernelGGL(gs_se3_exp_kernel, dim3(1), dim3(64), 0, gs_stream(stream), xi6, T16);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

// relative_transformation (geometry/geometryutils.py:413-478, orthogonal_rotations=False):
// inv(T01) by Gauss-Jordan with partial pivoting in double, rounded once, then kornia's
// compose_transformations ([R1 R2, R1 t2 + t1], plain float32, ascending k).  One lane per pair.
__global__ void __launch_bounds__(64) gs_relative_pose_kernel(const float* __restrict__ T01, const float* __restrict__ T02,
                                                              int64_t n, float* __restrict__ out) {
  const int64_t m = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (m >= n) return;
  double a[4][8];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      a[i][j] = (double)T01[16 * m + 4 * i + j];
      a[i][4 + j] = (i == j) ? 1.0 : 0.0;
    }
  for (int c = 0; c < 4; ++c) {
    int p = c;
    for (int r = c + 1; r < 4; ++r)
      if (fabs(a[r][c]) > fabs(a[p][c])) p = r;
    if (p != c)
      for (int j = 0; j < 8; ++j) { const double t = a[c][j]; a[c][j] = a[p][j]; a[p][j] = t; }
    const double inv = 1.0 / a[c][c];
    for (int j = 0; j < 8; ++j) a[c][j] *= inv;
    for (int r = 0; r < 4; ++r) {
      if (r == c) continue;
      const double f = a[r][c];
      for (int j = 0; j < 8; ++j) a[r][j] -= f * a[c][j];
    }
  }
  float A[16], B[16], C[16];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      A[4 * i + j] = (float)a[i][4 + j];
      B[4 * i + j] = T02[16 * m + 4 * i + j];
    }
  gs_compose_rigid(A, B, C);
  for (int i = 0; i < 16; ++i) out[16 * m + i] = C[i];
}
extern "C" int gs_relative_pose_f32(const float* T01, const float* T02, int64_t n, float* out, void* stream) {
  GS_REQUIRE(n >= 0, "bad size");
  if (n == 0) return GS_OK;
  GS_REQUIRE(T01 && T02 && out, "NULL pointer");
  hipLaunchKernelGGL(gs_relative_pose_kernel, dim3((unsigned)gs_ceil_div(n, 64)), dim3(64), 0, gs_stream(stream), T01,
                     T02, n, out);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

__global__ void __launch_bounds__(256) gs_transform_points_kernel(const float* __restrict__ pts, int64_t n,
                                                                  const float* __restrict__ T16,
                                                                  float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float T[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) T[k] = T16[k];
  float o0, o1, o2;
  gs_rigid_fma(T, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], o0, o1, o2);
  out[3 * i] = o0; out[3 * i + 1] = o1; out[3 * i + 2] = o2;
}
extern "C" int gs_transform_points_f32(const float* pts, int64_t n, const float* T16, float* out,
                                       void* stream) {
  GS_REQUIRE(n >= 0, "negative n");
  if (n == 0) return GS_OK;
  GS_REQUIRE(pts && T16 && out, "NULL pointer");
  hipLaunchKernelGGL(gs_transform_points_kernel, dim3((unsigned)gs_ceil_div(n, 256)), dim3(256), 0,
                     gs_stream(stream), pts, n, T16, out);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

// ---------------------------------------------------------------- generic normal eq ----
// solve_linear_system for A (n_rows, ncols <= 8): one block, float64 accumulation.
__global__ void __launch_bounds__(256) gs_solve_normal_eq_kernel(const float* __restrict__ A,
                                                                 const float* __restrict__ b,
                                                                 const uint8_t* __restrict__ keep,
                                                                 int64_t n_rows, int ncols, float damp,
                                                                 float* __restrict__ x) {
  __shared__ double red[4][44];
  double acc[44];
#pragma unroll
  for (int i = 0; i < 44; ++i) acc[i] = 0.0;
  for (int64_t r = threadIdx.x; r < n_rows; r += 256) {
    if (keep && !keep[r]) continue;
    float a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = (k < ncols) ? A[ncols * r + k] : 0.0f;
    const float br = b[r];
    int q = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = i; j < 8; ++j) acc[q++] += (double)a[i] * (double)a[j];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[36 + i] += (double)a[i] * (double)br;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 44; ++i) {
    const double s = gs_wave_sum_f64(acc[i]);
    if (lane == 0) red[wave][i] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double S[44];
    for (int i = 0; i < 44; ++i) S[i] = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
    float AtA[64], Atb[8], xs[8];
    int q = 0;
    for (int i = 0; i < 8; ++i)
      for (int j = i; j < 8; ++j) {
        if (i < ncols && j < ncols) AtA[ncols * i + j] = AtA[ncols * j + i] = (float)S[q];
        ++q;
      }
    for (int i = 0; i < ncols; ++i) Atb[i] = (float)S[36 + i];
    gs_solve_normal(AtA, Atb, damp, ncols, xs);
    for (int i = 0; i < ncols; ++i) x[i] = xs[i];
  }
}

extern "C" int gs_solve_normal_eq_f32(const float* A, const float* b, const uint8_t* keep,
                                      int64_t n_rows, int ncols, float damp, float* x, void* stream) {
  GS_REQUIRE(n_rows > 0 && ncols >= 1 && ncols <= 8, "need n_rows > 0 and 1 <= ncols <= 8");
  GS_REQUIRE(A && b && x, "NULL pointer");
  hipLaunchKernelGGL(gs_solve_normal_eq_kernel, dim3(1), dim3(256), 0, gs_stream(stream), A, b, keep, n_rows,
                     ncols, damp, x);
  GS_LAUNCH_CHECK();
  return GS_OK;
}
