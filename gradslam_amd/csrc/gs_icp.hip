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

// ---------------------------------------------------------------- API-level projective helpers
// project_points (geometry/projutils.py:92-238): pts = proj_mat [x y z w]^T (tiny matmul: plain mul / add, ascending
// k), u = x' / z', v = y' / z' with z' replaced by 1 where it is 0.  Point i uses matrix i / pts_per_mat.
__global__ void __launch_bounds__(256) gs_project_points_kernel(const float* __restrict__ cam, int cdim, int64_t n,
                                                                const float* __restrict__ proj, int64_t pts_per_mat,
                                                                float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* M = proj + 16 * (i / pts_per_mat);
  float p[4] = {cam[cdim * i], cam[cdim * i + 1], cam[cdim * i + 2], cdim == 4 ? cam[4 * i + 3] : 1.0f};
  float r[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float acc = M[4 * a] * p[0];
#pragma unroll
    for (int k = 1; k < 4; ++k) acc = acc + M[4 * a + k] * p[k];
    r[a] = acc;
  }
  const float z = r[2] != 0.0f ? r[2] : 1.0f;
  out[2 * i] = r[0] / z;
  out[2 * i + 1] = r[1] / z;
}
extern "C" int gs_project_points_f32(const float* cam_coords, int cdim, int64_t n, const float* proj16,
                                     int64_t pts_per_mat, float* out_uv, void* stream) {
  GS_REQUIRE(n >= 0 && (cdim == 3 || cdim == 4) && pts_per_mat > 0, "bad arguments");
  if (n == 0) return GS_OK;
  GS_REQUIRE(cam_coords && proj16 && out_uv, "NULL pointer");
  hipLaunchKernelGGL(gs_project_points_kernel, dim3((unsigned)gs_ceil_div(n, 256)), dim3(256), 0, gs_stream(stream),
                     cam_coords, cdim, n, proj16, pts_per_mat, out_uv);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

// unproject_points (geometry/projutils.py:241-402): (K^-1 [u v w]^T) * depth, w = 1 for (*, 2) inputs.
__global__ void __launch_bounds__(256) gs_unproject_points_kernel(const float* __restrict__ pix, int pdim, int64_t n,
                                                                  const float* __restrict__ kinv, int64_t pts_per_mat,
                                                                  const float* __restrict__ depth,
                                                                  float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* M = kinv + 9 * (i / pts_per_mat);
  const float p[3] = {pix[pdim * i], pix[pdim * i + 1], pdim == 3 ? pix[3 * i + 2] : 1.0f};
  const float d = depth[i];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float acc = M[3 * a] * p[0];
    acc = acc + M[3 * a + 1] * p[1];
    acc = acc + M[3 * a + 2] * p[2];
    out[3 * i + a] = acc * d;
  }
}
extern "C" int gs_unproject_points_f32(const float* pixel_coords, int pdim, int64_t n, const float* kinv9,
                                       int64_t pts_per_mat, const float* depths, float* out_xyz, void* stream) {
  GS_REQUIRE(n >= 0 && (pdim == 2 || pdim == 3) && pts_per_mat > 0, "bad arguments");
  if (n == 0) return GS_OK;
  GS_REQUIRE(pixel_coords && kinv9 && depths && out_xyz, "NULL pointer");
  hipLaunchKernelGGL(gs_unproject_points_kernel, dim3((unsigned)gs_ceil_div(n, 256)), dim3(256), 0, gs_stream(stream),
                     pixel_coords, pdim, n, kinv9, pts_per_mat, depths, out_xyz);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

// so3_hat / se3_hat / so3_exp (geometry/se3utils.py:11-74).  op 0: omega (3) -> 3x3 hat; op 1: xi (6) -> 4x4 hat;
// op 2: omega (3) -> 3x3 rotation (the rotation block of gs_se3_exp_dev: double-precision Rodrigues, rounded once).
__global__ void gs_lie_small_kernel(int op, const float* __restrict__ in, float* __restrict__ out) {
  if (threadIdx.x != 0) return;
  if (op == 2) {
    const float xi[6] = {0.0f, 0.0f, 0.0f, in[0], in[1], in[2]};
    float T[16];
    gs_se3_exp_dev(xi, T);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) out[3 * i + j] = T[4 * i + j];
    return;
  }
  const int n = op == 0 ? 3 : 4;
  const float* w = op == 0 ? in : in + 3;
  for (int i = 0; i < n * n; ++i) out[i] = 0.0f;
  out[0 * n + 1] = -w[2]; out[1 * n + 0] = w[2];
  out[0 * n + 2] = w[1];  out[2 * n + 0] = -w[1];
  out[1 * n + 2] = -w[0]; out[2 * n + 1] = w[0];
  if (op == 1) { out[3] = in[0]; out[7] = in[1]; out[11] = in[2]; }
}
extern "C" int gs_lie_small_f32(int op, const float* in, float* out, void* stream) {
  GS_REQUIRE(op >= 0 && op <= 2 && in && out, "bad arguments");
  hipLaunchKernelGGL(gs_lie_small_kernel, dim3(1), dim3(64), 0, gs_stream(stream), op, in, out);
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
