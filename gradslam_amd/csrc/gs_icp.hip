// gs_icp.hip — K4: Gauss-Newton system, 6x6 solve, SE(3) exponential, LM / gradLM update, with
// the whole 20-iteration loop (2 exact 1-NN searches per iteration, gs_knn.hip) enqueued without
// a host sync.
//
// K4 accumulates J^T J / J^T r / r^T r in float64 from float32 products (order-independent to
// ~1e-16, so HIP and oracle agree after the single rounding to float32) with fixed-order
// wave -> block -> grid reduction, then one lane solves and updates on the device.
#include <stdlib.h>
#include <string.h>

#include "gs_knn.h"

// Device-resident state of one ICP solve (floats unless noted), lives in icp_scratch.
struct GsIcpState {
  float T_total[16];
  float Tr[16];      // residual transform of the current iteration: se3_exp(xi)
  float T_step[16];  // transform actually applied to the source cloud at the end of the iteration
  float xi[8];
  float damp;
  float err;
  float pad[6];
  float trace[64 * 12];  // up to 64 iterations
};

// ---------------------------------------------------------------- K4: rows -------------
// odometry/icputils.py:210-230 for one source point and its associated target.
GS_DEV void gn_row(float sx, float sy, float sz, const float* __restrict__ tgt,
                   const float* __restrict__ tn, int64_t j, float* a, float& b) {
  const float dx = tgt[3 * j], dy = tgt[3 * j + 1], dz = tgt[3 * j + 2];
  const float nx = tn[3 * j], ny = tn[3 * j + 1], nz = tn[3 * j + 2];
  a[0] = nx; a[1] = ny; a[2] = nz;
  a[3] = nz * sy - ny * sz;
  a[4] = nx * sz - nz * sx;
  a[5] = ny * sx - nx * sy;
  const float t = nx * (dx - sx) + ny * (dy - sy);
  b = t + nz * (dz - sz);
}

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

// ---------------------------------------------------------------- small dense algebra ---
// Solve (AtA + damp I) x = Atb (odometry/icputils.py:85-90; the reference inverts in float32 with
// LAPACK and multiplies).  The system is symmetric positive definite, so it is solved directly by
// un-pivoted Gauss-Jordan elimination in double on the augmented matrix and rounded once; N is a
// template parameter so that the whole elimination lives in registers (no scratch memory).
// Operation order is identical to oracle/gs_oracle.c:solve_spd_f64.
template <int N>
GS_DEV void gs_solve_spd(const float* AtA, const float* Atb, float damp, float* x) {
  double a[N][N + 1];
#pragma unroll
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const float e = (i == j) ? 1.0f : 0.0f;
      const float m = AtA[N * i + j] + e * damp;  // At_A + damp_matrix * damp, in float32
      a[i][j] = (double)m;
    }
    a[i][N] = (double)Atb[i];
  }
#pragma unroll
  for (int c = 0; c < N; ++c) {
    const double inv = 1.0 / a[c][c];
#pragma unroll
    for (int j = c; j <= N; ++j) a[c][j] *= inv;
#pragma unroll
    for (int r = 0; r < N; ++r) {
      if (r == c) continue;
      const double f = a[r][c];
#pragma unroll
      for (int j = c; j <= N; ++j) a[r][j] -= f * a[c][j];
    }
  }
#pragma unroll
  for (int i = 0; i < N; ++i) x[i] = (float)a[i][N];
}

GS_DEV void gs_solve_normal(const float* AtA, const float* Atb, float damp, int n, float* x) {
  switch (n) {
    case 1: gs_solve_spd<1>(AtA, Atb, damp, x); break;
    case 2: gs_solve_spd<2>(AtA, Atb, damp, x); break;
    case 3: gs_solve_spd<3>(AtA, Atb, damp, x); break;
    case 4: gs_solve_spd<4>(AtA, Atb, damp, x); break;
    case 5: gs_solve_spd<5>(AtA, Atb, damp, x); break;
    case 6: gs_solve_spd<6>(AtA, Atb, damp, x); break;
    case 7: gs_solve_spd<7>(AtA, Atb, damp, x); break;
    default: gs_solve_spd<8>(AtA, Atb, damp, x); break;
  }
}

// geometry/se3utils.py:77-115 in double, rounded once (same order as the oracle).
GS_DEV void gs_se3_exp_dev(const float* xi6, float* T16) {
  const double v[3] = {xi6[0], xi6[1], xi6[2]}, w[3] = {xi6[3], xi6[4], xi6[5]};
  const double wh[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
  double R[9], V[9];
  const double theta = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  if ((float)theta < 1e-6f) {
    for (int i = 0; i < 9; ++i) {
      R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + wh[i];
      V[i] = R[i];
    }
  } else {
    double wh2[9];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        double s = 0;
        for (int k = 0; k < 3; ++k) s += wh[3 * i + k] * wh[3 * k + j];
        wh2[3 * i + j] = s;
      }
    const double s = sin(theta), c = cos(theta);
    const double Ac = s / theta, Bc = (1 - c) / (theta * theta), Cc = (theta - s) / (theta * theta * theta);
    for (int i = 0; i < 9; ++i) {
      const double I = (i % 4 == 0) ? 1.0 : 0.0;
      R[i] = I + Ac * wh[i] + Bc * wh2[i];
      V[i] = I + Bc * wh[i] + Cc * wh2[i];
    }
  }
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) T16[4 * i + j] = (float)R[3 * i + j];
    T16[4 * i + 3] = (float)(V[3 * i] * v[0] + V[3 * i + 1] * v[1] + V[3 * i + 2] * v[2]);
  }
  T16[12] = 0; T16[13] = 0; T16[14] = 0; T16[15] = 1;
}

// torch.mm of two 4x4 (odometry/icputils.py:362,543): tiny matmul, plain, ascending k.
GS_DEV void gs_mm4(const float* A, const float* B, float* C) {
  float t[16];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      float acc = A[4 * i] * B[j];
      for (int k = 1; k < 4; ++k) acc = acc + A[4 * i + k] * B[4 * k + j];
      t[4 * i + j] = acc;
    }
  for (int i = 0; i < 16; ++i) C[i] = t[i];
}
// kornia compose_transformations (slam/icpslam.py:245-247).
GS_DEV void gs_compose_rigid(const float* A, const float* B, float* C) {
  float t[16];
  for (int i = 0; i < 16; ++i) t[i] = 0.0f;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) {
      float acc = A[4 * i] * B[j];
      for (int k = 1; k < 3; ++k) acc = acc + A[4 * i + k] * B[4 * k + j];
      t[4 * i + j] = acc;
    }
    float acc = A[4 * i] * B[3];
    for (int k = 1; k < 3; ++k) acc = acc + A[4 * i + k] * B[4 * k + 3];
    t[4 * i + 3] = acc + A[4 * i + 3];
  }
  t[15] = 1.0f;
  for (int i = 0; i < 16; ++i) C[i] = t[i];
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

// ---------------------------------------------------------------- ICP loop kernels -----
constexpr int LIN_BLOCK = 256;
constexpr int LIN_NV = 28;  // 21 upper-triangular JtJ + 6 Jtr + 1 rtr

// Reads the KNN result of every source point (and re-arms best[] for the next search),
// builds its row and reduces the normal equations.  FULL = false: residual only (look-ahead).
template <bool FULL>
__global__ void __launch_bounds__(LIN_BLOCK) gs_icp_linearize_kernel(
    const float* __restrict__ src, const float* __restrict__ Tapply, int64_t n_src,
    const float* __restrict__ tgt, const float* __restrict__ tn, int64_t n_tgt,
    unsigned long long* __restrict__ best, float dist_thresh, double* __restrict__ partials,
    int64_t* __restrict__ out_idx) {
  __shared__ double red[LIN_BLOCK / GS_WAVE][LIN_NV];
  const int64_t s = (int64_t)blockIdx.x * LIN_BLOCK + threadIdx.x;
  double v[LIN_NV];
#pragma unroll
  for (int i = 0; i < LIN_NV; ++i) v[i] = 0.0;
  if (s < n_src) {
    const unsigned long long bb = best[s];
    best[s] = ~0ull;
    int64_t j = (int64_t)(bb & 0xffffffffull);
    if (j >= n_tgt) j = 0;  // only when every distance was NaN
    const float d2 = __uint_as_float((uint32_t)(bb >> 32));
    const bool keep = (dist_thresh < 0.0f) || (d2 < dist_thresh);
    float p0 = src[3 * s], p1 = src[3 * s + 1], p2 = src[3 * s + 2];
    if (Tapply) {
      float T[12];
#pragma unroll
      for (int i = 0; i < 12; ++i) T[i] = Tapply[i];
      float q0, q1, q2;
      gs_rigid_fma(T, p0, p1, p2, q0, q1, q2);
      p0 = q0; p1 = q1; p2 = q2;
    }
    float a[6], r;
    gn_row(p0, p1, p2, tgt, tn, j, a, r);
    if (FULL && out_idx) out_idx[s] = j;
    if (keep) {
      if (FULL) {
        int q = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
          for (int k = i; k < 6; ++k) v[q++] = (double)a[i] * (double)a[k];
#pragma unroll
        for (int i = 0; i < 6; ++i) v[21 + i] = (double)a[i] * (double)r;
      }
      v[27] = (double)r * (double)r;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = FULL ? 0 : 27; i < LIN_NV; ++i) {
    const double sum = gs_wave_sum_f64(v[i]);
    if (lane == 0) red[wave][i] = sum;
  }
  __syncthreads();
  if (threadIdx.x < LIN_NV && (FULL || threadIdx.x == 27)) {
    const int i = threadIdx.x;
    partials[(int64_t)blockIdx.x * LIN_NV + i] = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
  }
}

// ---------------------------------------------------------------- fused search + linearise
// One kernel per half-iteration when the grid engine is active: a 512-thread block serves 32
// source points (one per 16-lane group), finishes the rare unresolved queries itself by a
// whole-block brute-force scan, then wave 0 builds the 32 rows and reduces them to ONE partial
// row.  No neighbour table round-trips through HBM, no separate fallback / linearise launches.
constexpr int FS_BLOCK = 512;            // 8 waves; 2-3 blocks per CU keep every block of a 640x480 solve resident
constexpr int FS_QPB = FS_BLOCK / GQ_G;  // 32 queries per block, their rows are built by wave 0

template <bool FULL>
__global__ void __launch_bounds__(FS_BLOCK) gs_icp_search_linearize_kernel(
    const float* __restrict__ src_in, const float* __restrict__ Tapply, float* __restrict__ src_out, int64_t n_src,
    const float* __restrict__ tgt, const float* __restrict__ tn, int64_t n_tgt, const GsGrid* __restrict__ gp,
    const int* __restrict__ cell_start, const float4* __restrict__ sorted, float dist_thresh,
    double* __restrict__ partials, int64_t* __restrict__ out_idx) {
  __shared__ unsigned long long keys_s[FS_QPB];
  __shared__ float qs[FS_QPB][3];
  __shared__ int unres_q[FS_QPB];
  __shared__ int unres_n;
  __shared__ unsigned long long red[FS_BLOCK / GS_WAVE];
  const int lane = threadIdx.x & (GQ_G - 1), slot = threadIdx.x / GQ_G;
  const int64_t s = (int64_t)blockIdx.x * FS_QPB + slot;
  if (threadIdx.x == 0) unres_n = 0;
  __syncthreads();
  if (s < n_src) {
    const GsGrid g = *gp;
    float qx = src_in[3 * s], qy = src_in[3 * s + 1], qz = src_in[3 * s + 2];
    if (Tapply) {
      float T[12];
#pragma unroll
      for (int i = 0; i < 12; ++i) T[i] = Tapply[i];
      float t0, t1, t2;
      gs_rigid_fma(T, qx, qy, qz, t0, t1, t2);
      qx = t0; qy = t1; qz = t2;
    }
    bool done;
    const unsigned long long key = grid_search16(g, cell_start, sorted, qx, qy, qz, lane, &done);
    if (lane == 0) {
      if (src_out) {
        src_out[3 * s] = qx;
        src_out[3 * s + 1] = qy;
        src_out[3 * s + 2] = qz;
      }
      qs[slot][0] = qx; qs[slot][1] = qy; qs[slot][2] = qz;
      keys_s[slot] = key;
      if (!done) unres_q[atomicAdd(&unres_n, 1)] = slot;
    }
  }
  __syncthreads();
  const int nun = unres_n;  // block-uniform
  for (int u = 0; u < nun; ++u) {
    const int us = unres_q[u];
    const unsigned long long key = block_brute_min<FS_BLOCK>(qs[us][0], qs[us][1], qs[us][2], tgt, n_tgt, red);
    if (threadIdx.x == 0) keys_s[us] = key;
  }
  __syncthreads();
  // rows: lane t < 32 of wave 0 builds the row of query t
  double v[LIN_NV];
#pragma unroll
  for (int i = 0; i < LIN_NV; ++i) v[i] = 0.0;
  const int64_t r = (int64_t)blockIdx.x * FS_QPB + threadIdx.x;
  if (threadIdx.x < FS_QPB && r < n_src) {
    const unsigned long long bb = keys_s[threadIdx.x];
    int64_t j = (int64_t)(bb & 0xffffffffull);
    if (j >= n_tgt) j = 0;  // only when every distance was NaN
    const float d2 = __uint_as_float((uint32_t)(bb >> 32));
    const bool keep = (dist_thresh < 0.0f) || (d2 < dist_thresh);
    float a[6], res;
    gn_row(qs[threadIdx.x][0], qs[threadIdx.x][1], qs[threadIdx.x][2], tgt, tn, j, a, res);
    if (FULL && out_idx) out_idx[r] = j;
    if (keep) {
      if (FULL) {
        int q = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
          for (int k = i; k < 6; ++k) v[q++] = (double)a[i] * (double)a[k];
#pragma unroll
        for (int i = 0; i < 6; ++i) v[21 + i] = (double)a[i] * (double)res;
      }
      v[27] = (double)res * (double)res;
    }
  }
  if (!FULL) {  // residual only: one value, a plain wave reduction is enough
    if (threadIdx.x < GS_WAVE) {
      const double sum = gs_wave_sum_f64(v[27]);
      if (threadIdx.x == 0) partials[(int64_t)blockIdx.x * LIN_NV + 27] = sum;
    }
    return;
  }
  // 32 rows x 28 values through LDS (a 28-fold wave shuffle reduction costs ~6 us of dependent
  // cross-lane traffic in one wave): 8 groups of 28 threads add 4 rows each, then 28 threads add
  // the 8 sub-sums, always in index order.
  __shared__ double rows_s[FS_QPB][LIN_NV + 1];
  __shared__ double sub_s[8][LIN_NV];
  if (threadIdx.x < FS_QPB) {
#pragma unroll
    for (int i = 0; i < LIN_NV; ++i) rows_s[threadIdx.x][i] = v[i];
  }
  __syncthreads();
  if (threadIdx.x < 8 * LIN_NV) {
    const int i = threadIdx.x % LIN_NV, part = threadIdx.x / LIN_NV;
    double t = rows_s[4 * part][i];
    t += rows_s[4 * part + 1][i];
    t += rows_s[4 * part + 2][i];
    t += rows_s[4 * part + 3][i];
    sub_s[part][i] = t;
  }
  __syncthreads();
  if (threadIdx.x < LIN_NV) {
    double t = sub_s[0][threadIdx.x];
#pragma unroll
    for (int k = 1; k < 8; ++k) t += sub_s[k][threadIdx.x];
    partials[(int64_t)blockIdx.x * LIN_NV + threadIdx.x] = t;
  }
}

// Fixed-order, block-parallel sum of the per-block partial rows (SUM_BLOCK threads): thread t adds
// rows t/32, t/32 + SUM_BLOCK/32, ... of value t%32, then value i is finished by adding the
// SUM_BLOCK/32 sub-sums in index order.  (A single lane walking all rows was latency-bound:
// one dependent global load per row.)  Values [first, LIN_NV) are produced in S[].
constexpr int SUM_BLOCK = 1024;
GS_DEV void icp_sum_partials(const double* __restrict__ partials, int nblk, int first, double* S,
                             double (*sub)[32]) {
  const int i = threadIdx.x & 31, j = threadIdx.x >> 5;
  double s = 0.0;
  if (i >= first && i < LIN_NV) {
    constexpr int STEP = SUM_BLOCK / 32;
    for (int b = j; b < nblk; b += 8 * STEP) {  // 8 independent loads in flight, added in row order
      double a[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = (b + u * STEP < nblk) ? partials[(int64_t)(b + u * STEP) * LIN_NV + i] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u) s += a[u];
    }
  }
  sub[j][i] = s;
  __syncthreads();
  if (threadIdx.x >= first && threadIdx.x < LIN_NV) {
    double t = 0.0;
    for (int k = 0; k < SUM_BLOCK / 32; ++k) t += sub[k][threadIdx.x];
    S[threadIdx.x] = t;
  }
  __syncthreads();
}

// After the first linearisation of an iteration: solve for xi, Tr = se3_exp(xi).
__global__ void __launch_bounds__(SUM_BLOCK) gs_icp_solve_kernel(const double* __restrict__ partials, int nblk,
                                                                 GsIcpState* __restrict__ st) {
  __shared__ double S[32];
  __shared__ double sub[SUM_BLOCK / 32][32];
  icp_sum_partials(partials, nblk, 0, S, sub);
  if (threadIdx.x != 0) return;
  float AtA[36], Atb[6], xi[6], Tr[16];
  int q = 0;
  for (int i = 0; i < 6; ++i)
    for (int k = i; k < 6; ++k) {
      AtA[6 * i + k] = AtA[6 * k + i] = (float)S[q];
      ++q;
    }
  for (int i = 0; i < 6; ++i) Atb[i] = (float)S[21 + i];
  st->err = (float)S[27];
  gs_solve_spd<6>(AtA, Atb, st->damp, xi);
  gs_se3_exp_dev(xi, Tr);
  for (int i = 0; i < 6; ++i) st->xi[i] = xi[i];
  for (int i = 0; i < 16; ++i) st->Tr[i] = Tr[i];
}

// After the look-ahead residual: LM accept/reject (mode 0, odometry/icputils.py:356-365) or
// the gradLM soft update (mode 1, :527-543).  On the last iteration writes the result.
__global__ void __launch_bounds__(SUM_BLOCK) gs_icp_update_kernel(const double* __restrict__ partials, int nblk,
                                                                  GsIcpState* __restrict__ st, gs_icp_params prm,
                                                                  int it, const float* __restrict__ compose16,
                                                                  float* __restrict__ out_T16) {
  __shared__ double S[32];
  __shared__ double sub[SUM_BLOCK / 32][32];
  icp_sum_partials(partials, nblk, 27, S, sub);
  if (threadIdx.x != 0) return;
  const float new_err = (float)S[27];
  const float err = st->err;
  float damp = st->damp;
  float Tstep[16], Ttot[16];
  for (int i = 0; i < 16; ++i) Ttot[i] = st->T_total[i];
  float sig = 1.0f;
  if (prm.mode == 0) {
    if (new_err < err) {
      for (int i = 0; i < 16; ++i) Tstep[i] = st->Tr[i];
      damp = damp / 2;
      gs_mm4(Tstep, Ttot, Ttot);
    } else {
      for (int i = 0; i < 16; ++i) Tstep[i] = (i % 5 == 0) ? 1.0f : 0.0f;
      damp = damp * 2;
    }
  } else {
    const float lmin = (float)(1.0 / (double)prm.lambda_max);
    const float lrange = (float)((double)prm.lambda_max - 1.0 / (double)prm.lambda_max);
    float errdiff = new_err - err;
    errdiff = errdiff < -70.0f ? -70.0f : (errdiff > 70.0f ? 70.0f : errdiff);
    const float e_b = (float)exp((double)((float)(-(double)prm.B) * errdiff));
    const float damp_new = lmin + lrange / (1.0f + e_b);
    damp = damp * damp_new;
    const float e_b2 = (float)exp((double)((float)(-(double)prm.B2) * errdiff));
    const float pw = (float)pow((double)(1.0f + e_b2), (double)(float)(1.0 / (double)prm.nu));
    sig = 1.0f / pw;
    float xs[6];
    for (int k = 0; k < 6; ++k) xs[k] = sig * st->xi[k];
    gs_se3_exp_dev(xs, Tstep);
    gs_mm4(Tstep, Ttot, Ttot);
  }
  st->damp = damp;
  for (int i = 0; i < 16; ++i) {
    st->T_step[i] = Tstep[i];
    st->T_total[i] = Ttot[i];
  }
  if (it < 64) {
    float* t = st->trace + 12 * it;
    t[0] = err; t[1] = new_err; t[2] = damp; t[3] = sig;
    for (int k = 0; k < 6; ++k) t[4 + k] = st->xi[k];
    t[10] = 0; t[11] = 0;
  }
  if (it == prm.numiters - 1) {
    float out[16];
    if (compose16) {
      float Cm[16];
      for (int i = 0; i < 16; ++i) Cm[i] = compose16[i];
      gs_compose_rigid(Ttot, Cm, out);
    } else {
      for (int i = 0; i < 16; ++i) out[i] = Ttot[i];
    }
    for (int i = 0; i < 16; ++i) out_T16[i] = out[i];
  }
}

__global__ void gs_icp_init_kernel(GsIcpState* __restrict__ st, const float* __restrict__ init16, float damp,
                                   int numiters, const float* __restrict__ compose16, float* __restrict__ out_T16) {
  if (threadIdx.x != 0) return;
  for (int i = 0; i < 16; ++i) {
    st->T_total[i] = init16[i];
    st->T_step[i] = init16[i];
    st->Tr[i] = (i % 5 == 0) ? 1.0f : 0.0f;
  }
  for (int i = 0; i < 8; ++i) st->xi[i] = 0.0f;
  st->damp = damp;
  st->err = 0.0f;
  if (numiters == 0) {  // degenerate: result is the (composed) initial transform
    float T[16], out[16];
    for (int i = 0; i < 16; ++i) T[i] = init16[i];
    if (compose16) {
      float Cm[16];
      for (int i = 0; i < 16; ++i) Cm[i] = compose16[i];
      gs_compose_rigid(T, Cm, out);
    } else {
      for (int i = 0; i < 16; ++i) out[i] = T[i];
    }
    for (int i = 0; i < 16; ++i) out_T16[i] = out[i];
  }
}

struct IcpScratch {
  unsigned long long* best;
  float* srcA;
  float* srcB;
  double* partials;
  GsIcpState* state;
  void* grid;
};
static IcpScratch icp_carve(void* scratch, int64_t n_src) {
  char* p = reinterpret_cast<char*>(scratch);
  IcpScratch s;
  s.state = reinterpret_cast<GsIcpState*>(p);
  p += gs_align(sizeof(GsIcpState));
  s.best = reinterpret_cast<unsigned long long*>(p);
  p += gs_align(8 * (size_t)n_src);
  s.srcA = reinterpret_cast<float*>(p);
  p += gs_align(12 * (size_t)n_src);
  s.srcB = reinterpret_cast<float*>(p);
  p += gs_align(12 * (size_t)n_src);
  s.partials = reinterpret_cast<double*>(p);
  p += gs_align(sizeof(double) * LIN_NV * (size_t)gs_ceil_div(n_src, FS_QPB));
  s.grid = p;
  return s;
}

// GRADSLAM_HIP_KNN=brute forces the brute-force engine (A/B runs; results are identical).
static bool icp_grid_enabled() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("GRADSLAM_HIP_KNN");
    v = (e && strcmp(e, "brute") == 0) ? 0 : 1;
  }
  return v == 1;
}

extern "C" int64_t gs_icp_scratch_bytes(int64_t n_src, int64_t n_tgt) {
  if (n_src < 1) n_src = 1;
  const int64_t nblk = gs_ceil_div(n_src, FS_QPB);  // the fused kernels emit one partial row per 32 points
  return (int64_t)(gs_align(sizeof(GsIcpState)) + gs_align(8 * (size_t)n_src) + 2 * gs_align(12 * (size_t)n_src) +
                   gs_align(sizeof(double) * LIN_NV * (size_t)nblk) + gs_knn_grid_scratch_bytes(n_src, n_tgt) + 4096);
}

extern "C" int gs_icp_f32(const float* src, int64_t n_src, const float* tgt, const float* tgt_normals,
                          int64_t n_tgt, const float* init16, const float* compose16,
                          const gs_icp_params* prm, float* out_T16, int64_t* out_idx, void* icp_scratch,
                          void* stream) {
  GS_REQUIRE(prm, "params_host must not be NULL");
  GS_REQUIRE(n_src > 0 && n_tgt > 0, "empty point set");
  GS_REQUIRE(n_tgt < 0xffffffffll, "too many targets");
  GS_REQUIRE(src && tgt && tgt_normals && init16 && out_T16 && icp_scratch, "NULL pointer");
  GS_REQUIRE(prm->numiters >= 0 && prm->numiters <= 64, "numiters must be in [0, 64]");
  GS_REQUIRE(prm->mode == 0 || prm->mode == 1, "mode must be 0 (ICP) or 1 (gradICP)");
  hipStream_t st = gs_stream(stream);
  IcpScratch sc = icp_carve(icp_scratch, n_src);
  const int nblk = (int)gs_ceil_div(n_src, LIN_BLOCK);
  hipLaunchKernelGGL(gs_icp_init_kernel, dim3(1), dim3(64), 0, st, sc.state, init16, prm->damp, prm->numiters,
                     compose16, out_T16);
  // the target set is fixed for all 2*numiters searches of this solve: bin it once
  const bool use_grid = icp_grid_enabled() && gs_knn_use_grid(n_src, n_tgt) && prm->numiters > 0;
  GridMem gm = grid_carve(sc.grid, n_src, n_tgt);
  if (use_grid) {
    int rc = gs_knn_grid_build(tgt, n_tgt, n_src, sc.grid, st);
    if (rc != GS_OK) return rc;
  } else {
    GS_HIP(hipMemsetAsync(sc.best, 0xff, 8 * (size_t)n_src, st));
  }
  const int nfs = (int)gs_ceil_div(n_src, FS_QPB);
  const int nrows = use_grid ? nfs : nblk;  // partial rows the solve / update kernels add up
  // one half-iteration: search (with the pending transform applied on load) + rows + partial sums
  auto half = [&](bool full, const float* s_in, const float* Tapply, float* s_out) {
    if (use_grid) {
      // compulsory bytes of one fused half-iteration: source in (+out on the first half), 27 cell bounds
      // (8 B) per query, matched target + normal gather, partial rows, one pass over the binned targets
      GsProf prof(GS_PROF_ICP_FUSED, (double)n_src * (full ? 271.0 : 259.0) + 16.0 * (double)n_tgt, st);
      if (full)
        hipLaunchKernelGGL((gs_icp_search_linearize_kernel<true>), dim3(nfs), dim3(FS_BLOCK), 0, st, s_in, Tapply,
                           s_out, n_src, tgt, tgt_normals, n_tgt, gm.g, gm.cell_start, gm.sorted, prm->dist_thresh,
                           sc.partials, out_idx);
      else
        hipLaunchKernelGGL((gs_icp_search_linearize_kernel<false>), dim3(nfs), dim3(FS_BLOCK), 0, st, s_in, Tapply,
                           s_out, n_src, tgt, tgt_normals, n_tgt, gm.g, gm.cell_start, gm.sorted, prm->dist_thresh,
                           sc.partials, nullptr);
      return;
    }
    gs_knn_brute_launch(s_in, Tapply, s_out, n_src, tgt, n_tgt, sc.best, st);
    GsProf prof(GS_PROF_LINEARIZE, 44.0 * (double)n_src, st);  // 8 B best + 12 B src + 24 B gather
    // the brute-force kernel wrote the transformed cloud to s_out (first search of an iteration)
    // or nothing (look-ahead: the row kernel re-applies Tr)
    if (full)
      hipLaunchKernelGGL((gs_icp_linearize_kernel<true>), dim3(nblk), dim3(LIN_BLOCK), 0, st, s_out, nullptr, n_src,
                         tgt, tgt_normals, n_tgt, sc.best, prm->dist_thresh, sc.partials, out_idx);
    else
      hipLaunchKernelGGL((gs_icp_linearize_kernel<false>), dim3(nblk), dim3(LIN_BLOCK), 0, st, s_in, Tapply, n_src,
                         tgt, tgt_normals, n_tgt, sc.best, prm->dist_thresh, sc.partials, nullptr);
  };
  const float* cur_in = src;   // source cloud before this iteration's pending transform
  float* cur = sc.srcA;        // where the transformed cloud of this iteration is written
  float* other = sc.srcB;
  for (int it = 0; it < prm->numiters; ++it) {
    // apply the pending transform (initial transform or last T_step) while searching
    half(true, cur_in, sc.state->T_step, cur);
    {
      GsProf prof(GS_PROF_SOLVE, 1.0, st);
      hipLaunchKernelGGL(gs_icp_solve_kernel, dim3(1), dim3(SUM_BLOCK), 0, st, sc.partials, nrows, sc.state);
    }
    // look-ahead: one_step = Tr * cur, searched and reduced without materialising it
    half(false, cur, sc.state->Tr, nullptr);
    {
      GsProf prof(GS_PROF_SOLVE, 1.0, st);
      hipLaunchKernelGGL(gs_icp_update_kernel, dim3(1), dim3(SUM_BLOCK), 0, st, sc.partials, nrows, sc.state, *prm, it,
                         compose16, out_T16);
    }
    cur_in = cur;
    float* t = cur; cur = other; other = t;
  }
  GS_LAUNCH_CHECK();
  return GS_OK;
}

extern "C" int gs_icp_trace_f32(const void* icp_scratch, int numiters, float* trace_out, void* stream) {
  GS_REQUIRE(icp_scratch && trace_out && numiters >= 0 && numiters <= 64, "bad arguments");
  const GsIcpState* st = reinterpret_cast<const GsIcpState*>(icp_scratch);
  GS_HIP(hipMemcpyAsync(trace_out, st->trace, sizeof(float) * 12 * (size_t)numiters, hipMemcpyDeviceToDevice,
                        gs_stream(stream)));
  return GS_OK;
}
