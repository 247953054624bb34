// gs_icp_math.h — device-side pieces shared by the API-level ICP kernels (gs_icp.hip) and the
// device-resident LM loop (gs_icp_loop.hip): Gauss-Newton row, SPD solve, SE(3) exponential, 4x4
// algebra, and the two scalar stages of an LM iteration.
#pragma once
#include "gs_common.h"

constexpr int LIN_NV = 28;  // 21 upper-triangular JtJ + 6 Jtr + 1 rtr

// 16 bytes per lane from global memory straight into LDS: lane l of the wave lands at lds_wave_base + 16 l (the LDS
// base must be wave-uniform; disabled lanes load nothing)
GS_DEV void it_load_lds16(const void* g, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

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

// the same row from a target point / normal already in registers (the binned copies): same operations, same order
GS_DEV void gn_row_pn(float sx, float sy, float sz, const float4 d, const float4 n, float* a, float& b) {
  const float dx = d.x, dy = d.y, dz = d.z;
  const float nx = n.x, ny = n.y, nz = n.z;
  a[0] = nx; a[1] = ny; a[2] = nz;
  a[3] = nz * sy - ny * sz;
  a[4] = nx * sz - nz * sx;
  a[5] = ny * sx - nx * sy;
  const float t = nx * (dx - sx) + ny * (dy - sy);
  b = t + nz * (dz - sz);
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

// ---------------------------------------------------------------- fast scalar functions --
// The scalar stages of the LM loop run on ONE lane and sit on the critical path of every
// half-iteration; libm's general-range double sin / cos / exp / pow are several hundred dependent
// instructions each.  These versions cover the ranges the loop actually visits with short Horner
// polynomials accurate to ~1 ulp(double) (results are rounded to float32 afterwards, so they match
// the oracle's libm values) and fall back to libm outside.
GS_DEV void gs_sincos_fast(double x, double* s, double* c) {
  if (!(fabs(x) <= 1.0)) {
    *s = sin(x);
    *c = cos(x);
    return;
  }
  const double z = x * x;
  // Taylor in z = x^2: terms up to x^21 / x^22 (next term < 1e-21)
  double ps = -1.0 / 51090942171709440000.0;   // -1/21!
  ps = ps * z + 1.0 / 121645100408832000.0;    // 1/19!
  ps = ps * z - 1.0 / 355687428096000.0;       // 1/17!
  ps = ps * z + 1.0 / 1307674368000.0;         // 1/15!
  ps = ps * z - 1.0 / 6227020800.0;            // 1/13!
  ps = ps * z + 1.0 / 39916800.0;              // 1/11!
  ps = ps * z - 1.0 / 362880.0;                // 1/9!
  ps = ps * z + 1.0 / 5040.0;                  // 1/7!
  ps = ps * z - 1.0 / 120.0;                   // 1/5!
  ps = ps * z + 1.0 / 6.0;                     // 1/3!
  *s = x - x * z * ps;
  double pc = 1.0 / 1124000727777607680000.0;  // 1/22!
  pc = pc * z - 1.0 / 2432902008176640000.0;   // 1/20!
  pc = pc * z + 1.0 / 6402373705728000.0;      // 1/18!
  pc = pc * z - 1.0 / 20922789888000.0;        // 1/16!
  pc = pc * z + 1.0 / 87178291200.0;           // 1/14!
  pc = pc * z - 1.0 / 479001600.0;             // 1/12!
  pc = pc * z + 1.0 / 3628800.0;               // 1/10!
  pc = pc * z - 1.0 / 40320.0;                 // 1/8!
  pc = pc * z + 1.0 / 720.0;                   // 1/6!
  pc = pc * z - 1.0 / 24.0;                    // 1/4!
  pc = pc * z + 0.5;                           // 1/2!
  *c = 1.0 - z * pc;
}

GS_DEV double gs_exp_fast(double x) {
  if (!(fabs(x) <= 700.0)) return exp(x);
  const double k = rint(x * 1.4426950408889634074);
  // ln2 split so that k * hi is exact for |k| < 2^11
  double r = x - k * 6.93147180369123816490e-01;
  r = r - k * 1.90821492927058770002e-10;
  // Taylor on |r| <= 0.3466: terms up to r^14 / 14! (< 5e-18)
  double p = 1.0 / 87178291200.0;
  p = p * r + 1.0 / 6227020800.0;
  p = p * r + 1.0 / 479001600.0;
  p = p * r + 1.0 / 39916800.0;
  p = p * r + 1.0 / 3628800.0;
  p = p * r + 1.0 / 362880.0;
  p = p * r + 1.0 / 40320.0;
  p = p * r + 1.0 / 5040.0;
  p = p * r + 1.0 / 720.0;
  p = p * r + 1.0 / 120.0;
  p = p * r + 1.0 / 24.0;
  p = p * r + 1.0 / 6.0;
  p = p * r + 0.5;
  p = p * r + 1.0;
  p = p * r + 1.0;
  return ldexp(p, (int)k);
}

GS_DEV double gs_log_fast(double y) {
  if (!(y > 1e-300 && y < 1e300)) return log(y);
  int e;
  double m = frexp(y, &e);  // y = m * 2^e, m in [0.5, 1)
  if (m < 0.70710678118654752440) {
    m *= 2.0;
    e -= 1;
  }
  const double z = (m - 1.0) / (m + 1.0), z2 = z * z;  // |z| <= 0.1716
  double p = 1.0 / 23.0;
  p = p * z2 + 1.0 / 21.0;
  p = p * z2 + 1.0 / 19.0;
  p = p * z2 + 1.0 / 17.0;
  p = p * z2 + 1.0 / 15.0;
  p = p * z2 + 1.0 / 13.0;
  p = p * z2 + 1.0 / 11.0;
  p = p * z2 + 1.0 / 9.0;
  p = p * z2 + 1.0 / 7.0;
  p = p * z2 + 1.0 / 5.0;
  p = p * z2 + 1.0 / 3.0;
  p = p * z2 + 1.0;
  return 2.0 * z * p + (double)e * 0.69314718055994530942;
}

// geometry/se3utils.py:77-115 in double, rounded once (same order as the oracle).  Written row by
// row (each entry: the same operations in the same order as the matrix form) so that only a handful
// of doubles are live at a time: this runs on one lane inside kernels capped at 80 VGPRs.
GS_DEV void gs_se3_exp_dev(const float* xi6, float* T16) {
  const double v[3] = {xi6[0], xi6[1], xi6[2]}, w[3] = {xi6[3], xi6[4], xi6[5]};
  const double theta = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  const bool small = (float)theta < 1e-6f;
  double Ac = 0.0, Bc = 0.0, Cc = 0.0;
  if (!small) {
    double sn, cs;
    gs_sincos_fast(theta, &sn, &cs);
    Ac = sn / theta; Bc = (1 - cs) / (theta * theta); Cc = (theta - sn) / (theta * theta * theta);
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    // row i of the hat matrix: wh[i][j]
    const double whi[3] = {i == 0 ? 0.0 : (i == 1 ? w[2] : -w[1]), i == 0 ? -w[2] : (i == 1 ? 0.0 : w[0]),
                           i == 0 ? w[1] : (i == 1 ? -w[0] : 0.0)};
    double Ri[3], Vi[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const double I = (i == j) ? 1.0 : 0.0;
      if (small) {
        Ri[j] = I + whi[j];
        Vi[j] = Ri[j];
      } else {
        // wh2[i][j] = sum_k wh[i][k] * wh[k][j], k ascending, starting from 0
        double s2 = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const double wkj = (k == j) ? 0.0
                             : (k == 0 ? (j == 1 ? -w[2] : w[1]) : (k == 1 ? (j == 0 ? w[2] : -w[0]) : (j == 0 ? -w[1] : w[0])));
          s2 += whi[k] * wkj;
        }
        Ri[j] = I + Ac * whi[j] + Bc * s2;
        Vi[j] = I + Bc * whi[j] + Cc * s2;
      }
      T16[4 * i + j] = (float)Ri[j];
    }
    T16[4 * i + 3] = (float)(Vi[0] * v[0] + Vi[1] * v[1] + Vi[2] * v[2]);
  }
  T16[12] = 0; T16[13] = 0; T16[14] = 0; T16[15] = 1;
}

// The same exponential spread over the lanes of ONE wave (round 4: the scalar stage of the half-iteration kernels is
// 2 us of one lane's float64 code on the critical path of every launch).  Every lane computes the common scalars (theta,
// sin, cos, A, B, C: the same operations, so the same bits); lane i < 3 then computes row i of R and V -- every entry sees
// exactly the operations of gs_se3_exp_dev in the same order (the hat matrix is materialised, so that the products with
// its zeros are the same products) -- and lane 3 the constant row.  T16 may live in LDS; all 64 lanes must call it.
GS_DEV void gs_se3_exp_wave(const float* xi6, float* T16, const int lane) {
  const double v[3] = {xi6[0], xi6[1], xi6[2]}, w[3] = {xi6[3], xi6[4], xi6[5]};
  const double theta = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  const bool small = (float)theta < 1e-6f;
  double Ac = 0.0, Bc = 0.0, Cc = 0.0;
  if (!small) {
    double sn, cs;
    gs_sincos_fast(theta, &sn, &cs);
    Ac = sn / theta; Bc = (1 - cs) / (theta * theta); Cc = (theta - sn) / (theta * theta * theta);
  }
  // hat matrix wh[k][j] (the values gs_se3_exp_dev selects entry by entry)
  const double wh[3][3] = {{0.0, -w[2], w[1]}, {w[2], 0.0, -w[0]}, {-w[1], w[0], 0.0}};
  if (lane < 3) {
    const int i = lane;
    double whi[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) whi[j] = i == 0 ? wh[0][j] : (i == 1 ? wh[1][j] : wh[2][j]);
    double Vi[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const double I = (i == j) ? 1.0 : 0.0;
      double Rij;
      if (small) {
        Rij = I + whi[j];
        Vi[j] = Rij;
      } else {
        double s2 = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) s2 += whi[k] * wh[k][j];
        Rij = I + Ac * whi[j] + Bc * s2;
        Vi[j] = I + Bc * whi[j] + Cc * s2;
      }
      T16[4 * i + j] = (float)Rij;
    }
    T16[4 * i + 3] = (float)(Vi[0] * v[0] + Vi[1] * v[1] + Vi[2] * v[2]);
  } else if (lane == 3) {
    T16[12] = 0; T16[13] = 0; T16[14] = 0; T16[15] = 1;
  }
}

// LDS traffic of one wave is ordered; these keep the compiler from reordering it (all lanes of the wave call them)
GS_DEV void gs_wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// gs_mm4(A, B, C = B) by 16 lanes of one wave on operands in LDS: lane e computes entry e with the operations of gs_mm4
// (plain multiply / add, k ascending); every entry is read before any is written.  All 64 lanes must call it.
GS_DEV void gs_mm4_wave_inplace(const float* A, float* B, const int lane) {
  float acc = 0.0f;
  if (lane < 16) {
    const int i = lane >> 2, j = lane & 3;
    acc = A[4 * i] * B[j];
#pragma unroll
    for (int k = 1; k < 4; ++k) acc = acc + A[4 * i + k] * B[4 * k + j];
  }
  gs_wave_sync_lds();
  if (lane < 16) B[lane] = acc;
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

// ---------------------------------------------------------------- LM iteration, scalar stages
// State carried between the kernels of one ICP solve.
struct IcpSmall {
  float T_total[16];
  float Tr[16];      // residual transform of the current iteration: se3_exp(xi)
  float T_step[16];  // transform applied to the source cloud at the end of the iteration
  float xi[8];
  float damp;
  float err;
  float pad[2];
};

// gs_solve_spd<6> spread over one wave: lane l < 42 owns a[r][j] (r = l / 7, j = l % 7) of the augmented
// matrix; every element sees exactly the operations of the serial code, in the same order, so x is
// bit-identical -- but the 84 VGPRs of a[6][7] in one lane become one double per lane and the 6 pivot
// steps cost a division and three shuffles each.  All 64 lanes of the wave must call it.
// S = the 28 sums (21 upper-triangular AtA, 6 Atb, rtr); x_out receives the 6 solution components.
GS_DEV void gs_solve_spd6_wave(const double* S, float damp, float* x_out) {
  const int l = threadIdx.x & 63, r = l / 7, j = l % 7;
  const bool active = l < 42;
  double a = 0.0;
  if (active) {
    if (j < 6) {
      const int lo = r < j ? r : j, hi = r < j ? j : r;
      const float v = (float)S[lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo)];
      const float e = (r == j) ? 1.0f : 0.0f;
      a = (double)(v + e * damp);  // At_A + damp_matrix * damp, in float32
    } else {
      a = (double)(float)S[21 + r];
    }
  }
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    const double piv = __shfl(a, c * 7 + c, 64);
    const double rowv = __shfl(a, c * 7 + j, 64);
    const double f = __shfl(a, (r < 6 ? r : 0) * 7 + c, 64);
    const double inv = 1.0 / piv;
    if (active && j >= c) {
      const double sc = rowv * inv;
      a = (r == c) ? sc : a - f * sc;
    }
  }
  if (active && j == 6) x_out[r] = (float)a;
}

// After the first linearisation of an iteration (S = the 28 normal-equation sums, float64):
// err = r.r, xi = (AtA + damp I)^-1 Atb, Tr = se3_exp(xi)   (odometry/icputils.py:328-337 / :498-507)
// Two-step form used by the kernels: gs_solve_spd6_wave(S, s.damp, s.xi) by one wave, a barrier, then
// icp_solve_finish by one lane.  icp_solve_math is the same computation on a single lane.
GS_DEV void icp_solve_finish(const double* S, IcpSmall& s) {
  s.err = (float)S[27];
  gs_se3_exp_dev(s.xi, s.Tr);
}
GS_DEV void icp_solve_math(const double* S, IcpSmall& s) {
  float AtA[36], Atb[6], xi[6];
  int q = 0;
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int k = i; k < 6; ++k) {
      AtA[6 * i + k] = AtA[6 * k + i] = (float)S[q];
      ++q;
    }
#pragma unroll
  for (int i = 0; i < 6; ++i) Atb[i] = (float)S[21 + i];
  s.err = (float)S[27];
  gs_solve_spd<6>(AtA, Atb, s.damp, xi);
  gs_se3_exp_dev(xi, s.Tr);
#pragma unroll
  for (int i = 0; i < 6; ++i) s.xi[i] = xi[i];
}

// After the look-ahead residual (new_err = r'.r'): LM accept / reject (mode 0,
// odometry/icputils.py:356-365) or the gradLM soft update (mode 1, :527-543).  Sets T_step,
// T_total, damp; trace_row (12 floats, may be NULL) = [err, new_err, damp, sigmoid, xi(6), 0, 0].
GS_DEV void icp_update_math(float new_err, IcpSmall& s, const gs_icp_params& prm, float* trace_row) {
  // works in place on `s` (LDS in the fused kernels): T_step is written where it lives and T_total is
  // updated through gs_mm4's 16-float temporary, so few values are live at a time
  const float err = s.err;
  float damp = s.damp;
  float sig = 1.0f;
  if (prm.mode == 0) {
    if (new_err < err) {
#pragma unroll
      for (int i = 0; i < 16; ++i) s.T_step[i] = s.Tr[i];
      damp = damp / 2;
      gs_mm4(s.T_step, s.T_total, s.T_total);
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) s.T_step[i] = (i % 5 == 0) ? 1.0f : 0.0f;
      damp = damp * 2;
    }
  } else {
    const float lmin = (float)(1.0 / (double)prm.lambda_max);
    const float lrange = (float)((double)prm.lambda_max - 1.0 / (double)prm.lambda_max);
    float errdiff = new_err - err;
    errdiff = errdiff < -70.0f ? -70.0f : (errdiff > 70.0f ? 70.0f : errdiff);
    const float e_b = (float)gs_exp_fast((double)((float)(-(double)prm.B) * errdiff));
    const float damp_new = lmin + lrange / (1.0f + e_b);
    damp = damp * damp_new;
    const float e_b2 = (float)gs_exp_fast((double)((float)(-(double)prm.B2) * errdiff));
    const float pw = (float)gs_exp_fast((double)(float)(1.0 / (double)prm.nu) * gs_log_fast((double)(1.0f + e_b2)));
    sig = 1.0f / pw;
    float xs[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) xs[k] = sig * s.xi[k];
    gs_se3_exp_dev(xs, s.T_step);
    gs_mm4(s.T_step, s.T_total, s.T_total);
  }
  s.damp = damp;
  if (trace_row) {
    trace_row[0] = err; trace_row[1] = new_err; trace_row[2] = damp; trace_row[3] = sig;
#pragma unroll
    for (int k = 0; k < 6; ++k) trace_row[4 + k] = s.xi[k];
    trace_row[10] = 0; trace_row[11] = 0;
  }
}

// The two scalar stages by ONE WAVE on a state that lives in LDS (the half-iteration kernels): the scalars are computed
// by every lane (same operations, same bits), the 12 + 16 matrix entries by one lane each.  All 64 lanes must call them;
// the caller publishes `s` with a workgroup barrier afterwards.
GS_DEV void icp_solve_finish_wave(const double* S, IcpSmall& s, const int lane) {
  gs_wave_sync_lds();   // (s.xi was written by gs_solve_spd6_wave's lanes)
  gs_se3_exp_wave(s.xi, s.Tr, lane);
  if (lane == 0) s.err = (float)S[27];
}
GS_DEV void icp_update_math_wave(float new_err, IcpSmall& s, const gs_icp_params& prm, float* trace_row, const int lane) {
  const float err = s.err;
  float damp = s.damp;
  float sig = 1.0f;
  float xi[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) xi[k] = s.xi[k];
  if (prm.mode == 0) {
    const bool accept = new_err < err;   // (the same for every lane)
    float tr = 0.0f;
    if (lane < 16) tr = s.Tr[lane];
    gs_wave_sync_lds();
    if (lane < 16) s.T_step[lane] = accept ? tr : ((lane % 5 == 0) ? 1.0f : 0.0f);
    damp = accept ? damp / 2 : damp * 2;
    gs_wave_sync_lds();
    if (accept) gs_mm4_wave_inplace(s.T_step, s.T_total, lane);
  } else {
    const float lmin = (float)(1.0 / (double)prm.lambda_max);
    const float lrange = (float)((double)prm.lambda_max - 1.0 / (double)prm.lambda_max);
    float errdiff = new_err - err;
    errdiff = errdiff < -70.0f ? -70.0f : (errdiff > 70.0f ? 70.0f : errdiff);
    const float e_b = (float)gs_exp_fast((double)((float)(-(double)prm.B) * errdiff));
    const float damp_new = lmin + lrange / (1.0f + e_b);
    damp = damp * damp_new;
    const float e_b2 = (float)gs_exp_fast((double)((float)(-(double)prm.B2) * errdiff));
    const float pw = (float)gs_exp_fast((double)(float)(1.0 / (double)prm.nu) * gs_log_fast((double)(1.0f + e_b2)));
    sig = 1.0f / pw;
    float xs[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) xs[k] = sig * xi[k];
    gs_se3_exp_wave(xs, s.T_step, lane);
    gs_wave_sync_lds();
    gs_mm4_wave_inplace(s.T_step, s.T_total, lane);
  }
  if (lane == 0) {
    s.damp = damp;
    if (trace_row) {
      trace_row[0] = err; trace_row[1] = new_err; trace_row[2] = damp; trace_row[3] = sig;
#pragma unroll
      for (int k = 0; k < 6; ++k) trace_row[4 + k] = xi[k];
      trace_row[10] = 0; trace_row[11] = 0;
    }
  }
}

// final result: T_total, optionally composed with the caller's pose (slam/icpslam.py:245-247)
GS_DEV void icp_write_result(const IcpSmall& s, const float* __restrict__ compose16, float* __restrict__ out_T16) {
  float out[16];
  if (compose16) {
    float Cm[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) Cm[i] = compose16[i];
    gs_compose_rigid(s.T_total, Cm, out);
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) out[i] = s.T_total[i];
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) out_T16[i] = out[i];
}

// ---------------------------------------------------------------- forward tape ----------
// Everything the backward pass needs from a gradICP forward, in one caller-owned buffer:
//   trace [K][12] float  (err, new_err, damp_after, sigmoid, xi(6), 0, 0)
//   sys   [K][28] float  (21 upper-triangular AtA, 6 Atb, damping before the iteration)
//   src   [K][n_src][3] float  source cloud at the start of each iteration
//   idx   [K][2][n_src] int32  neighbour of the first / look-ahead search, -1 if filtered out
struct GsIcpTape {
  float* trace;
  float* sys;
  float* src;
  int32_t* idx;
};
static inline size_t gs_icp_tape_size(int64_t n_src, int numiters) {
  const size_t K = (size_t)(numiters > 0 ? numiters : 1), n = (size_t)(n_src > 0 ? n_src : 1);
  return gs_align(4 * 12 * K) + gs_align(4 * 28 * K) + gs_align(12 * K * n) + gs_align(8 * K * n) + 256;
}
static inline GsIcpTape gs_icp_tape_carve(void* tape, int64_t n_src, int numiters) {
  const size_t K = (size_t)(numiters > 0 ? numiters : 1), n = (size_t)(n_src > 0 ? n_src : 1);
  char* p = reinterpret_cast<char*>(tape);
  GsIcpTape t;
  t.trace = reinterpret_cast<float*>(p); p += gs_align(4 * 12 * K);
  t.sys = reinterpret_cast<float*>(p); p += gs_align(4 * 28 * K);
  t.src = reinterpret_cast<float*>(p); p += gs_align(12 * K * n);
  t.idx = reinterpret_cast<int32_t*>(p);
  return t;
}
