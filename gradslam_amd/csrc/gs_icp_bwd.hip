// gs_icp_bwd.hip — K7: backward pass of point_to_plane_gradICP (odometry/icputils.py:479-545) and of the
// hard-LM point_to_plane_ICP (odometry/icputils.py:310-367; mode 0: an accepted step applies exp(xi), a
// rejected one nothing, the accept test and the damping schedule are constants of the differentiation, as
// in the reference's autograd graph).
//
// Reverse-mode differentiation of the gradLM loop given the forward tape (gs_icp_tape_f32): the
// gradient of any scalar loss w.r.t. the source points, the target points, the target normals and
// the initial transform, from dL/dT.  Nearest-neighbour indices and the dist_thresh filter are
// constants of the differentiation exactly as in the reference (index / boolean ops are not
// differentiable, icputils.py:201-208); everything else (rows, normal equations, 6x6 solve, both
// SE(3) exponentials, the sigmoid damping and step scaling, the running transform) is
// differentiated analytically.  oracle/icp_backward.py is the float64 numpy restatement, pinned
// against the reference's own autograd (tests/golden/icp_grad.npz).
//
// Per iteration (last to first): scalar stage S1 (adjoint of T_step = exp(sigma xi), of sigma and
// of the damping) -> point stage P2 (look-ahead residual, scatter to targets, 12 sums for the
// adjoint of Tr) -> scalar stage S2 (adjoint of xi through exp and the 6x6 solve) -> point stage
// P3 (Gauss-Newton rows, scatter to targets, 12 sums for the next S1).  As in the forward loop the
// scalar stages are the PROLOGUE of the point kernel that follows them (every block adds up the
// previous kernel's partial rows and evaluates the stage redundantly, block 0 records the carried
// state), so an iteration is 2 launches, not 4; carried state and partial rows are double-buffered.
// All arithmetic in float64; scatters are float64 atomics (order-independent to ~1e-16, rounded once
// at the end) -- fast, but the order of the additions into a target that several source points share
// depends on scheduling, so two runs may differ in the last bit of a float32 gradient.
// GRADSLAM_HIP_DETERMINISTIC_BACKWARD=1 (round 5, VERDICT r04 #7b; read per call) makes the ICP backward bitwise reproducible:
// the point stages write every source point's contribution next to its target index, the pairs are
// sorted by target (stable radix sort: equal targets stay in source order) and one thread per target
// adds its contributions in that order.  About 2.5x the time of the atomic form.
#include <stdlib.h>

#include <hipcub/hipcub.hpp>

#include "gs_icp_math.h"

constexpr int BW_BLOCK = 256;
constexpr int BW_NV = 12;  // 3x3 outer-product sum + 3-vector sum

struct BwdCarry {      // what one kernel hands to the next
  double Tb[16];       // adjoint of the running transform
  double lam_bar;      // adjoint of the damping carried to the next (earlier) iteration
  double xi_bar[6];    // S1 -> S2
  double e_bar;        // S1 -> P3
};
struct BwdState {
  double Tk[65][16];  // running transform before iteration k (Tk[K] = final)
  double Ts[64][16];  // exp(sigma_k xi_k)
  double Tr[64][16];  // exp(xi_k)
  BwdCarry carry[2];  // A(k) reads [0] and writes [1]; B(k) reads [1] and writes [0]
};
struct BwdLocal {      // results of a prologue, shared through LDS
  double e1_bar, e_bar;
  double g_bar[6];
  double Hs[36];
};

GS_DEV void d_mm4(const double* A, const double* B, double* C) {
  double t[16];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double acc = 0.0;
      for (int k = 0; k < 4; ++k) acc += A[4 * i + k] * B[4 * k + j];
      t[4 * i + j] = acc;
    }
  for (int i = 0; i < 16; ++i) C[i] = t[i];
}

GS_DEV void d_hat(const double* w, double* wh) {
  wh[0] = 0; wh[1] = -w[2]; wh[2] = w[1];
  wh[3] = w[2]; wh[4] = 0; wh[5] = -w[0];
  wh[6] = -w[1]; wh[7] = w[0]; wh[8] = 0;
}
GS_DEV void d_mm3(const double* A, const double* B, double* C) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

// geometry/se3utils.py:77-115 in double
GS_DEV void d_se3_exp(const double* xi, double* T) {
  const double* v = xi;
  const double* w = xi + 3;
  double wh[9], wh2[9], R[9], V[9];
  d_hat(w, wh);
  const double th = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  if ((float)th < 1e-6f) {
    for (int i = 0; i < 9; ++i) { R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + wh[i]; V[i] = R[i]; }
  } else {
    d_mm3(wh, wh, wh2);
    const double s = sin(th), c = cos(th);
    const double A = s / th, B = (1 - c) / (th * th), Cc = (th - s) / (th * th * th);
    for (int i = 0; i < 9; ++i) {
      const double I = (i % 4 == 0) ? 1.0 : 0.0;
      R[i] = I + A * wh[i] + B * wh2[i];
      V[i] = I + B * wh[i] + Cc * wh2[i];
    }
  }
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) T[4 * i + j] = R[3 * i + j];
    T[4 * i + 3] = V[3 * i] * v[0] + V[3 * i + 1] * v[1] + V[3 * i + 2] * v[2];
  }
  T[12] = 0; T[13] = 0; T[14] = 0; T[15] = 1;
}

// d <Tbar, Exp(xi)> / d xi (Tbar: 4x4, only its top three rows matter)
GS_DEV void d_se3_exp_adjoint(const double* xi, const double* Tbar, double* out) {
  const double* v = xi;
  const double* w = xi + 3;
  double Rb[9], tb[3], wh[9], wh2[9], V[9], Vb[9], whb[9];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) Rb[3 * i + j] = Tbar[4 * i + j];
    tb[i] = Tbar[4 * i + 3];
  }
  d_hat(w, wh);
  d_mm3(wh, wh, wh2);
  const double th = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  const bool small = (float)th < 1e-6f;
  double s = 0, c = 1, A = 1, B = 0, Cc = 0;
  if (small) {
    for (int i = 0; i < 9; ++i) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + wh[i];
  } else {
    s = sin(th); c = cos(th);
    A = s / th; B = (1 - c) / (th * th); Cc = (th - s) / (th * th * th);
    for (int i = 0; i < 9; ++i) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + B * wh[i] + Cc * wh2[i];
  }
  for (int j = 0; j < 3; ++j) out[j] = V[j] * tb[0] + V[3 + j] * tb[1] + V[6 + j] * tb[2];  // V^T tb
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Vb[3 * i + j] = tb[i] * v[j];
  double thb = 0.0;
  if (small) {
    for (int i = 0; i < 9; ++i) whb[i] = Rb[i] + Vb[i];
  } else {
    double Ab = 0, Bb = 0, Cb = 0, W2b[9], whT[9], t1[9], t2[9];
    for (int i = 0; i < 9; ++i) {
      Ab += Rb[i] * wh[i];
      Bb += Rb[i] * wh2[i] + Vb[i] * wh[i];
      Cb += Vb[i] * wh2[i];
      W2b[i] = B * Rb[i] + Cc * Vb[i];
    }
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) whT[3 * i + j] = wh[3 * j + i];
    d_mm3(W2b, whT, t1);
    d_mm3(whT, W2b, t2);
    for (int i = 0; i < 9; ++i) whb[i] = A * Rb[i] + B * Vb[i] + t1[i] + t2[i];
    const double dA = (c * th - s) / (th * th);
    const double dB = (s * th - 2 * (1 - c)) / (th * th * th);
    const double dC = ((1 - c) * th - 3 * (th - s)) / (th * th * th * th);
    thb = Ab * dA + Bb * dB + Cb * dC;
  }
  out[3] = whb[7] - whb[5];
  out[4] = whb[2] - whb[6];
  out[5] = whb[3] - whb[1];
  if (!small)
    for (int j = 0; j < 3; ++j) out[3 + j] += thb * w[j] / th;
}

// 6x6 symmetric positive definite solve in double (un-pivoted Gauss-Jordan)
GS_DEV void d_solve6(const double* H, const double* rhs, double* x) {
  double a[6][7];
  for (int i = 0; i < 6; ++i) {
    for (int j = 0; j < 6; ++j) a[i][j] = H[6 * i + j];
    a[i][6] = rhs[i];
  }
  for (int c = 0; c < 6; ++c) {
    const double inv = 1.0 / a[c][c];
    for (int j = c; j <= 6; ++j) a[c][j] *= inv;
    for (int r = 0; r < 6; ++r) {
      if (r == c) continue;
      const double f = a[r][c];
      for (int j = c; j <= 6; ++j) a[r][j] -= f * a[c][j];
    }
  }
  for (int i = 0; i < 6; ++i) x[i] = a[i][6];
}

// block sum of BW_NV doubles per thread -> one partial row
GS_DEV void bw_block_reduce(const double* v, double* __restrict__ partial_row) {
  __shared__ double red[BW_BLOCK / GS_WAVE][BW_NV];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < BW_NV; ++i) {
    const double sum = gs_wave_sum_f64(v[i]);
    if (lane == 0) red[wave][i] = sum;
  }
  __syncthreads();
  if (threadIdx.x < BW_NV) {
    double t = 0.0;
    for (int w = 0; w < BW_BLOCK / GS_WAVE; ++w) t += red[w][threadIdx.x];
    partial_row[threadIdx.x] = t;
  }
}

// Adds up partial rows with the first wave of the block (12 values, rows strided over its 64 lanes, fixed
// order); every thread of the block must call it (it synchronises), every thread receives G.
GS_DEV void bw_sum_rows(const double* __restrict__ partials, int nrows, double* G) {
  __shared__ double acc[BW_NV];
  const int lane = threadIdx.x;
  if (lane < GS_WAVE) {
    for (int i = 0; i < BW_NV; ++i) {
      double s = 0.0;
      for (int b = lane; b < nrows; b += GS_WAVE) s += partials[(int64_t)b * BW_NV + i];
      s = gs_wave_sum_f64(s);
      if (lane == 0) acc[i] = s;
    }
  }
  __syncthreads();
  for (int i = 0; i < BW_NV; ++i) G[i] = acc[i];
  __syncthreads();
}

// Replays the forward transforms and seeds the adjoints.
__global__ void gs_bwd_init_kernel(BwdState* __restrict__ bs, GsIcpTape tape, const float* __restrict__ init16,
                                   const float* __restrict__ T_bar16, int K, int mode) {
  if (threadIdx.x != 0) return;
  for (int i = 0; i < 16; ++i) bs->Tk[0][i] = (double)init16[i];
  for (int k = 0; k < K; ++k) {
    double xi[6], xs[6];
    // mode 0 (hard LM): the step is exp(xi) when the look-ahead error dropped, the identity otherwise
    const double sig = mode == 0 ? (tape.trace[12 * k + 1] < tape.trace[12 * k] ? 1.0 : 0.0)
                                 : (double)tape.trace[12 * k + 3];
    for (int i = 0; i < 6; ++i) {
      xi[i] = (double)tape.trace[12 * k + 4 + i];
      xs[i] = sig * xi[i];
    }
    d_se3_exp(xi, bs->Tr[k]);
    d_se3_exp(xs, bs->Ts[k]);
    d_mm4(bs->Ts[k], bs->Tk[k], bs->Tk[k + 1]);
  }
  for (int i = 0; i < 16; ++i) bs->carry[0].Tb[i] = (double)T_bar16[i];
  bs->carry[0].lam_bar = 0.0;
  for (int i = 0; i < 6; ++i) bs->carry[0].xi_bar[i] = 0.0;
  bs->carry[0].e_bar = 0.0;
}

// S1 of iteration k (one lane): needs G = sum_i sbar_next_i (x) s_k,i and h = sum_i sbar_next_i
GS_DEV void bw_stage_s1(const BwdState* __restrict__ bs, const GsIcpTape& tape, int k, const double* G,
                        const gs_icp_params& prm, const BwdCarry& in, BwdCarry& out, BwdLocal& loc) {
  const double* Tk = bs->Tk[k];
  const double* Ts = bs->Ts[k];
  double Tb[16], Tsb[16];
  for (int i = 0; i < 16; ++i) Tb[i] = in.Tb[i];
  // Ts_bar = Tb Tk^T (+ the point-cloud part)
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double acc = 0.0;
      for (int m = 0; m < 4; ++m) acc += Tb[4 * i + m] * Tk[4 * j + m];
      Tsb[4 * i + j] = acc;
    }
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) Tsb[4 * i + j] += G[3 * i + j];
    Tsb[4 * i + 3] += G[9 + i];
  }
  // Tb <- Ts^T Tb
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double acc = 0.0;
      for (int m = 0; m < 4; ++m) acc += Ts[4 * m + i] * Tb[4 * m + j];
      out.Tb[4 * i + j] = acc;
    }
  double xi[6], xs[6], ub[6];
  if (prm.mode == 0) {
    // hard LM: T_step = exp(xi) if accepted (xi_bar = adjoint through exp), identity otherwise (nothing flows);
    // neither the accept test nor the damping schedule carries a gradient
    const bool accepted = tape.trace[12 * k + 1] < tape.trace[12 * k];
    for (int i = 0; i < 6; ++i) xi[i] = (double)tape.trace[12 * k + 4 + i];
    if (accepted) d_se3_exp_adjoint(xi, Tsb, ub);
    for (int i = 0; i < 6; ++i) out.xi_bar[i] = accepted ? ub[i] : 0.0;
    out.lam_bar = 0.0;
    loc.e1_bar = 0.0;
    loc.e_bar = 0.0;
    out.e_bar = 0.0;
    return;
  }
  const double sig = (double)tape.trace[12 * k + 3];
  for (int i = 0; i < 6; ++i) {
    xi[i] = (double)tape.trace[12 * k + 4 + i];
    xs[i] = sig * xi[i];
  }
  d_se3_exp_adjoint(xs, Tsb, ub);
  double sig_bar = 0.0;
  for (int i = 0; i < 6; ++i) {
    sig_bar += ub[i] * xi[i];
    out.xi_bar[i] = sig * ub[i];
  }
  const float err = tape.trace[12 * k], new_err = tape.trace[12 * k + 1];
  const double lam = (double)tape.sys[28 * k + 27];
  const float diff = new_err - err;
  const bool inside = diff >= -70.0f && diff <= 70.0f;
  const double d = (double)(diff < -70.0f ? -70.0f : (diff > 70.0f ? 70.0f : diff));
  const double lmin = (double)(float)(1.0 / (double)prm.lambda_max);
  const double lrange = (double)(float)((double)prm.lambda_max - 1.0 / (double)prm.lambda_max);
  const double Bp = (double)prm.B, B2p = (double)prm.B2, nu = (double)prm.nu;
  const double E = exp(-Bp * d), E2 = exp(-B2p * d);
  const double q = lmin + lrange / (1 + E);
  const double dq = lrange * Bp * E / ((1 + E) * (1 + E));
  const double dsig = (B2p * E2 / nu) * pow(1 + E2, -1.0 / nu - 1.0);
  double d_bar = sig_bar * dsig + in.lam_bar * lam * dq;
  out.lam_bar = in.lam_bar * q;
  if (!inside) d_bar = 0.0;
  loc.e1_bar = d_bar;
  loc.e_bar = -d_bar;
  out.e_bar = -d_bar;
}

// P2 of iteration k: sb_mid = Rs^T sbar_next + look-ahead residual part; scatter; sums for Tr_bar
__global__ void __launch_bounds__(BW_BLOCK) gs_bwd_p2_kernel(
    BwdState* __restrict__ bs, GsIcpTape tape, int k, gs_icp_params prm, const double* __restrict__ partials_in,
    int nrows_in, const float* __restrict__ src_k, const int32_t* __restrict__ idx1,
    int64_t n_src, const float* __restrict__ tgt, const float* __restrict__ tn, const double* __restrict__ sbar_next,
    double* __restrict__ sbar_mid, double* __restrict__ tgt_bar, double* __restrict__ tn_bar,
    double* __restrict__ partials, double* __restrict__ contrib, int32_t* __restrict__ ckey) {
  __shared__ BwdLocal loc;
  {  // prologue: S1 of this iteration, identical in every block
    double G[BW_NV];
    bw_sum_rows(partials_in, nrows_in, G);
    if (threadIdx.x == 0) {
      BwdCarry out;
      bw_stage_s1(bs, tape, k, G, prm, bs->carry[0], out, loc);
      if (blockIdx.x == 0) bs->carry[1] = out;
    }
    __syncthreads();
  }
  const int64_t i = (int64_t)blockIdx.x * BW_BLOCK + threadIdx.x;
  double v[BW_NV];
#pragma unroll
  for (int q = 0; q < BW_NV; ++q) v[q] = 0.0;
  if (i < n_src) {
    const double* Ts = bs->Ts[k];
    const double* Tr = bs->Tr[k];
    const double e1_bar = loc.e1_bar;
    const double s[3] = {(double)src_k[3 * i], (double)src_k[3 * i + 1], (double)src_k[3 * i + 2]};
    const double sn[3] = {sbar_next[3 * i], sbar_next[3 * i + 1], sbar_next[3 * i + 2]};
    double sb[3];
    for (int j = 0; j < 3; ++j) sb[j] = sn[0] * Ts[j] + sn[1] * Ts[4 + j] + sn[2] * Ts[8 + j];
    const int32_t j1 = idx1[i];
    if (j1 >= 0) {
      double s1[3], n1[3], d1[3];
      for (int a = 0; a < 3; ++a) {
        s1[a] = Tr[4 * a] * s[0] + Tr[4 * a + 1] * s[1] + Tr[4 * a + 2] * s[2] + Tr[4 * a + 3];
        n1[a] = (double)tn[3 * (int64_t)j1 + a];
        d1[a] = (double)tgt[3 * (int64_t)j1 + a];
      }
      const double b1 = n1[0] * (d1[0] - s1[0]) + n1[1] * (d1[1] - s1[1]) + n1[2] * (d1[2] - s1[2]);
      const double b1_bar = 2 * b1 * e1_bar;
      double s1b[3];
      for (int a = 0; a < 3; ++a) {
        s1b[a] = -n1[a] * b1_bar;
        if (contrib) {   // (deterministic mode: the contribution goes on record, bw_segment_add_kernel adds it up)
          contrib[6 * i + a] = n1[a] * b1_bar;
          contrib[6 * i + 3 + a] = (d1[a] - s1[a]) * b1_bar;
        } else {
          atomicAdd(&tgt_bar[3 * (int64_t)j1 + a], n1[a] * b1_bar);
          atomicAdd(&tn_bar[3 * (int64_t)j1 + a], (d1[a] - s1[a]) * b1_bar);
        }
      }
      for (int j = 0; j < 3; ++j) sb[j] += s1b[0] * Tr[j] + s1b[1] * Tr[4 + j] + s1b[2] * Tr[8 + j];
      for (int a = 0; a < 3; ++a) {
        for (int j = 0; j < 3; ++j) v[3 * a + j] = s1b[a] * s[j];
        v[9 + a] = s1b[a];
      }
    }
    for (int j = 0; j < 3; ++j) sbar_mid[3 * i + j] = sb[j];
    if (ckey) ckey[i] = j1 >= 0 ? j1 : 0x7fffffff;
  }
  bw_block_reduce(v, partials + (int64_t)blockIdx.x * BW_NV);
}

// S2 of iteration k (one lane): xi_bar += adjoint through Tr = exp(xi); then through xi = H^-1 g
GS_DEV void bw_stage_s2(const GsIcpTape& tape, int k, const double* G, const BwdCarry& in, BwdCarry& out,
                        BwdLocal& loc) {
  double Trb[16];
  for (int i = 0; i < 16; ++i) Trb[i] = 0.0;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) Trb[4 * i + j] = G[3 * i + j];
    Trb[4 * i + 3] = G[9 + i];
  }
  double xi[6], ub[6], xb[6], H[36], gb[6];
  for (int i = 0; i < 6; ++i) xi[i] = (double)tape.trace[12 * k + 4 + i];
  d_se3_exp_adjoint(xi, Trb, ub);
  for (int i = 0; i < 6; ++i) xb[i] = in.xi_bar[i] + ub[i];
  const float lam = tape.sys[28 * k + 27];
  int q = 0;
  for (int r = 0; r < 6; ++r)
    for (int c = r; c < 6; ++c) {
      H[6 * r + c] = H[6 * c + r] = (double)tape.sys[28 * k + q];
      ++q;
    }
  for (int r = 0; r < 6; ++r) H[6 * r + r] += (double)lam;
  d_solve6(H, xb, gb);
  double tr = 0.0;
  for (int r = 0; r < 6; ++r) {
    loc.g_bar[r] = gb[r];
    tr += -gb[r] * xi[r];
  }
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c < 6; ++c) loc.Hs[6 * r + c] = -(gb[r] * xi[c] + gb[c] * xi[r]);
  for (int i = 0; i < 16; ++i) out.Tb[i] = in.Tb[i];
  for (int i = 0; i < 6; ++i) out.xi_bar[i] = 0.0;
  out.lam_bar = in.lam_bar + tr;
  out.e_bar = in.e_bar;
  loc.e_bar = in.e_bar;
}

// P3 of iteration k: Gauss-Newton rows; sbar (adjoint of src_k); scatter; sums for the next S1
__global__ void __launch_bounds__(BW_BLOCK) gs_bwd_p3_kernel(
    BwdState* __restrict__ bs, GsIcpTape tape, int k, const double* __restrict__ partials_in, int nrows_in,
    const float* __restrict__ src_k, const float* __restrict__ src_prev,
    const int32_t* __restrict__ idx0, int64_t n_src, const float* __restrict__ tgt, const float* __restrict__ tn,
    const double* __restrict__ sbar_mid, double* __restrict__ sbar_out, double* __restrict__ tgt_bar,
    double* __restrict__ tn_bar, double* __restrict__ partials, double* __restrict__ contrib, int32_t* __restrict__ ckey) {
  __shared__ BwdLocal loc;
  {  // prologue: S2 of this iteration, identical in every block
    double G[BW_NV];
    bw_sum_rows(partials_in, nrows_in, G);
    if (threadIdx.x == 0) {
      BwdCarry out;
      bw_stage_s2(tape, k, G, bs->carry[1], out, loc);
      if (blockIdx.x == 0) bs->carry[0] = out;
    }
    __syncthreads();
  }
  const int64_t i = (int64_t)blockIdx.x * BW_BLOCK + threadIdx.x;
  double v[BW_NV];
#pragma unroll
  for (int q = 0; q < BW_NV; ++q) v[q] = 0.0;
  if (i < n_src) {
    const double s[3] = {(double)src_k[3 * i], (double)src_k[3 * i + 1], (double)src_k[3 * i + 2]};
    double sb[3] = {sbar_mid[3 * i], sbar_mid[3 * i + 1], sbar_mid[3 * i + 2]};
    const int32_t j = idx0[i];
    if (j >= 0) {
      const double e_bar = loc.e_bar;
      double n0[3], d0[3], a[6];
      for (int c = 0; c < 3; ++c) {
        n0[c] = (double)tn[3 * (int64_t)j + c];
        d0[c] = (double)tgt[3 * (int64_t)j + c];
      }
      a[0] = n0[0]; a[1] = n0[1]; a[2] = n0[2];
      a[3] = s[1] * n0[2] - s[2] * n0[1];
      a[4] = s[2] * n0[0] - s[0] * n0[2];
      a[5] = s[0] * n0[1] - s[1] * n0[0];
      const double b = n0[0] * (d0[0] - s[0]) + n0[1] * (d0[1] - s[1]) + n0[2] * (d0[2] - s[2]);
      double ab[6], b_bar = 2 * b * e_bar;
      for (int r = 0; r < 6; ++r) {
        double acc = loc.g_bar[r] * b;
        for (int c = 0; c < 6; ++c) acc += loc.Hs[6 * r + c] * a[c];
        ab[r] = acc;
        b_bar += a[r] * loc.g_bar[r];
      }
      const double* an = ab;
      const double* ac = ab + 3;
      // c = s x n: s_bar += n x c_bar, n_bar += c_bar x s ; b = n.(d - s)
      sb[0] += -n0[0] * b_bar + (n0[1] * ac[2] - n0[2] * ac[1]);
      sb[1] += -n0[1] * b_bar + (n0[2] * ac[0] - n0[0] * ac[2]);
      sb[2] += -n0[2] * b_bar + (n0[0] * ac[1] - n0[1] * ac[0]);
      const double cxs[3] = {ac[1] * s[2] - ac[2] * s[1], ac[2] * s[0] - ac[0] * s[2], ac[0] * s[1] - ac[1] * s[0]};
      for (int c = 0; c < 3; ++c) {
        if (contrib) {
          contrib[6 * i + c] = n0[c] * b_bar;
          contrib[6 * i + 3 + c] = an[c] + cxs[c] + (d0[c] - s[c]) * b_bar;
        } else {
          atomicAdd(&tn_bar[3 * (int64_t)j + c], an[c] + cxs[c] + (d0[c] - s[c]) * b_bar);
          atomicAdd(&tgt_bar[3 * (int64_t)j + c], n0[c] * b_bar);
        }
      }
    }
    if (ckey) ckey[i] = j >= 0 ? j : 0x7fffffff;
    for (int c = 0; c < 3; ++c) sbar_out[3 * i + c] = sb[c];
    // sums for the adjoint of the transform that produced src_k from src_prev
    const double sp[3] = {(double)src_prev[3 * i], (double)src_prev[3 * i + 1], (double)src_prev[3 * i + 2]};
    for (int a = 0; a < 3; ++a) {
      for (int c = 0; c < 3; ++c) v[3 * a + c] = sb[a] * sp[c];
      v[9 + a] = sb[a];
    }
  }
  bw_block_reduce(v, partials + (int64_t)blockIdx.x * BW_NV);
}

// Deterministic mode: the (target, source) pairs of a point stage sorted by target, equal targets in source order (the
// sort is stable and its input is in source order).  The thread at the head of a run of equal targets adds the run's
// contributions to that target, one after the other: a fixed order, and no other thread of any launch touches the
// target meanwhile (the launches of the backward are ordered).
__global__ void __launch_bounds__(BW_BLOCK) bw_iota_kernel(int32_t* __restrict__ v, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * BW_BLOCK + threadIdx.x;
  if (i < n) v[i] = (int32_t)i;
}
__global__ void __launch_bounds__(BW_BLOCK) bw_segment_add_kernel(const int32_t* __restrict__ key, const int32_t* __restrict__ src,
                                                                  int64_t n, const double* __restrict__ contrib,
                                                                  double* __restrict__ tgt_bar, double* __restrict__ tn_bar) {
  const int64_t p = (int64_t)blockIdx.x * BW_BLOCK + threadIdx.x;
  if (p >= n) return;
  const int32_t j = key[p];
  if (j == 0x7fffffff || (p > 0 && key[p - 1] == j)) return;   // filtered-out pair, or not the head of its run
  double t[3] = {0.0, 0.0, 0.0}, m[3] = {0.0, 0.0, 0.0};
  for (int64_t q = p; q < n && key[q] == j; ++q) {
    const double* c = contrib + 6 * (int64_t)src[q];
    for (int a = 0; a < 3; ++a) { t[a] += c[a]; m[a] += c[3 + a]; }
  }
  for (int a = 0; a < 3; ++a) {
    tgt_bar[3 * (int64_t)j + a] += t[a];
    tn_bar[3 * (int64_t)j + a] += m[a];
  }
}

// final: init_bar = T0_bar + sums ; src_bar = R_init^T sbar_0 ; float64 scatters -> float32
__global__ void __launch_bounds__(GS_WAVE) gs_bwd_final_scalar_kernel(const BwdState* __restrict__ bs,
                                                                      const double* __restrict__ partials, int nrows,
                                                                      float* __restrict__ init_bar16) {
  double G[BW_NV];
  bw_sum_rows(partials, nrows, G);
  if (threadIdx.x != 0 || !init_bar16) return;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) init_bar16[4 * i + j] = (float)(bs->carry[0].Tb[4 * i + j] + G[3 * i + j]);
    init_bar16[4 * i + 3] = (float)(bs->carry[0].Tb[4 * i + 3] + G[9 + i]);
  }
  for (int j = 0; j < 4; ++j) init_bar16[12 + j] = 0.0f;
}
__global__ void __launch_bounds__(BW_BLOCK) gs_bwd_src_out_kernel(const double* __restrict__ sbar0, int64_t n_src,
                                                                  const float* __restrict__ init16,
                                                                  float* __restrict__ src_bar) {
  const int64_t i = (int64_t)blockIdx.x * BW_BLOCK + threadIdx.x;
  if (i >= n_src) return;
  const double sb[3] = {sbar0[3 * i], sbar0[3 * i + 1], sbar0[3 * i + 2]};
  for (int j = 0; j < 3; ++j)
    src_bar[3 * i + j] = (float)(sb[0] * (double)init16[j] + sb[1] * (double)init16[4 + j] + sb[2] * (double)init16[8 + j]);
}
__global__ void __launch_bounds__(BW_BLOCK) gs_bwd_cast_kernel(const double* __restrict__ in, int64_t n,
                                                               float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * BW_BLOCK + threadIdx.x;
  if (i < n) out[i] = (float)in[i];
}

// ---------------------------------------------------------------- pose adjoint of the global maps ----
// gvertex = (R v + t) * valid, gnormal = R n (structures/rgbdimages.py:681-762):
//   R_bar = sum_p valid_p gv_bar_p (x) v_p + sum_p gn_bar_p (x) n_p,   t_bar = sum_p valid_p gv_bar_p
// 12 sums over the pixels: block partial rows, then one wave adds them up (fixed order, float64).
__global__ void __launch_bounds__(BW_BLOCK) gs_pose_bar_partial_kernel(
    const float* __restrict__ vertex, const float* __restrict__ normal, const float* __restrict__ depth,
    const float* __restrict__ gv_bar, const float* __restrict__ gn_bar, int64_t P, double* __restrict__ partials) {
  const int64_t p = (int64_t)blockIdx.x * BW_BLOCK + threadIdx.x;
  double v[BW_NV];
#pragma unroll
  for (int q = 0; q < BW_NV; ++q) v[q] = 0.0;
  if (p < P) {
    const bool valid = depth[p] > 0.0f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double gb = (gv_bar && valid) ? (double)gv_bar[3 * p + a] : 0.0;
      const double nb = gn_bar ? (double)gn_bar[3 * p + a] : 0.0;
#pragma unroll
      for (int c = 0; c < 3; ++c)
        v[3 * a + c] = gb * (double)vertex[3 * p + c] + nb * (normal ? (double)normal[3 * p + c] : 0.0);
      v[9 + a] = gb;
    }
  }
  bw_block_reduce(v, partials + (int64_t)blockIdx.x * BW_NV);
}
__global__ void __launch_bounds__(GS_WAVE) gs_pose_bar_final_kernel(const double* __restrict__ partials, int nrows,
                                                                    float* __restrict__ pose_bar16) {
  double G[BW_NV];
  bw_sum_rows(partials, nrows, G);
  if (threadIdx.x != 0) return;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) pose_bar16[4 * i + j] = (float)G[3 * i + j];
    pose_bar16[4 * i + 3] = (float)G[9 + i];
  }
  for (int j = 0; j < 4; ++j) pose_bar16[12 + j] = 0.0f;
}
// Reverse mode of gs_se3_exp_f32 alone (geometry/se3utils.py:77-115 under PyTorch autograd): xi_bar = d <T_bar, Exp(xi)> / d xi,
// the adjoint the ICP backward uses inside its scalar stages, float64 inside, rounded once.
__global__ void __launch_bounds__(GS_WAVE) gs_se3_exp_bwd_kernel(const float* __restrict__ xi6, const float* __restrict__ Tbar16,
                                                                 float* __restrict__ xi_bar6) {
  if (threadIdx.x != 0) return;
  double xi[6], Tb[16], out[6];
  for (int i = 0; i < 6; ++i) xi[i] = (double)xi6[i];
  for (int i = 0; i < 16; ++i) Tb[i] = (double)Tbar16[i];
  d_se3_exp_adjoint(xi, Tb, out);
  for (int i = 0; i < 6; ++i) xi_bar6[i] = (float)out[i];
}

extern "C" int gs_se3_exp_backward_f32(const float* xi6, const float* Tbar16, float* xi_bar6, void* stream) {
  GS_REQUIRE(xi6 && Tbar16 && xi_bar6, "NULL pointer");
  hipLaunchKernelGGL(gs_se3_exp_bwd_kernel, dim3(1), dim3(GS_WAVE), 0, gs_stream(stream), xi6, Tbar16, xi_bar6);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

extern "C" int64_t gs_global_maps_pose_backward_scratch_bytes(int H, int W) {
  return (int64_t)(gs_align(8 * BW_NV * (size_t)gs_ceil_div((int64_t)H * W, BW_BLOCK)) + 256);
}
extern "C" int gs_global_maps_pose_backward_f32(const float* vertex, const float* normal, const float* depth,
                                                const float* gvertex_bar, const float* gnormal_bar, int H, int W,
                                                float* pose_bar16, void* scratch, void* stream) {
  GS_REQUIRE(H > 0 && W > 0 && vertex && depth && pose_bar16 && scratch && (gvertex_bar || gnormal_bar), "bad arguments");
  GS_REQUIRE(!gnormal_bar || normal, "gnormal_bar needs the normal map");
  hipStream_t st = gs_stream(stream);
  const int64_t P = (int64_t)H * W;
  const int nblk = (int)gs_ceil_div(P, BW_BLOCK);
  double* partials = reinterpret_cast<double*>(scratch);
  hipLaunchKernelGGL(gs_pose_bar_partial_kernel, dim3(nblk), dim3(BW_BLOCK), 0, st, vertex, normal, depth, gvertex_bar,
                     gnormal_bar, P, partials);
  hipLaunchKernelGGL(gs_pose_bar_final_kernel, dim3(1), dim3(GS_WAVE), 0, st, partials, nblk, pose_bar16);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

struct BwdScratch {
  BwdState* state;
  double* sbar_a;   // [n_src][3]
  double* sbar_b;   // [n_src][3]
  double* tgt_bar;  // [n_tgt][3]
  double* tn_bar;   // [n_tgt][3]
  double* partials; // [nblk][12]   read by A(k) / final, written by B(k)
  double* partials2; // [nblk][12]  written by A(k), read by B(k)
  // deterministic mode (GRADSLAM_HIP_DETERMINISTIC_BACKWARD=1)
  double* contrib;   // [n_src][6] contribution of every source point of the current point stage to its target
  int32_t* key_in;   // [n_src] target of the pair (0x7fffffff: none)
  int32_t* key_out;
  int32_t* val_in;   // [n_src] 0, 1, 2, ...
  int32_t* val_out;
  void* sort_temp;
  size_t sort_temp_bytes;
};
static size_t bwd_sort_temp_bytes(int64_t n_src) { return gs_align(64 * (size_t)n_src + (1u << 20)); }
static BwdScratch bwd_carve(void* scratch, int64_t n_src, int64_t n_tgt) {
  char* p = reinterpret_cast<char*>(scratch);
  BwdScratch s;
  s.state = reinterpret_cast<BwdState*>(p); p += gs_align(sizeof(BwdState));
  s.sbar_a = reinterpret_cast<double*>(p); p += gs_align(24 * (size_t)n_src);
  s.sbar_b = reinterpret_cast<double*>(p); p += gs_align(24 * (size_t)n_src);
  s.tgt_bar = reinterpret_cast<double*>(p); p += gs_align(24 * (size_t)n_tgt);
  s.tn_bar = reinterpret_cast<double*>(p); p += gs_align(24 * (size_t)n_tgt);
  s.partials = reinterpret_cast<double*>(p); p += gs_align(8 * BW_NV * (size_t)gs_ceil_div(n_src, BW_BLOCK));
  s.partials2 = reinterpret_cast<double*>(p); p += gs_align(8 * BW_NV * (size_t)gs_ceil_div(n_src, BW_BLOCK));
  s.contrib = reinterpret_cast<double*>(p); p += gs_align(48 * (size_t)n_src);
  s.key_in = reinterpret_cast<int32_t*>(p); p += gs_align(4 * (size_t)n_src);
  s.key_out = reinterpret_cast<int32_t*>(p); p += gs_align(4 * (size_t)n_src);
  s.val_in = reinterpret_cast<int32_t*>(p); p += gs_align(4 * (size_t)n_src);
  s.val_out = reinterpret_cast<int32_t*>(p); p += gs_align(4 * (size_t)n_src);
  s.sort_temp = p;
  s.sort_temp_bytes = bwd_sort_temp_bytes(n_src);
  return s;
}

extern "C" int64_t gs_icp_backward_scratch_bytes(int64_t n_src, int64_t n_tgt) {
  if (n_src < 1) n_src = 1;
  if (n_tgt < 1) n_tgt = 1;
  return (int64_t)(gs_align(sizeof(BwdState)) + 2 * gs_align(24 * (size_t)n_src) + 2 * gs_align(24 * (size_t)n_tgt) +
                   2 * gs_align(8 * BW_NV * (size_t)gs_ceil_div(n_src, BW_BLOCK)) + gs_align(48 * (size_t)n_src) +
                   4 * gs_align(4 * (size_t)n_src) + bwd_sort_temp_bytes(n_src) + 4096);
}

extern "C" int gs_icp_backward_f32(const void* tape, const float* src_in, int64_t n_src, const float* tgt,
                                   const float* tgt_normals, int64_t n_tgt, const float* init16,
                                   const gs_icp_params* prm, const float* T_bar16, float* src_bar, float* tgt_bar,
                                   float* normals_bar, float* init_bar16, void* scratch, void* stream) {
  GS_REQUIRE(prm && tape && src_in && tgt && tgt_normals && init16 && T_bar16 && scratch, "NULL pointer");
  GS_REQUIRE(n_src > 0 && n_tgt > 0, "empty point set");
  GS_REQUIRE(prm->mode == 0 || prm->mode == 1, "mode must be 0 (ICP) or 1 (gradICP)");
  GS_REQUIRE(prm->numiters >= 0 && prm->numiters <= 64, "numiters must be in [0, 64]");
  hipStream_t st = gs_stream(stream);
  const int K = prm->numiters;
  GsIcpTape tp = gs_icp_tape_carve(const_cast<void*>(tape), n_src, K);
  BwdScratch sc = bwd_carve(scratch, n_src, n_tgt);
  const int nblk = (int)gs_ceil_div(n_src, BW_BLOCK);
  GS_HIP(hipMemsetAsync(sc.sbar_a, 0, 24 * (size_t)n_src, st));
  GS_HIP(hipMemsetAsync(sc.tgt_bar, 0, 24 * (size_t)n_tgt, st));
  GS_HIP(hipMemsetAsync(sc.tn_bar, 0, 24 * (size_t)n_tgt, st));
  GS_HIP(hipMemsetAsync(sc.partials, 0, 8 * BW_NV * (size_t)nblk, st));  // sbar_next = 0 for the last iteration
  hipLaunchKernelGGL(gs_bwd_init_kernel, dim3(1), dim3(64), 0, st, sc.state, tp, init16, T_bar16, K, prm->mode);
  double* sbar_next = sc.sbar_a;  // adjoint of src_{k+1}
  double* sbar_mid = sc.sbar_b;
  // (read on every call -- the backward is milliseconds, the lookup nanoseconds -- so that a process can switch the mode
  // between two backwards, ADVICE r05.  Scope: the scatter-adds of THIS function, the ICP backward; the fusion and
  // frame-map backwards have no competing additions into one word.  The sort scratch is part of the backward scratch in
  // either mode: gs_icp_backward_scratch_bytes is called before the mode of the run that uses it is known.)
  const char* det_env = getenv("GRADSLAM_HIP_DETERMINISTIC_BACKWARD");
  const int deterministic = (det_env && atoi(det_env) != 0) ? 1 : 0;
  double* contrib = deterministic ? sc.contrib : nullptr;
  int32_t* ckey = deterministic ? sc.key_in : nullptr;
  if (deterministic) {
    hipLaunchKernelGGL(bw_iota_kernel, dim3(nblk), dim3(BW_BLOCK), 0, st, sc.val_in, n_src);
    size_t need = 0;
    GS_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, need, sc.key_in, sc.key_out, sc.val_in, sc.val_out, (int)n_src, 0, 31, st));   // (31 key bits: target indices and the 0x7fffffff of a pair without one)
    GS_REQUIRE(need <= sc.sort_temp_bytes, "sort scratch too small");
  }
  // adds the recorded contributions of the point stage just enqueued to the targets, in (target, source) order
  auto settle = [&]() -> int {
    size_t tb = sc.sort_temp_bytes;
    GS_HIP(hipcub::DeviceRadixSort::SortPairs(sc.sort_temp, tb, sc.key_in, sc.key_out, sc.val_in, sc.val_out, (int)n_src, 0, 31, st));
    hipLaunchKernelGGL(bw_segment_add_kernel, dim3(nblk), dim3(BW_BLOCK), 0, st, sc.key_out, sc.val_out, n_src, sc.contrib,
                       sc.tgt_bar, sc.tn_bar);
    return GS_OK;
  };
  for (int k = K - 1; k >= 0; --k) {
    const float* src_k = tp.src + (size_t)k * 3 * (size_t)n_src;
    const float* src_prev = (k > 0) ? tp.src + (size_t)(k - 1) * 3 * (size_t)n_src : src_in;
    const int32_t* idx0 = tp.idx + ((size_t)k * 2) * (size_t)n_src;
    const int32_t* idx1 = idx0 + (size_t)n_src;
    // A(k) = S1 + P2: reads partials / carry[0], writes partials2 / carry[1]
    hipLaunchKernelGGL(gs_bwd_p2_kernel, dim3(nblk), dim3(BW_BLOCK), 0, st, sc.state, tp, k, *prm, sc.partials, nblk,
                       src_k, idx1, n_src, tgt, tgt_normals, sbar_next, sbar_mid, sc.tgt_bar, sc.tn_bar, sc.partials2, contrib,
                       ckey);
    if (deterministic) { const int rc = settle(); if (rc != GS_OK) return rc; }
    // B(k) = S2 + P3: reads partials2 / carry[1], writes partials / carry[0]; P3 overwrites sbar_next with the
    // adjoint of src_k (it only reads sbar_mid)
    hipLaunchKernelGGL(gs_bwd_p3_kernel, dim3(nblk), dim3(BW_BLOCK), 0, st, sc.state, tp, k, sc.partials2, nblk, src_k,
                       src_prev, idx0, n_src, tgt, tgt_normals, sbar_mid, sbar_next, sc.tgt_bar, sc.tn_bar, sc.partials, contrib,
                       ckey);
    if (deterministic) { const int rc = settle(); if (rc != GS_OK) return rc; }
  }
  if (K == 0) {  // T = init: sums stay zero, Tb = T_bar
    GS_HIP(hipMemsetAsync(sc.partials, 0, 8 * BW_NV * (size_t)nblk, st));
  }
  hipLaunchKernelGGL(gs_bwd_final_scalar_kernel, dim3(1), dim3(GS_WAVE), 0, st, sc.state, sc.partials, nblk, init_bar16);
  if (src_bar)
    hipLaunchKernelGGL(gs_bwd_src_out_kernel, dim3(nblk), dim3(BW_BLOCK), 0, st, sbar_next, n_src, init16, src_bar);
  if (tgt_bar)
    hipLaunchKernelGGL(gs_bwd_cast_kernel, dim3((unsigned)gs_ceil_div(3 * n_tgt, BW_BLOCK)), dim3(BW_BLOCK), 0, st,
                       sc.tgt_bar, 3 * n_tgt, tgt_bar);
  if (normals_bar)
    hipLaunchKernelGGL(gs_bwd_cast_kernel, dim3((unsigned)gs_ceil_div(3 * n_tgt, BW_BLOCK)), dim3(BW_BLOCK), 0, st,
                       sc.tn_bar, 3 * n_tgt, normals_bar);
  GS_LAUNCH_CHECK();
  return GS_OK;
}
