// gs_assoc_dev.h — device helpers of the projective association shared by gs_assoc.hip (table-level and
// fused association) and gs_fuse.hip (the one-call map update): map point -> pixel, the similarity test and
// the per-pixel ordering key.
#pragma once
#include "gs_common.h"

// ---------------------------------------------------------------- K5a: projection ------
struct GsCamera {
  float Ri[9];  // R^T
  float ti[3];  // -R^T t   (kornia inverse_transformation: tiny matmul, plain arithmetic)
  float K[12];  // rows 0..2 of the 4x4 intrinsics
};

GS_DEV GsCamera gs_camera(const float* __restrict__ pose16, const float* __restrict__ K16) {
  GsCamera c;
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int k = 0; k < 3; ++k) c.Ri[3 * j + k] = pose16[4 * k + j];
  const float t0 = pose16[3], t1 = pose16[7], t2 = pose16[11];
#pragma unroll
  for (int j = 0; j < 3; ++j)
    c.ti[j] = gs_dot3_plain(-c.Ri[3 * j], -c.Ri[3 * j + 1], -c.Ri[3 * j + 2], t0, t1, t2);
#pragma unroll
  for (int i = 0; i < 12; ++i) c.K[i] = K16[i];
  return c;
}

// slam/fusionutils.py:250-274 for one point: pixel (h, w) it projects to; false when it falls outside the frame or
// behind the camera.  32-bit integer arithmetic throughout: in_frame bounds u, v to (-1e-3, W - 0.999) x (-1e-3,
// H - 0.999), so the rounded coordinates fit (the reference's int64 casts and clamps are value-identical).
GS_DEV bool gs_project_point_hw(const GsCamera& c, float p0, float p1, float p2, int H, int W, float u_hi, float v_hi,
                                int& h_out, int& w_out) {
  // Pointclouds.transform: rotate_ (einsum over N: FMA chain) then offset_
  const float q0 = gs_dot3_fma(p0, p1, p2, c.Ri[0], c.Ri[1], c.Ri[2]) + c.ti[0];
  const float q1 = gs_dot3_fma(p0, p1, p2, c.Ri[3], c.Ri[4], c.Ri[5]) + c.ti[1];
  const float q2 = gs_dot3_fma(p0, p1, p2, c.Ri[6], c.Ri[7], c.Ri[8]) + c.ti[2];
  const bool front = q2 > 0.0f;
  // project_points: 4x4 . (x, y, z, 1), tiny matmul: plain, ascending k
  float r[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    float acc = c.K[4 * j] * q0;
    acc = acc + c.K[4 * j + 1] * q1;
    acc = acc + c.K[4 * j + 2] * q2;
    acc = acc + c.K[4 * j + 3] * 1.0f;
    r[j] = acc;
  }
  const float zz = (r[2] != 0.0f) ? r[2] : 1.0f;
  const float u = r[0] / zz, v = r[1] / zz;
  const bool in_frame = (u > -1e-3f) && (u < u_hi) && (v > -1e-3f) && (v < v_hi) && front;
  if (!in_frame) return false;
  int wi = (int)__builtin_rintf(u), hi = (int)__builtin_rintf(v);
  wi = wi < 0 ? 0 : (wi > W - 1 ? W - 1 : wi);
  hi = hi < 0 ? 0 : (hi > H - 1 ? H - 1 : hi);
  h_out = hi;
  w_out = wi;
  return true;
}

// the same as the flat pixel index h * W + w, or -1
GS_DEV int32_t gs_project_point(const GsCamera& c, float p0, float p1, float p2, int H, int W,
                                float u_hi, float v_hi) {
  int h, w;
  return gs_project_point_hw(c, p0, p1, p2, H, W, u_hi, v_hi, h, w) ? (int32_t)(h * W + w) : -1;
}

// (h, w) on the [::ds, ::ds] lattice; ds is block-uniform: a power of two (the usual 2 / 4 / 8) is a mask
GS_DEV bool gs_on_lattice(int h, int w, int ds) {
  if ((ds & (ds - 1)) == 0) return ((h | w) & (ds - 1)) == 0;
  return ((unsigned)h % (unsigned)ds == 0u) && ((unsigned)w % (unsigned)ds == 0u);
}

// ---------------------------------------------------------------- K5b: similarity ------
// slam/fusionutils.py:396-399 for map point n against frame pixel p.
GS_DEV bool gs_is_similar(const float* __restrict__ points, const float* __restrict__ normals,
                          const float* __restrict__ gvertex, const float* __restrict__ gnormal,
                          int64_t n, int64_t p, float dist_th, float dot_th) {
  const float f0 = gvertex[3 * p], f1 = gvertex[3 * p + 1], f2 = gvertex[3 * p + 2];
  const float q0 = points[3 * n], q1 = points[3 * n + 1], q2 = points[3 * n + 2];
  const float dist = gs_norm3(f0 - q0, f1 - q1, f2 - q2);
  const float dot = gs_dot3_plain(gnormal[3 * p], gnormal[3 * p + 1], gnormal[3 * p + 2], normals[3 * n],
                                  normals[3 * n + 1], normals[3 * n + 2]);
  return (dist < dist_th) && (dot > dot_th);
}

// the same test on the pixel's global vertex f and normal g held in registers
GS_DEV bool gs_is_similar_v(const float* __restrict__ points, const float* __restrict__ normals, const float* f,
                            const float* g, int64_t n, float dist_th, float dot_th) {
  const float q0 = points[3 * n], q1 = points[3 * n + 1], q2 = points[3 * n + 2];
  const float dist = gs_norm3(f[0] - q0, f[1] - q1, f[2] - q2);
  const float dot = gs_dot3_plain(g[0], g[1], g[2], normals[3 * n], normals[3 * n + 1], normals[3 * n + 2]);
  return (dist < dist_th) && (dot > dot_th);
}

// slam/fusionutils.py:491-517: (1/(ccount+1e-20), |p - f|^2) packed so that unsigned order ==
// lexicographic float order (both are >= 0).
GS_DEV uint64_t gs_assoc_key(const float* __restrict__ points, const float* __restrict__ ccounts,
                             const float* __restrict__ gvertex, int64_t n, int64_t p) {
  const float inv = 1.0f / (ccounts[n] + 1e-20f);
  const float d0 = points[3 * n] - gvertex[3 * p];
  const float d1 = points[3 * n + 1] - gvertex[3 * p + 1];
  const float d2 = points[3 * n + 2] - gvertex[3 * p + 2];
  float ray = d0 * d0 + d1 * d1;
  ray = ray + d2 * d2;
  return ((uint64_t)__float_as_uint(inv) << 32) | (uint64_t)__float_as_uint(ray);
}
// the same key with the pixel's global vertex f in registers
GS_DEV uint64_t gs_assoc_key_v(const float* __restrict__ points, const float* __restrict__ ccounts, const float* f,
                               int64_t n) {
  const float inv = 1.0f / (ccounts[n] + 1e-20f);
  const float d0 = points[3 * n] - f[0];
  const float d1 = points[3 * n + 1] - f[1];
  const float d2 = points[3 * n + 2] - f[2];
  float ray = d0 * d0 + d1 * d1;
  ray = ray + d2 * d2;
  return ((uint64_t)__float_as_uint(inv) << 32) | (uint64_t)__float_as_uint(ray);
}
