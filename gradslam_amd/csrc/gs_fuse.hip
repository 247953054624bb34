// gs_fuse.hip — K6: confidence-weighted merge of matched surfels + ordered append of new ones
// into a capacity-backed surfel store (no per-frame reallocation / re-padding, unlike
// structures/pointclouds.py:1117-1237).  HBM-bound: 2 x 40 B per map row (parity mode rewrites
// every row exactly like the reference, slam/fusionutils.py:678-699) + 40 B read + 40 B write
// per appended pixel.
#include "gs_assoc_dev.h"
#include "gs_compact.h"

// pix_of[] = -1 and the "table is non-empty" flag = 0, in one launch
__global__ void __launch_bounds__(256) gs_fuse_init_kernel(int32_t* __restrict__ pix_of, int64_t n,
                                                           int32_t* __restrict__ any_flag) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i == 0) *any_flag = 0;
  if (i < n) pix_of[i] = -1;
}

// pixel -> matched map row scatter; also raises the "table is non-empty" flag.
__global__ void __launch_bounds__(256) gs_fuse_scatter_kernel(const int32_t* __restrict__ best_pix,
                                                              int64_t P, GsCount n_map_c,
                                                              int32_t* __restrict__ pix_of,
                                                              int32_t* __restrict__ any_flag) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const int32_t n = best_pix[p];
  if (n >= 0 && n < gs_count(n_map_c)) {
    pix_of[n] = (int32_t)p;
    *any_flag = 1;  // benign race: every writer stores the same value
  }
}

// slam/fusionutils.py:678-699 for one map row n (p = its winning pixel or -1).
// All loads first, then the arithmetic, then the stores: the three floats of an attribute then travel as ONE 12-byte
// access (interleaved loads and stores compile to 21 + 10 single-dword accesses, and the scattered frame gathers of
// the matched rows make the kernel address-rate bound).
// `frame(p, fp, fn)` yields the global vertex and normal of pixel p: from the materialised global maps (FrameGlobalMaps)
// or computed on the spot from the local maps and the pose (FrameLocalMaps: the same operations, so the same bits).
struct FrameGlobalMaps {
  const float* __restrict__ gvertex;
  const float* __restrict__ gnormal;
  __device__ void operator()(const int64_t p, float* fp, float* fn) const {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      fp[k] = gvertex[3 * p + k];
      fn[k] = gnormal[3 * p + k];
    }
  }
  __device__ void vertex_of(const int64_t p, float* fp) const {
#pragma unroll
    for (int k = 0; k < 3; ++k) fp[k] = gvertex[3 * p + k];
  }
  __device__ void normal_of(const int64_t p, float* fn) const {
#pragma unroll
    for (int k = 0; k < 3; ++k) fn[k] = gnormal[3 * p + k];
  }
};
// rgbdimages.py:700-708 / :760-762 for ONE pixel, as gs_global_maps_kernel / gs_mu_pixel_init_kernel compute it.
// The validity mask is read off the local vertex: for a vertex map made by the frame-map kernel (gs_frame.hip, fm_vertex:
// z = (1.0f * d) * (d > 0 ? 1 : 0)) `z > 0` holds exactly when `d > 0` does -- d <= 0 gives +-0 or NaN, NaN gives NaN.
// (A gather of depth[p] instead was measured: these passes are bound by the number of scattered accesses, and that
// fifth one cost 15 us in the projection pass and 5 us in the merge at 8 x 640x480.)  Only the one-call step, which makes
// the vertex map itself, uses this form.
struct FrameLocalMaps {
  const float* __restrict__ vertex;
  const float* __restrict__ normal;
  float T[12];
  __device__ void operator()(const int64_t p, float* fp, float* fn) const {
    float v[3], n[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      v[k] = vertex[3 * p + k];
      n[k] = normal[3 * p + k];
    }
    const float validf = v[2] > 0.0f ? 1.0f : 0.0f;
    float g0, g1, g2;
    gs_rigid_fma(T, v[0], v[1], v[2], g0, g1, g2);
    fp[0] = g0 * validf; fp[1] = g1 * validf; fp[2] = g2 * validf;
    fn[0] = gs_dot3_fma(T[0], T[1], T[2], n[0], n[1], n[2]);
    fn[1] = gs_dot3_fma(T[4], T[5], T[6], n[0], n[1], n[2]);
    fn[2] = gs_dot3_fma(T[8], T[9], T[10], n[0], n[1], n[2]);
  }
  __device__ void vertex_of(const int64_t p, float* fp) const {
    float v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] = vertex[3 * p + k];
    const float validf = v[2] > 0.0f ? 1.0f : 0.0f;
    float g0, g1, g2;
    gs_rigid_fma(T, v[0], v[1], v[2], g0, g1, g2);
    fp[0] = g0 * validf; fp[1] = g1 * validf; fp[2] = g2 * validf;
  }
  __device__ void normal_of(const int64_t p, float* fn) const {
    float n[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) n[k] = normal[3 * p + k];
    fn[0] = gs_dot3_fma(T[0], T[1], T[2], n[0], n[1], n[2]);
    fn[1] = gs_dot3_fma(T[4], T[5], T[6], n[0], n[1], n[2]);
    fn[2] = gs_dot3_fma(T[8], T[9], T[10], n[0], n[1], n[2]);
  }
};
GS_DEV FrameLocalMaps frame_local_maps(const float* vertex, const float* normal, const float* pose16) {
  FrameLocalMaps f;
  f.vertex = vertex; f.normal = normal;
#pragma unroll
  for (int i = 0; i < 12; ++i) f.T[i] = pose16[i];
  return f;
}

template <class Frame>
GS_DEV void fuse_merge_row(float* __restrict__ points, float* __restrict__ normals, float* __restrict__ colors,
                           float* __restrict__ ccounts, const int64_t n, const int32_t p, const Frame& frame,
                           const float* __restrict__ rgb, const float* __restrict__ alpha) {
  const float a = p >= 0 ? alpha[p] : 0.0f;
  const float cc = ccounts[n];
  float P[3], N[3], C[3], fp[3] = {0.0f, 0.0f, 0.0f}, fn[3] = {0.0f, 0.0f, 0.0f}, fc[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    P[k] = points[3 * n + k];
    N[k] = normals[3 * n + k];
    C[k] = colors[3 * n + k];
  }
  if (p >= 0) {
    frame((int64_t)p, fp, fn);
#pragma unroll
    for (int k = 0; k < 3; ++k) fc[k] = rgb[3 * (int64_t)p + k];
  }
  const float cc2 = cc + a;
  const float inv = 1.0f / (cc2 == 0.0f ? 1.0f : cc2);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    P[k] = ((cc * P[k]) + (a * fp[k])) * inv;
    N[k] = ((cc * N[k]) + (a * fn[k])) * inv;
    C[k] = ((cc * C[k]) + (a * fc[k])) * inv;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) points[3 * n + k] = P[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) normals[3 * n + k] = N[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) colors[3 * n + k] = C[k];
  ccounts[n] = cc2;
}

// slam/fusionutils.py:678-699 applied to rows [0, n_map).
__global__ void __launch_bounds__(256) gs_fuse_merge_kernel(
    float* __restrict__ points, float* __restrict__ normals, float* __restrict__ colors,
    float* __restrict__ ccounts, GsCount n_map_c, const int32_t* __restrict__ pix_of,
    const int32_t* __restrict__ any_flag, const float* __restrict__ gvertex,
    const float* __restrict__ gnormal, const float* __restrict__ rgb, const float* __restrict__ alpha,
    int renorm_all) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= gs_count(n_map_c)) return;
  // :659 — empty table: the reference skips the whole merge.  Its test is on the table of the WHOLE batch, so a
  // caller that handles the sequences of a batch one call at a time passes renorm_all = 2 ("another sequence has
  // matches": rewrite every row even though this table is empty)
  if (*any_flag == 0 && renorm_all != 2) return;
  const int32_t p = pix_of[n];
  if (p < 0 && !renorm_all) return;
  fuse_merge_row(points, normals, colors, ccounts, n, p, FrameGlobalMaps{gvertex, gnormal}, rgb, alpha);
}

struct PredNewPixel {
  const float* depth;
  const int32_t* best_pix;  // may be NULL: every valid pixel is new
  __device__ bool operator()(int64_t p) const {
    return depth[p] > 0.0f && (best_pix == nullptr || best_pix[p] < 0);
  }
};
template <class Frame>
struct EmitAppendT {
  float* points;
  float* normals;
  float* colors;
  float* ccounts;
  GsCount n_map;
  Frame frame;   // global vertex / normal of a pixel (FrameGlobalMaps | FrameLocalMaps)
  const float* rgb;
  const float* alpha;
  __device__ void operator()(int64_t p, int64_t pos) const {
    const int64_t r = gs_count(n_map) + pos;
    float v[3], nn[3] = {0.0f, 0.0f, 0.0f}, c[3] = {0.0f, 0.0f, 0.0f};  // loads, then stores: 12-byte accesses
    frame.vertex_of(p, v);
    if (normals) frame.normal_of(p, nn);
    if (colors) {
#pragma unroll
      for (int k = 0; k < 3; ++k) c[k] = rgb[3 * p + k];
    }
    const float a = ccounts ? alpha[p] : 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) points[3 * r + k] = v[k];
    if (normals) {
#pragma unroll
      for (int k = 0; k < 3; ++k) normals[3 * r + k] = nn[k];
    }
    if (colors) {
#pragma unroll
      for (int k = 0; k < 3; ++k) colors[3 * r + k] = c[k];
    }
    if (ccounts) ccounts[r] = a;
  }
};
typedef EmitAppendT<FrameGlobalMaps> EmitAppend;

static int fuse_append(float* points, float* normals, float* colors, float* ccounts, GsCount n_map_c,
                       int64_t capacity, const int32_t* best_pix, const float* gvertex, const float* gnormal,
                       const float* rgb, const float* alpha, const float* depth, int H, int W, int renorm_all,
                       int64_t* new_count_out, void* scratch, void* stream) {
  const int64_t n_map_host = n_map_c.host;
  GS_REQUIRE(H > 0 && W > 0 && n_map_host >= 0 && capacity >= n_map_host, "bad sizes");
  GS_REQUIRE(points && normals && colors && ccounts && best_pix && gvertex && gnormal && rgb && alpha &&
                 depth && new_count_out && scratch,
             "NULL pointer");
  hipStream_t st = gs_stream(stream);
  const int64_t P = (int64_t)H * W;
  const int64_t n_map = n_map_host;
  // scratch: [compaction | any_flag (256 B) | pix_of int32[n_map]]
  char* base = reinterpret_cast<char*>(scratch) + gs_cp_scratch_bytes(P > n_map ? P : n_map);
  int32_t* any_flag = reinterpret_cast<int32_t*>(base);
  int32_t* pix_of = reinterpret_cast<int32_t*>(base + 256);
  // parity mode rewrites every row (2 x 40 B) + 8 B pix_of traffic; appended pixels 40 B + 40 B
  GsProf prof(GS_PROF_FUSE, 88.0 * (double)n_map + 45.0 * (double)P, st);
  if (n_map > 0) {
    hipLaunchKernelGGL(gs_fuse_init_kernel, dim3((unsigned)gs_ceil_div(n_map, 256)), dim3(256), 0, st,
                       pix_of, n_map, any_flag);
    hipLaunchKernelGGL(gs_fuse_scatter_kernel, dim3((unsigned)gs_ceil_div(P, 256)), dim3(256), 0, st,
                       best_pix, P, n_map_c, pix_of, any_flag);
    hipLaunchKernelGGL(gs_fuse_merge_kernel, dim3((unsigned)gs_ceil_div(n_map, 256)), dim3(256), 0, st,
                       points, normals, colors, ccounts, n_map_c, pix_of, any_flag, gvertex, gnormal, rgb,
                       alpha, renorm_all);
    GS_LAUNCH_CHECK();
  }
  EmitAppend emit{points, normals, colors, ccounts, n_map_c, FrameGlobalMaps{gvertex, gnormal}, rgb, alpha};
  return gs_compact(GsCount{P, nullptr}, PredNewPixel{depth, best_pix}, emit, new_count_out, n_map_c, capacity,
                    scratch, st);
}

extern "C" int gs_fuse_append_f32(float* points, float* normals, float* colors, float* ccounts,
                                  int64_t n_map_host, int64_t capacity, const int32_t* best_pix,
                                  const float* gvertex, const float* gnormal, const float* rgb,
                                  const float* alpha, const float* depth, int H, int W, int renorm_all,
                                  int64_t* new_count_out, void* scratch, void* stream) {
  return fuse_append(points, normals, colors, ccounts, GsCount{n_map_host, nullptr}, capacity, best_pix, gvertex,
                     gnormal, rgb, alpha, depth, H, W, renorm_all, new_count_out, scratch, stream);
}
extern "C" int gs_fuse_append_dc_f32(float* points, float* normals, float* colors, float* ccounts,
                                     int64_t n_map_bound, const int64_t* n_map_dev, int64_t capacity,
                                     const int32_t* best_pix, const float* gvertex, const float* gnormal,
                                     const float* rgb, const float* alpha, const float* depth, int H, int W,
                                     int renorm_all, int64_t* new_count_out, void* scratch, void* stream) {
  GS_REQUIRE(n_map_dev && capacity >= n_map_bound + (int64_t)H * W, "device-count fuse needs capacity >= bound + H*W");
  return fuse_append(points, normals, colors, ccounts, GsCount{n_map_bound, n_map_dev}, capacity, best_pix,
                     gvertex, gnormal, rgb, alpha, depth, H, W, renorm_all, new_count_out, scratch, stream);
}

static int append_valid(float* points, float* normals, float* colors, float* ccounts, GsCount n_map_c,
                        int64_t capacity, const float* gvertex, const float* gnormal, const float* rgb,
                        const float* alpha, const float* depth, int H, int W, int64_t* new_count_out,
                        void* scratch, void* stream) {
  GS_REQUIRE(H > 0 && W > 0 && n_map_c.host >= 0 && capacity >= n_map_c.host, "bad sizes");
  GS_REQUIRE(points && gvertex && depth && new_count_out && scratch, "NULL pointer");
  EmitAppend emit{points, gnormal ? normals : nullptr, rgb ? colors : nullptr, alpha ? ccounts : nullptr,
                  n_map_c, FrameGlobalMaps{gvertex, gnormal}, rgb, alpha};
  return gs_compact(GsCount{(int64_t)H * W, nullptr}, PredNewPixel{depth, nullptr}, emit, new_count_out, n_map_c,
                    capacity, scratch, gs_stream(stream));
}
extern "C" int gs_append_valid_f32(float* points, float* normals, float* colors, float* ccounts,
                                   int64_t n_map_host, int64_t capacity, const float* gvertex,
                                   const float* gnormal, const float* rgb, const float* alpha,
                                   const float* depth, int H, int W, int64_t* new_count_out,
                                   void* scratch, void* stream) {
  return append_valid(points, normals, colors, ccounts, GsCount{n_map_host, nullptr}, capacity, gvertex, gnormal,
                      rgb, alpha, depth, H, W, new_count_out, scratch, stream);
}
extern "C" int gs_append_valid_dc_f32(float* points, float* normals, float* colors, float* ccounts,
                                      int64_t n_map_bound, const int64_t* n_map_dev, int64_t capacity,
                                      const float* gvertex, const float* gnormal, const float* rgb,
                                      const float* alpha, const float* depth, int H, int W,
                                      int64_t* new_count_out, void* scratch, void* stream) {
  GS_REQUIRE(n_map_dev && capacity >= n_map_bound + (int64_t)H * W, "device-count append needs capacity >= bound + H*W");
  return append_valid(points, normals, colors, ccounts, GsCount{n_map_bound, n_map_dev}, capacity, gvertex,
                      gnormal, rgb, alpha, depth, H, W, new_count_out, scratch, stream);
}

// ---------------------------------------------------------------- K7: backward of the fuse ----
// Reverse mode of fuse_with_map (slam/fusionutils.py:653-720) for one sequence: given the adjoints of the
// fused map (points / normals / colours / confidence counts, n_new rows), the adjoints of the OLD map rows and
// of the frame's global vertex / normal maps, colours and alpha.  Correspondences are constants (index ops).
// Per old row n with matched pixel p (alpha = 0, frame value = 0 when unmatched):
//   u = cc x + alpha f,  cc' = cc + alpha,  x' = u / where(cc' == 0, 1, cc')           (three attributes)
// Every pixel is written by at most one row (its winner) or by one appended row: plain stores, no atomics.
__global__ void __launch_bounds__(256) gs_fuse_bwd_rows_kernel(
    const float* __restrict__ points, const float* __restrict__ normals, const float* __restrict__ colors,
    const float* __restrict__ ccounts, int64_t n_old, const int32_t* __restrict__ pix_of,
    const int32_t* __restrict__ any_flag, const float* __restrict__ gvertex, const float* __restrict__ gnormal,
    const float* __restrict__ rgb, const float* __restrict__ alpha, int renorm_all,
    const float* __restrict__ P_bar, const float* __restrict__ N_bar, const float* __restrict__ C_bar,
    const float* __restrict__ F_bar, float* __restrict__ oP_bar, float* __restrict__ oN_bar,
    float* __restrict__ oC_bar, float* __restrict__ oF_bar, float* __restrict__ gv_bar, float* __restrict__ gn_bar,
    float* __restrict__ rgb_bar, float* __restrict__ alpha_bar) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= n_old) return;
  // (renorm_all == 2: the forward pass rewrote every row even when this sequence had no match -- another sequence of the
  // batch had one)
  const int32_t p = (*any_flag != 0) ? pix_of[n] : -1;
  if ((*any_flag == 0 && renorm_all != 2) || (p < 0 && !renorm_all)) {  // the row was not rewritten: identity
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      oP_bar[3 * n + k] = P_bar[3 * n + k];
      oN_bar[3 * n + k] = N_bar[3 * n + k];
      oC_bar[3 * n + k] = C_bar[3 * n + k];
    }
    oF_bar[n] = F_bar[n];
    return;
  }
  const double a = p >= 0 ? (double)alpha[p] : 0.0;
  const double cc = (double)ccounts[n];
  const double cc2 = cc + a;
  const double inv = 1.0 / (cc2 == 0.0 ? 1.0 : cc2);
  double inv_bar = 0.0, cc_bar = 0.0, a_bar = 0.0;
  const float* olds[3] = {points, normals, colors};
  const float* frames[3] = {gvertex, gnormal, rgb};
  const float* bars[3] = {P_bar, N_bar, C_bar};
  float* obars[3] = {oP_bar, oN_bar, oC_bar};
  float* fbars[3] = {gv_bar, gn_bar, rgb_bar};
#pragma unroll
  for (int t = 0; t < 3; ++t) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double x = (double)olds[t][3 * n + k];
      const double f = p >= 0 ? (double)frames[t][3 * (int64_t)p + k] : 0.0;
      const double xb = (double)bars[t][3 * n + k];
      const double ub = xb * inv;
      inv_bar += xb * (cc * x + a * f);
      obars[t][3 * n + k] = (float)(cc * ub);
      if (p >= 0) fbars[t][3 * (int64_t)p + k] = (float)(a * ub);
      cc_bar += x * ub;
      a_bar += f * ub;
    }
  }
  const double cc2_bar = (double)F_bar[n] + (cc2 != 0.0 ? -inv * inv * inv_bar : 0.0);
  oF_bar[n] = (float)(cc_bar + cc2_bar);
  if (p >= 0) alpha_bar[p] = (float)(a_bar + cc2_bar);
}

struct EmitAppendBackward {
  const float* P_bar;
  const float* N_bar;
  const float* C_bar;
  const float* F_bar;
  int64_t n_old;
  float* gv_bar;
  float* gn_bar;
  float* rgb_bar;
  float* alpha_bar;
  __device__ void operator()(int64_t p, int64_t pos) const {
    const int64_t r = n_old + pos;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      gv_bar[3 * p + k] = P_bar[3 * r + k];
      gn_bar[3 * p + k] = N_bar[3 * r + k];
      rgb_bar[3 * p + k] = C_bar[3 * r + k];
    }
    alpha_bar[p] = F_bar[r];
  }
};

extern "C" int gs_fuse_append_backward_f32(const float* points, const float* normals, const float* colors,
                                           const float* ccounts, int64_t n_old, const int32_t* best_pix,
                                           const float* gvertex, const float* gnormal, const float* rgb,
                                           const float* alpha, const float* depth, int H, int W, int renorm_all,
                                           const float* P_bar, const float* N_bar, const float* C_bar,
                                           const float* F_bar, int64_t n_new, float* old_points_bar,
                                           float* old_normals_bar, float* old_colors_bar, float* old_ccounts_bar,
                                           float* gvertex_bar, float* gnormal_bar, float* rgb_bar, float* alpha_bar,
                                           void* scratch, void* stream) {
  GS_REQUIRE(H > 0 && W > 0 && n_old >= 0 && n_new >= n_old, "bad sizes");
  GS_REQUIRE(best_pix && gvertex && gnormal && rgb && alpha && depth && P_bar && N_bar && C_bar && F_bar && gvertex_bar &&
                 gnormal_bar && rgb_bar && alpha_bar && scratch,
             "NULL pointer");
  hipStream_t st = gs_stream(stream);
  const int64_t P = (int64_t)H * W;
  GS_HIP(hipMemsetAsync(gvertex_bar, 0, 12 * (size_t)P, st));
  GS_HIP(hipMemsetAsync(gnormal_bar, 0, 12 * (size_t)P, st));
  GS_HIP(hipMemsetAsync(rgb_bar, 0, 12 * (size_t)P, st));
  GS_HIP(hipMemsetAsync(alpha_bar, 0, 4 * (size_t)P, st));
  char* base = reinterpret_cast<char*>(scratch) + gs_cp_scratch_bytes(P > n_old ? P : n_old);
  int32_t* any_flag = reinterpret_cast<int32_t*>(base);
  int32_t* pix_of = reinterpret_cast<int32_t*>(base + 256);
  if (n_old > 0) {
    GS_REQUIRE(points && normals && colors && ccounts && old_points_bar && old_normals_bar && old_colors_bar &&
                   old_ccounts_bar,
               "NULL pointer");
    const GsCount n_old_c{n_old, nullptr};
    hipLaunchKernelGGL(gs_fuse_init_kernel, dim3((unsigned)gs_ceil_div(n_old, 256)), dim3(256), 0, st, pix_of, n_old,
                       any_flag);
    hipLaunchKernelGGL(gs_fuse_scatter_kernel, dim3((unsigned)gs_ceil_div(P, 256)), dim3(256), 0, st, best_pix, P,
                       n_old_c, pix_of, any_flag);
    hipLaunchKernelGGL(gs_fuse_bwd_rows_kernel, dim3((unsigned)gs_ceil_div(n_old, 256)), dim3(256), 0, st, points,
                       normals, colors, ccounts, n_old, pix_of, any_flag, gvertex, gnormal, rgb, alpha, renorm_all, P_bar,
                       N_bar, C_bar, F_bar, old_points_bar, old_normals_bar, old_colors_bar, old_ccounts_bar,
                       gvertex_bar, gnormal_bar, rgb_bar, alpha_bar);
    GS_LAUNCH_CHECK();
  }
  // appended rows: row n_old + k is the k-th new pixel in raster order (the forward's ordered compaction)
  EmitAppendBackward emit{P_bar, N_bar, C_bar, F_bar, n_old, gvertex_bar, gnormal_bar, rgb_bar, alpha_bar};
  int64_t* cnt = reinterpret_cast<int64_t*>(base + 128);
  return gs_compact(P, PredNewPixel{depth, best_pix}, emit, cnt, 0, n_new - n_old, scratch, st);
}

// ---------------------------------------------------------------- one-call map update ----
// update_map_fusion (slam/fusionutils.py:761-789) with the kernels regrouped by DOMAIN so that a frame costs 6
// launches instead of 11 (global maps 1 + projection 1 + association 3 + fuse 6), for B independent sequences per
// launch (block b -> sequence b % B on its block b / B); every value is computed by the same arithmetic as in the
// separate entry points:
//   U1 per pixel : global vertex / normal under the new pose; clear the per-pixel key and winner tables
//   U2 per surfel: project, similarity test, per-pixel atomicMin of the (1/ccount, ray) key
//   U3 (round 5: no longer a pass over the map) the winner of a pixel is the row that attains its key; two rows with the SAME
//      key (same confidence bits, same distance bits: the lower index wins, slam/fusionutils.py:491-536) are noticed by U2
//      -- the atomicMin of the second one returns its own key -- which marks the pixel; the per-pixel pass U4 then settles the
//      marked pixels by an atomicMin of the row index (skipped at once when no pixel of the frame is marked)
//   U4 per pixel : "any match" flag, count of new pixels per tile (a pixel is matched iff its key was written)
//   U5 per surfel: winner test (key attained, and no lower-indexed row holds the pixel) + confidence-weighted merge
//                  (parity mode rewrites every row); the winner writes best_pix                } one launch
//   U6 per pixel : ordered append of the new pixels; every block derives its output offset from the tile
//                  counts itself (no separate scan launch); block 0 also writes the new surfel count
constexpr int32_t MU_TIE_MARK = 0x7fffffff;   // best_pix of a pixel with two rows of the same key, until settled
struct MuSeq {
  float* points; float* normals; float* colors; float* ccounts;
  GsCount n_map;
  int64_t capacity;
  const float* vertex; const float* normal; const float* depth; const float* rgb; const float* alpha;
  const float* pose16; const float* K16;
  float* gvertex; float* gnormal;
  int32_t* best_pix;
  int64_t* new_count_out;
  // scratch
  int32_t* any_flag;
  uint64_t* key_pix;
  int32_t* tile_counts;
  uint64_t* key_pt;
  int32_t* pix;
  int32_t* pix_of;
};
struct MuBatch {
  int B, H, W, renorm_all;
  int64_t P, ntiles;
  float u_hi, v_hi, dist_th, dot_th;
  // "some sequence of the CALL has a match": batches beyond GS_MAX_BATCH sequences run in chunks, and the reference's
  // test (fusionutils.py:659) looks at the table of the whole batch.  One word shared by all chunks of a call (in the
  // first sequence's scratch), zeroed by the first chunk's pixel pass, set by every chunk's winner pass, read by the
  // merge passes, which are enqueued after the winner passes of ALL chunks.
  int32_t* call_flag;
  int zero_call_flag;
  MuSeq s[GS_MAX_BATCH];
};

__global__ void __launch_bounds__(256) gs_mu_pixel_init_kernel(const MuBatch mb) {
  const MuSeq& q = mb.s[blockIdx.x % mb.B];
  const int64_t p = (int64_t)(blockIdx.x / mb.B) * 256 + threadIdx.x;
  if (p == 0) { q.any_flag[0] = 0; q.any_flag[1] = 0; }   // ([1]: "some pixel has two rows with the same key")
  if (blockIdx.x == 0 && threadIdx.x == 0 && mb.zero_call_flag) *mb.call_flag = 0;
  if (p >= mb.P) return;
  float T[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) T[i] = q.pose16[i];
  const float validf = q.depth[p] > 0.0f ? 1.0f : 0.0f;
  float g0, g1, g2;
  gs_rigid_fma(T, q.vertex[3 * p], q.vertex[3 * p + 1], q.vertex[3 * p + 2], g0, g1, g2);
  q.gvertex[3 * p] = g0 * validf;
  q.gvertex[3 * p + 1] = g1 * validf;
  q.gvertex[3 * p + 2] = g2 * validf;
  const float n0 = q.normal[3 * p], n1 = q.normal[3 * p + 1], n2 = q.normal[3 * p + 2];
  q.gnormal[3 * p] = gs_dot3_fma(T[0], T[1], T[2], n0, n1, n2);
  q.gnormal[3 * p + 1] = gs_dot3_fma(T[4], T[5], T[6], n0, n1, n2);
  q.gnormal[3 * p + 2] = gs_dot3_fma(T[8], T[9], T[10], n0, n1, n2);
  q.key_pix[p] = ~0ull;
  q.best_pix[p] = -1;
}

__global__ void __launch_bounds__(256) gs_mu_project_key_kernel(const MuBatch mb) {
  const MuSeq& q = mb.s[blockIdx.x % mb.B];
  const int64_t n = (int64_t)(blockIdx.x / mb.B) * 256 + threadIdx.x;
  if (n >= gs_count(q.n_map)) return;
  const GsCamera c = gs_camera(q.pose16, q.K16);
  const int32_t p = gs_project_point(c, q.points[3 * n], q.points[3 * n + 1], q.points[3 * n + 2], mb.H, mb.W, mb.u_hi,
                                     mb.v_hi);
  // pix[n] = the pixel this row competes for, or -1 (not in the frame, or not similar to the pixel): the pick pass
  // then reads 4 bytes of most rows instead of 12, and the key is only stored for the rows that have one
  int32_t pk = -1;
  if (p >= 0) {
    // the pixel's global vertex / normal: materialised by the pixel pass, or computed here (MuSeq::gvertex == NULL: the
    // one-call step, whose frame-map launch initialised the tables -- there is no pixel pass then)
    float fp[3], fn[3];
    if (q.gvertex) FrameGlobalMaps{q.gvertex, q.gnormal}((int64_t)p, fp, fn);
    else frame_local_maps(q.vertex, q.normal, q.pose16)((int64_t)p, fp, fn);
    if (gs_is_similar_v(q.points, q.normals, fp, fn, n, mb.dist_th, mb.dot_th)) {
      const uint64_t k = gs_assoc_key_v(q.points, q.ccounts, fp, n);
      const unsigned long long old = atomicMin(reinterpret_cast<unsigned long long*>(&q.key_pix[p]), (unsigned long long)k);
      // Two rows with the same key at one pixel: whichever arrives second gets its own key back.  (If the key is beaten
      // later the mark is spurious and harmless; if it stays the minimum, every tie of the winning key has been seen:
      // the first of the tied rows to arrive set it, every later one reads it back.)  The pixel is marked -- any value
      // that is neither "no winner" (-1) nor a row index below it -- and settled by mu_settle_ties (in the per-pixel pass).
      if (old == (unsigned long long)k) {
        q.best_pix[p] = MU_TIE_MARK;
        q.any_flag[1] = 1;   // benign race: every writer stores the same value
      }
      q.key_pt[n] = k;
      pk = p;
    }
  }
  q.pix[n] = pk;
}

// Settles the pixels U2 marked: the lowest-indexed row among those that attain the pixel's key.  Run by the blocks of the
// per-pixel pass below (which comes between the key pass and the merge anyway) and skipped at once unless the frame has a
// marked pixel (two surfels with bit-identical confidence and distance).
GS_DEV void mu_settle_ties(const MuSeq& q, const unsigned blk, const unsigned nblk) {
  if (q.any_flag[1] == 0) return;
  const int64_t n_map = gs_count(q.n_map);
  for (int64_t n = (int64_t)blk * GS_CP_BLOCK + threadIdx.x; n < n_map; n += (int64_t)nblk * GS_CP_BLOCK) {
    const int32_t p = q.pix[n];
    if (p < 0 || q.best_pix[p] == -1) continue;   // (-1: unmarked pixel -- its key has one holder)
    if (q.key_pt[n] == q.key_pix[p]) atomicMin(reinterpret_cast<unsigned*>(&q.best_pix[p]), (unsigned)n);
  }
}

// per pixel tile of GS_CP_TILE pixels: inverse map of the winners + number of new pixels of the tile
__global__ void __launch_bounds__(GS_CP_BLOCK) gs_mu_winner_count_kernel(const MuBatch mb) {
  __shared__ int smem[GS_CP_BLOCK / GS_WAVE + 1];
  const MuSeq& q = mb.s[blockIdx.x % mb.B];
  const unsigned blk = blockIdx.x / mb.B;
  const int64_t base = (int64_t)blk * GS_CP_TILE + (int64_t)threadIdx.x * GS_CP_ITEMS;
  int c = 0;
  // all loads of the tile first (clamped index), then the work: inside the per-pixel control flow they are serialised
  uint64_t kk[GS_CP_ITEMS];
  float dd[GS_CP_ITEMS];
#pragma unroll
  for (int i = 0; i < GS_CP_ITEMS; ++i) {
    const int64_t pc = base + i < mb.P ? base + i : mb.P - 1;
    kk[i] = q.key_pix[pc];
    dd[i] = q.depth[pc];
  }
  bool any = false;
#pragma unroll
  for (int i = 0; i < GS_CP_ITEMS; ++i) {
    const int64_t p = base + i;
    if (p < mb.P) {
      const bool matched = kk[i] != ~0ull;   // some row wrote its key: the pixel has a winner
      any = any || matched;
      if (dd[i] > 0.0f && !matched) ++c;
    }
  }
  if (any) {
    q.any_flag[0] = 1;  // benign race: every writer stores the same value
    *mb.call_flag = 1;
  }
  int total;
  (void)gs_block_excl_scan<GS_CP_BLOCK>(c, smem, &total);
  if (threadIdx.x == 0) q.tile_counts[blk] = total;
  mu_settle_ties(q, blk, gridDim.x / mb.B);
}

GS_DEV void mu_merge_body(const MuBatch& mb, const unsigned bid) {
  const MuSeq& q = mb.s[bid % mb.B];
  const int64_t n = (int64_t)(bid / mb.B) * 256 + threadIdx.x;
  if (n >= gs_count(q.n_map)) return;
  // :659 -- the reference skips the whole merge only when the correspondence table of the WHOLE batch is empty
  // (pc2im_bnhw.shape[0] != 0 is a batch-level test): a sequence without matches is still renormalised when
  // another sequence of the same call has some
  if (*mb.call_flag == 0) return;
  // the pixel this row competes for; it wins iff it attains the pixel's key and no lower-indexed row with the same key
  // holds the pixel (best_pix: -1 = the key has one holder, else the settled holder)
  int32_t p = q.pix[n];
  if (p >= 0) {
    bool win = q.key_pt[n] == q.key_pix[p];
    if (win) {   // (only the holders of the key look at the entry: an unmarked pixel's has one reader and writer)
      const int32_t bp = q.best_pix[p];
      win = bp == -1 || bp == (int32_t)n;
      if (win && bp == -1) q.best_pix[p] = (int32_t)n;
    }
    if (!win) p = -1;
  }
  if (p < 0 && !mb.renorm_all) return;
  if (q.gvertex) fuse_merge_row(q.points, q.normals, q.colors, q.ccounts, n, p, FrameGlobalMaps{q.gvertex, q.gnormal}, q.rgb, q.alpha);
  else fuse_merge_row(q.points, q.normals, q.colors, q.ccounts, n, p, frame_local_maps(q.vertex, q.normal, q.pose16), q.rgb, q.alpha);
}

// ordered append without a scan launch: block b adds up the counts of the tiles before it (fixed order)
GS_DEV void mu_append_body(const MuBatch& mb, const unsigned bid) {
  __shared__ int smem[GS_CP_BLOCK / GS_WAVE + 1];
  const MuSeq& q = mb.s[bid % mb.B];
  const unsigned blk = bid / mb.B;
  int before = 0, all = 0;
  for (int64_t t = threadIdx.x; t < mb.ntiles; t += GS_CP_BLOCK) {
    const int v = q.tile_counts[t];
    all += v;
    if (t < (int64_t)blk) before += v;
  }
  int tile_prefix, total_new, tile_new;
  (void)gs_block_excl_scan<GS_CP_BLOCK>(before, smem, &tile_prefix);
  (void)gs_block_excl_scan<GS_CP_BLOCK>(all, smem, &total_new);
  const int64_t n_map = gs_count(q.n_map);
  if (blk == 0 && threadIdx.x == 0) q.new_count_out[0] = n_map + total_new;
  const int64_t base = (int64_t)blk * GS_CP_TILE + (int64_t)threadIdx.x * GS_CP_ITEMS;
  bool keep[GS_CP_ITEMS];
  int c = 0;
  {
    uint64_t kk[GS_CP_ITEMS];  // loads first (see gs_mu_winner_count_kernel)
    float dd[GS_CP_ITEMS];
#pragma unroll
    for (int i = 0; i < GS_CP_ITEMS; ++i) {
      const int64_t pc = base + i < mb.P ? base + i : mb.P - 1;
      kk[i] = q.key_pix[pc];   // (not best_pix: the merge blocks of this launch are writing it)
      dd[i] = q.depth[pc];
    }
#pragma unroll
    for (int i = 0; i < GS_CP_ITEMS; ++i) {
      keep[i] = base + i < mb.P && dd[i] > 0.0f && kk[i] == ~0ull;
      c += keep[i] ? 1 : 0;
    }
  }
  int w = gs_block_excl_scan<GS_CP_BLOCK>(c, smem, &tile_new);
  // new pixels of the tile listed in LDS (pixel order), then row r of the tile's output range is written by thread r
  // (see gs_cp_scatter_kernel: consecutive lanes -> consecutive rows)
  __shared__ unsigned short loc_s[GS_CP_TILE];
#pragma unroll
  for (int i = 0; i < GS_CP_ITEMS; ++i) {
    if (keep[i]) loc_s[w++] = (unsigned short)(threadIdx.x * GS_CP_ITEMS + i);
  }
  __syncthreads();
  const int64_t tile_base = (int64_t)blk * GS_CP_TILE;
  if (q.gvertex) {
    const EmitAppend emit{q.points, q.normals, q.colors, q.ccounts, q.n_map, FrameGlobalMaps{q.gvertex, q.gnormal}, q.rgb, q.alpha};
    for (int r = threadIdx.x; r < tile_new; r += GS_CP_BLOCK) {
      const int64_t pos = (int64_t)tile_prefix + r;
      if (n_map + pos < q.capacity) emit(tile_base + loc_s[r], pos);
    }
  } else {
    const EmitAppendT<FrameLocalMaps> emit{q.points, q.normals, q.colors, q.ccounts, q.n_map,
                                           frame_local_maps(q.vertex, q.normal, q.pose16), q.rgb, q.alpha};
    for (int r = threadIdx.x; r < tile_new; r += GS_CP_BLOCK) {
      const int64_t pos = (int64_t)tile_prefix + r;
      if (n_map + pos < q.capacity) emit(tile_base + loc_s[r], pos);
    }
  }
}

// U5 + U6 in ONE launch: the merge touches rows [0, n_map), the append rows [n_map, ...) and neither reads what the
// other writes (both read the frame, best_pix / pix_of and the tile counts of the winner pass).  The append blocks
// (few, latency-bound: 23 us alone) come first and run under the merge blocks' streaming.
static_assert(GS_CP_BLOCK == 256, "merge and append blocks share a launch");
__global__ void __launch_bounds__(256) gs_mu_merge_append_kernel(const MuBatch mb, const unsigned nb_append) {
  if (blockIdx.x < nb_append) mu_append_body(mb, blockIdx.x);
  else mu_merge_body(mb, blockIdx.x - nb_append);
}

extern "C" int64_t gs_update_map_scratch_bytes(int64_t n_map_bound, int H, int W) {
  const int64_t P = (int64_t)H * W;
  return (int64_t)(256 + gs_align(8 * (size_t)P) + gs_align(4 * (size_t)gs_cp_tiles(P)) +
                   gs_align(8 * (size_t)(n_map_bound > 0 ? n_map_bound : 1)) +
                   2 * gs_align(4 * (size_t)(n_map_bound > 0 ? n_map_bound : 1)) + 4096);
}

// where the tables the frame-map launch of the one-call step initialises live inside an update scratch (as update_chunk
// carves it); the call flag is the second flag word of the FIRST sequence of the call (see MuBatch::call_flag)
int32_t* gs_update_map_call_flag(void* scratch0) { return reinterpret_cast<int32_t*>(scratch0) + 32; }
void gs_update_map_tables(void* scratch, int32_t** any_flag, uint64_t** key_pix) {
  *any_flag = reinterpret_cast<int32_t*>(scratch);
  *key_pix = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(scratch) + 256);
}

// phase 0: global maps, association, winner / tile counts; phase 1: merge + append (after phase 0 of every chunk)
static int update_chunk(const gs_update_seq* seqs, int B, int H, int W, float dist_th, float dot_th, int renorm_all,
                        hipStream_t st, int phase, int32_t* call_flag, bool first_chunk, bool tables_ready) {
  MuBatch mb;
  mb.B = B; mb.H = H; mb.W = W; mb.renorm_all = renorm_all;
  mb.call_flag = call_flag; mb.zero_call_flag = first_chunk ? 1 : 0;
  mb.P = (int64_t)H * W;
  mb.ntiles = gs_cp_tiles(mb.P);
  mb.u_hi = (float)((double)W - 0.999); mb.v_hi = (float)((double)H - 0.999);
  mb.dist_th = dist_th; mb.dot_th = dot_th;
  int64_t n_max = 0;
  double bytes_assoc = 0.0, bytes_fuse = 0.0;
  for (int b = 0; b < B; ++b) {
    const gs_update_seq& u = seqs[b];
    const int64_t n_map = u.map.n_bound;
    n_max = n_map > n_max ? n_map : n_max;
    char* q = reinterpret_cast<char*>(u.scratch);
    MuSeq& m = mb.s[b];
    m.points = u.map.points; m.normals = u.map.normals; m.colors = u.map.colors; m.ccounts = u.map.ccounts;
    m.n_map = GsCount{u.map.n_bound, u.map.n_dev};
    m.capacity = u.map.capacity;
    m.vertex = u.vertex; m.normal = u.normal; m.depth = u.depth; m.rgb = u.rgb; m.alpha = u.alpha;
    m.pose16 = u.pose16; m.K16 = u.K16;
    m.gvertex = u.gvertex; m.gnormal = u.gnormal; m.best_pix = u.best_pix; m.new_count_out = u.new_count_out;
    m.any_flag = reinterpret_cast<int32_t*>(q); q += 256;
    m.key_pix = reinterpret_cast<uint64_t*>(q); q += gs_align(8 * (size_t)mb.P);
    m.tile_counts = reinterpret_cast<int32_t*>(q); q += gs_align(4 * (size_t)mb.ntiles);
    m.key_pt = reinterpret_cast<uint64_t*>(q); q += gs_align(8 * (size_t)(n_map > 0 ? n_map : 1));
    m.pix = reinterpret_cast<int32_t*>(q); q += gs_align(4 * (size_t)(n_map > 0 ? n_map : 1));
    m.pix_of = reinterpret_cast<int32_t*>(q);
    // as built: projection 28 B read + 12 B written (+ 24 B frame gather and 8 B key per competing row); merge 4 + 80 B per
    // row (+ 20 B of keys / winner entry per competing row), 49 B per pixel for the winner count and the append
    bytes_assoc += 72.0 * (double)n_map;
    bytes_fuse += 104.0 * (double)n_map + 49.0 * (double)mb.P;
  }
  const unsigned uB = (unsigned)B;
  const unsigned pb = uB * (unsigned)gs_ceil_div(mb.P, 256), nb = uB * (unsigned)gs_ceil_div(n_max > 0 ? n_max : 1, 256);
  if (phase == 0) {
    if (!tables_ready) {   // (the one-call step: the frame-map launch initialised the tables, the global maps stay implicit)
      GsProf prof(GS_PROF_FRAME, (double)B * (double)mb.P * 64.0, st);   // 28 B read + 24 B + 12 B written per pixel
      hipLaunchKernelGGL(gs_mu_pixel_init_kernel, dim3(pb), dim3(256), 0, st, mb);
    }
    if (n_max > 0) {
      GsProf prof(GS_PROF_ASSOC, bytes_assoc, st, 1);
      hipLaunchKernelGGL(gs_mu_project_key_kernel, dim3(nb), dim3(256), 0, st, mb);
    }
    GsProf prof(GS_PROF_FUSE, bytes_fuse / 3.0, st, 1);
    hipLaunchKernelGGL(gs_mu_winner_count_kernel, dim3(uB * (unsigned)mb.ntiles), dim3(GS_CP_BLOCK), 0, st, mb);
  } else {
    GsProf prof(GS_PROF_FUSE, bytes_fuse * (2.0 / 3.0), st, 1);
    const unsigned nb_append = uB * (unsigned)mb.ntiles;
    hipLaunchKernelGGL(gs_mu_merge_append_kernel, dim3(nb_append + (n_max > 0 ? nb : 0u)), dim3(256), 0, st, mb, nb_append);
  }
  GS_LAUNCH_CHECK();
  return GS_OK;
}

// tables_ready: the caller's frame-map launch has initialised key_pix / best_pix / the flags (gs_update_map_tables says
// where they live); gvertex / gnormal may then be NULL -- the global vertex and normal of a pixel are computed where they
// are used (the same operations as the pixel pass: the same bits) and never written to memory.
int gs_update_map_fusion_batch_impl(const gs_update_seq* seqs_host, int B, int H, int W, float dist_th, float dot_th,
                                    int renorm_all, void* stream, bool tables_ready) {
  GS_REQUIRE(seqs_host && B > 0 && H > 0 && W > 0, "bad arguments");
  GS_REQUIRE((int64_t)H * W < (1ll << 31), "too large for int32 indices");
  for (int b = 0; b < B; ++b) {
    const gs_update_seq& u = seqs_host[b];
    GS_REQUIRE(u.map.n_bound >= 0 && u.map.n_bound < 0x7fffffff, "bad map size");
    GS_REQUIRE(u.map.capacity >= u.map.n_bound + (int64_t)H * W, "capacity must cover n_bound + H*W rows");
    GS_REQUIRE(u.map.points && u.map.normals && u.map.colors && u.map.ccounts && u.vertex && u.normal && u.depth && u.rgb &&
                   u.alpha && u.pose16 && u.K16 && u.best_pix && u.new_count_out && u.scratch,
               "NULL pointer");
    GS_REQUIRE((u.gvertex && u.gnormal) || (tables_ready && !u.gvertex && !u.gnormal), "gvertex / gnormal: both or none");
    GS_REQUIRE(u.new_count_out != u.map.n_dev, "new_count_out must not alias the map's device count");
  }
  hipStream_t st = gs_stream(stream);
  int32_t* call_flag = gs_update_map_call_flag(seqs_host[0].scratch);
  for (int phase = 0; phase < 2; ++phase) {
    for (int c0 = 0; c0 < B; c0 += GS_MAX_BATCH) {
      const int nb = B - c0 < GS_MAX_BATCH ? B - c0 : GS_MAX_BATCH;
      const int rc = update_chunk(seqs_host + c0, nb, H, W, dist_th, dot_th, renorm_all, st, phase, call_flag, c0 == 0,
                                  tables_ready);
      if (rc != GS_OK) return rc;
    }
  }
  return GS_OK;
}

extern "C" int gs_update_map_fusion_batch_f32(const gs_update_seq* seqs_host, int B, int H, int W, float dist_th,
                                              float dot_th, int renorm_all, void* stream) {
  return gs_update_map_fusion_batch_impl(seqs_host, B, H, W, dist_th, dot_th, renorm_all, stream, false);
}

extern "C" int gs_update_map_fusion_dc_f32(float* points, float* normals, float* colors, float* ccounts,
                                           int64_t n_map_bound, const int64_t* n_map_dev, int64_t capacity,
                                           const float* vertex, const float* normal, const float* depth,
                                           const float* rgb, const float* alpha, const float* pose16,
                                           const float* K16, int H, int W, float dist_th, float dot_th,
                                           int renorm_all, float* gvertex, float* gnormal, int32_t* best_pix,
                                           int64_t* new_count_out, void* scratch, void* stream) {
  gs_update_seq u;
  u.map = gs_map_view{points, normals, colors, ccounts, capacity, n_map_bound, n_map_dev};
  u.vertex = vertex; u.normal = normal; u.depth = depth; u.rgb = rgb; u.alpha = alpha;
  u.pose16 = pose16; u.K16 = K16;
  u.gvertex = gvertex; u.gnormal = gnormal; u.best_pix = best_pix; u.new_count_out = new_count_out; u.scratch = scratch;
  return gs_update_map_fusion_batch_f32(&u, 1, H, W, dist_th, dot_th, renorm_all, stream);
}
