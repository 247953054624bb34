// gs_fuse.hip — K6: confidence-weighted merge of matched surfels + ordered append of new ones
// into a capacity-backed surfel store (no per-frame reallocation / re-padding, unlike
// structures/pointclouds.py:1117-1237).  HBM-bound: 2 x 40 B per map row (parity mode rewrites
// every row exactly like the reference, slam/fusionutils.py:678-699) + 40 B read + 40 B write
// per appended pixel.
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

// slam/fusionutils.py:678-699 applied to rows [0, n_map).
__global__ void __launch_bounds__(256) gs_fuse_merge_kernel(
    float* __restrict__ points, float* __restrict__ normals, float* __restrict__ colors,
    float* __restrict__ ccounts, GsCount n_map_c, const int32_t* __restrict__ pix_of,
    const int32_t* __restrict__ any_flag, const float* __restrict__ gvertex,
    const float* __restrict__ gnormal, const float* __restrict__ rgb, const float* __restrict__ alpha,
    int renorm_all) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= gs_count(n_map_c)) return;
  if (*any_flag == 0) return;  // :659 — empty table: the reference skips the whole merge
  const int32_t p = pix_of[n];
  if (p < 0 && !renorm_all) return;
  const float a = p >= 0 ? alpha[p] : 0.0f;
  const float cc = ccounts[n];
  const float cc2 = cc + a;
  const float inv = 1.0f / (cc2 == 0.0f ? 1.0f : cc2);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float fp = p >= 0 ? gvertex[3 * (int64_t)p + k] : 0.0f;
    const float fn = p >= 0 ? gnormal[3 * (int64_t)p + k] : 0.0f;
    const float fc = p >= 0 ? rgb[3 * (int64_t)p + k] : 0.0f;
    points[3 * n + k] = ((cc * points[3 * n + k]) + (a * fp)) * inv;
    normals[3 * n + k] = ((cc * normals[3 * n + k]) + (a * fn)) * inv;
    colors[3 * n + k] = ((cc * colors[3 * n + k]) + (a * fc)) * inv;
  }
  ccounts[n] = cc2;
}

struct PredNewPixel {
  const float* depth;
  const int32_t* best_pix;  // may be NULL: every valid pixel is new
  __device__ bool operator()(int64_t p) const {
    return depth[p] > 0.0f && (best_pix == nullptr || best_pix[p] < 0);
  }
};
struct EmitAppend {
  float* points;
  float* normals;
  float* colors;
  float* ccounts;
  GsCount n_map;
  const float* gvertex;
  const float* gnormal;
  const float* rgb;
  const float* alpha;
  __device__ void operator()(int64_t p, int64_t pos) const {
    const int64_t r = gs_count(n_map) + pos;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      points[3 * r + k] = gvertex[3 * p + k];
      if (normals) normals[3 * r + k] = gnormal[3 * p + k];
      if (colors) colors[3 * r + k] = rgb[3 * p + k];
    }
    if (ccounts) ccounts[r] = alpha[p];
  }
};

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
  EmitAppend emit{points, normals, colors, ccounts, n_map_c, gvertex, gnormal, rgb, alpha};
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
                  n_map_c, gvertex, gnormal, rgb, alpha};
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
