// gs_assoc.hip — K2 (ordered down-samplers) and K5 (projective surfel association).
// All kernels stream the surfel arrays once, fully coalesced (12 B/point AoS rows are
// contiguous across lanes); frame look-ups are 12 B gathers that hit L2 because projected
// neighbours land on neighbouring pixels.  HBM-bound (28 B/point + 24 B per
// in-frame point + 8 B/pixel key traffic).
#include "gs_assoc_dev.h"
#include "gs_compact.h"

__global__ void __launch_bounds__(256) gs_project_map_kernel(
    const float* __restrict__ points, GsCount n_map_c, const float* __restrict__ pose16,
    const float* __restrict__ K16, int H, int W, float u_hi, float v_hi, int32_t* __restrict__ pix) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= gs_count(n_map_c)) return;
  const GsCamera c = gs_camera(pose16, K16);
  pix[n] = gs_project_point(c, points[3 * n], points[3 * n + 1], points[3 * n + 2], H, W, u_hi, v_hi);
}

static int project_map(const float* points, GsCount n_map_c, const float* pose16, const float* K16, int H, int W,
                       int32_t* pix, void* stream) {
  const int64_t n_map = n_map_c.host;
  GS_REQUIRE(n_map >= 0 && H > 0 && W > 0, "bad sizes");
  if (n_map == 0) return GS_OK;
  GS_REQUIRE(points && pose16 && K16 && pix, "NULL pointer");
  GS_REQUIRE((int64_t)H * W < (1ll << 31), "image too large for int32 pixel ids");
  const float u_hi = (float)((double)W - 0.999), v_hi = (float)((double)H - 0.999);
  GsProf prof(GS_PROF_PROJECT, 16.0 * (double)n_map, gs_stream(stream));  // 12 B point read + 4 B pix write
  hipLaunchKernelGGL(gs_project_map_kernel, dim3((unsigned)gs_ceil_div(n_map, 256)), dim3(256), 0,
                     gs_stream(stream), points, n_map_c, pose16, K16, H, W, u_hi, v_hi, pix);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

extern "C" int gs_project_map_f32(const float* points, int64_t n_map, const float* pose16,
                                  const float* K16, int H, int W, int32_t* pix, void* stream) {
  return project_map(points, GsCount{n_map, nullptr}, pose16, K16, H, W, pix, stream);
}
extern "C" int gs_project_map_dc_f32(const float* points, int64_t n_map_bound, const int64_t* n_map_dev,
                                     const float* pose16, const float* K16, int H, int W, int32_t* pix,
                                     void* stream) {
  GS_REQUIRE(n_map_dev, "NULL device count");
  return project_map(points, GsCount{n_map_bound, n_map_dev}, pose16, K16, H, W, pix, stream);
}

// ---------------------------------------------------------------- ordered tables -------
struct PredActive {
  const int32_t* pix;
  __device__ bool operator()(int64_t n) const { return pix[n] >= 0; }
};
struct EmitActiveRow {
  const int32_t* pix;
  int W;
  int64_t b;
  int64_t* rows;
  __device__ void operator()(int64_t n, int64_t pos) const {
    const int32_t p = pix[n];
    rows[4 * pos] = b;
    rows[4 * pos + 1] = n;
    rows[4 * pos + 2] = p / W;
    rows[4 * pos + 3] = p % W;
  }
};

extern "C" int gs_active_table_i64(const int32_t* pix, int64_t n_map, int W, int64_t b,
                                   int64_t* rows_out, int64_t* count_out, void* scratch,
                                   void* stream) {
  GS_REQUIRE(n_map >= 0 && W > 0 && count_out && scratch, "bad arguments");
  return gs_compact(n_map, PredActive{pix}, EmitActiveRow{pix, W, b, rows_out}, count_out, 0, -1, scratch,
                    gs_stream(stream));
}

struct Gather3 {
  const float* points;
  const float* normals;
  const float* colors;
  float* out_pts;
  float* out_nrm;
  float* out_rgb;
  __device__ void copy(int64_t src, int64_t dst) const {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      out_pts[3 * dst + k] = points[3 * src + k];
      if (out_nrm) out_nrm[3 * dst + k] = normals[3 * src + k];
      if (out_rgb) out_rgb[3 * dst + k] = colors[3 * src + k];
    }
  }
};

struct PredLattice {
  const int32_t* pix;
  int W, ds;
  __device__ bool operator()(int64_t n) const {
    const int32_t p = pix[n];
    if (p < 0) return false;
    return ((p / W) % ds == 0) && ((p % W) % ds == 0);
  }
};
struct EmitGatherByN {
  Gather3 g;
  __device__ void operator()(int64_t n, int64_t pos) const { g.copy(n, pos); }
};

static int select_targets(const int32_t* pix, GsCount n_map, int W, int ds, const float* points,
                          const float* normals, const float* colors, float* out_pts, float* out_nrm,
                          float* out_rgb, int64_t cap, int64_t* count_out, void* scratch, void* stream) {
  GS_REQUIRE(n_map.host >= 0 && W > 0 && ds > 0 && count_out && scratch && out_pts, "bad arguments");
  Gather3 g{points, normals, colors, out_pts, normals ? out_nrm : nullptr, colors ? out_rgb : nullptr};
  return gs_compact(n_map, PredLattice{pix, W, ds}, EmitGatherByN{g}, count_out, GsCount{0, nullptr}, cap,
                    scratch, gs_stream(stream));
}
extern "C" int gs_select_targets_f32(const int32_t* pix, int64_t n_map, int W, int ds,
                                     const float* points, const float* normals, const float* colors,
                                     float* out_pts, float* out_nrm, float* out_rgb, int64_t cap,
                                     int64_t* count_out, void* scratch, void* stream) {
  return select_targets(pix, GsCount{n_map, nullptr}, W, ds, points, normals, colors, out_pts, out_nrm, out_rgb,
                        cap, count_out, scratch, stream);
}
extern "C" int gs_select_targets_dc_f32(const int32_t* pix, int64_t n_map_bound, const int64_t* n_map_dev, int W,
                                        int ds, const float* points, const float* normals,
                                        const float* colors, float* out_pts, float* out_nrm, float* out_rgb,
                                        int64_t cap, int64_t* count_out, void* scratch, void* stream) {
  GS_REQUIRE(n_map_dev, "NULL device count");
  return select_targets(pix, GsCount{n_map_bound, n_map_dev}, W, ds, points, normals, colors, out_pts, out_nrm,
                        out_rgb, cap, count_out, scratch, stream);
}

struct PredRowLattice {
  const int64_t* rows;
  int ds;
  __device__ bool operator()(int64_t r) const {
    return (rows[4 * r + 2] % ds == 0) && (rows[4 * r + 3] % ds == 0);
  }
};
struct EmitGatherByRow {
  const int64_t* rows;
  Gather3 g;
  __device__ void operator()(int64_t r, int64_t pos) const { g.copy(rows[4 * r + 1], pos); }
};

extern "C" int gs_downsample_table_f32(const int64_t* rows, int64_t n_rows, int ds,
                                       const float* points, const float* normals,
                                       const float* colors, float* out_pts, float* out_nrm,
                                       float* out_rgb, int64_t* count_out, void* scratch,
                                       void* stream) {
  GS_REQUIRE(n_rows >= 0 && ds > 0 && count_out && scratch, "bad arguments");
  Gather3 g{points, normals, colors, out_pts, normals ? out_nrm : nullptr, colors ? out_rgb : nullptr};
  return gs_compact(n_rows, PredRowLattice{rows, ds}, EmitGatherByRow{rows, g}, count_out, 0, -1, scratch,
                    gs_stream(stream));
}

// valid pixels of the [::ds, ::ds] lattice, raster order (odometry/icputils.py:654-668)
struct PredFrameLattice {
  const float* depth;
  int W, ds, Wl;
  __device__ bool operator()(int64_t e) const {
    const int64_t h = (e / Wl) * ds, w = (e % Wl) * ds;
    return depth[h * W + w] > 0.0f;
  }
};
struct EmitFrameLattice {
  int W, ds, Wl;
  Gather3 g;
  __device__ void operator()(int64_t e, int64_t pos) const {
    const int64_t h = (e / Wl) * ds, w = (e % Wl) * ds;
    g.copy(h * W + w, pos);
  }
};

extern "C" int gs_downsample_frame_f32(const float* gvertex, const float* gnormal, const float* rgb,
                                       const float* depth, int H, int W, int ds, float* out_pts,
                                       float* out_nrm, float* out_rgb, int64_t* count_out,
                                       void* scratch, void* stream) {
  GS_REQUIRE(H > 0 && W > 0 && ds > 0 && gvertex && depth && out_pts && count_out && scratch,
             "bad arguments");
  const int Hl = (H + ds - 1) / ds, Wl = (W + ds - 1) / ds;
  Gather3 g{gvertex, gnormal, rgb, out_pts, gnormal ? out_nrm : nullptr, rgb ? out_rgb : nullptr};
  return gs_compact((int64_t)Hl * Wl, PredFrameLattice{depth, W, ds, Wl}, EmitFrameLattice{W, ds, Wl, g},
                    count_out, 0, -1, scratch, gs_stream(stream));
}

// ICP source of a frame WITHOUT compaction: slot e of the [::ds, ::ds] lattice (raster order) holds the
// global vertex R v + t of its pixel (the arithmetic of gs_global_maps_f32) or NaN when the pixel has no
// depth.  gs_icp_*'s grid path ignores NaN source points, so the lattice can be fed to it as is: one
// launch instead of global maps + an ordered compaction (4 launches) per frame.
__global__ void __launch_bounds__(256) gs_lattice_source_kernel(const float* __restrict__ vertex,
                                                                const float* __restrict__ depth,
                                                                const float* __restrict__ pose16, int W, int ds, int Wl,
                                                                int64_t n_lat, float* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n_lat) return;
  const int64_t p = (e / Wl) * ds * (int64_t)W + (e % Wl) * ds;
  float g0 = __builtin_nanf(""), g1 = g0, g2 = g0;
  if (depth[p] > 0.0f) {
    float T[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = pose16[i];
    gs_rigid_fma(T, vertex[3 * p], vertex[3 * p + 1], vertex[3 * p + 2], g0, g1, g2);
  }
  out[3 * e] = g0; out[3 * e + 1] = g1; out[3 * e + 2] = g2;
}
extern "C" int gs_lattice_source_f32(const float* vertex, const float* depth, const float* pose16, int H, int W,
                                     int ds, float* out_pts, void* stream) {
  GS_REQUIRE(H > 0 && W > 0 && ds > 0 && vertex && depth && pose16 && out_pts, "bad arguments");
  const int Hl = (H + ds - 1) / ds, Wl = (W + ds - 1) / ds;
  const int64_t n_lat = (int64_t)Hl * Wl;
  hipLaunchKernelGGL(gs_lattice_source_kernel, dim3((unsigned)gs_ceil_div(n_lat, 256)), dim3(256), 0, gs_stream(stream),
                     vertex, depth, pose16, W, ds, Wl, n_lat, out_pts);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

// backward of gs_downsample_frame_f32 (points): scatter the compact adjoints back to their pixels
struct EmitFrameLatticeScatter {
  int W, ds, Wl;
  const float* pts_bar;
  float* gvertex_bar;
  __device__ void operator()(int64_t e, int64_t pos) const {
    const int64_t h = (e / Wl) * ds, w = (e % Wl) * ds, p = h * W + w;
#pragma unroll
    for (int k = 0; k < 3; ++k) gvertex_bar[3 * p + k] = pts_bar[3 * pos + k];
  }
};

extern "C" int gs_downsample_frame_backward_f32(const float* pts_bar, const float* depth, int H, int W, int ds,
                                                float* gvertex_bar, void* scratch, void* stream) {
  GS_REQUIRE(H > 0 && W > 0 && ds > 0 && pts_bar && depth && gvertex_bar && scratch, "bad arguments");
  hipStream_t st = gs_stream(stream);
  GS_HIP(hipMemsetAsync(gvertex_bar, 0, 12 * (size_t)H * W, st));
  const int Hl = (H + ds - 1) / ds, Wl = (W + ds - 1) / ds;
  // the count lands in the (unused) first slot after the compaction scratch
  int64_t* cnt = reinterpret_cast<int64_t*>(reinterpret_cast<char*>(scratch) + gs_cp_scratch_bytes((int64_t)H * W));
  return gs_compact((int64_t)Hl * Wl, PredFrameLattice{depth, W, ds, Wl},
                    EmitFrameLatticeScatter{W, ds, Wl, pts_bar, gvertex_bar}, cnt, 0, -1, scratch, st);
}

__global__ void __launch_bounds__(256) gs_similar_rows_kernel(
    const int64_t* __restrict__ rows, int64_t n_rows, const float* __restrict__ points,
    const float* __restrict__ normals, const float* __restrict__ gvertex,
    const float* __restrict__ gnormal, int W, float dist_th, float dot_th, uint8_t* __restrict__ mask) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n_rows) return;
  const int64_t n = rows[4 * r + 1], p = rows[4 * r + 2] * W + rows[4 * r + 3];
  mask[r] = gs_is_similar(points, normals, gvertex, gnormal, n, p, dist_th, dot_th) ? 1 : 0;
}

extern "C" int gs_similar_rows_f32(const int64_t* rows, int64_t n_rows, const float* points,
                                   const float* normals, const float* gvertex, const float* gnormal,
                                   int W, float dist_th, float dot_th, uint8_t* mask, void* stream) {
  GS_REQUIRE(n_rows >= 0 && W > 0, "bad sizes");
  if (n_rows == 0) return GS_OK;
  GS_REQUIRE(rows && points && normals && gvertex && gnormal && mask, "NULL pointer");
  hipLaunchKernelGGL(gs_similar_rows_kernel, dim3((unsigned)gs_ceil_div(n_rows, 256)), dim3(256), 0,
                     gs_stream(stream), rows, n_rows, points, normals, gvertex, gnormal, W, dist_th,
                     dot_th, mask);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

// ---------------------------------------------------------------- K5c: best per pixel --
// Two order-independent (hence deterministic) atomic passes replace the reference's
// torch.unique(dim=0) sort: pass 1 takes the per-pixel minimum of the (1/cc, ray) key, pass 2
// the minimum map index among the rows that attain it.
// best_pix starts as -1 = 0xffffffff: the winners are then reduced with an UNSIGNED atomicMin, so a
// pixel nobody claims keeps -1 ("no correspondence") and no fix-up pass is needed.
__global__ void __launch_bounds__(256) gs_fill_i32_kernel(int32_t* p, int64_t n, int32_t v) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ void __launch_bounds__(256) gs_assoc_init_kernel(uint64_t* __restrict__ key_pix,
                                                            int32_t* __restrict__ best_pix, int64_t P) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < P) {
    key_pix[i] = ~0ull;
    best_pix[i] = -1;
  }
}

__global__ void __launch_bounds__(256) gs_rows_key_kernel(
    const int64_t* __restrict__ rows, int64_t n_rows, const float* __restrict__ points,
    const float* __restrict__ ccounts, const float* __restrict__ gvertex, int W,
    uint64_t* __restrict__ key_row, unsigned long long* __restrict__ key_pix) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n_rows) return;
  const int64_t n = rows[4 * r + 1], p = rows[4 * r + 2] * W + rows[4 * r + 3];
  const uint64_t k = gs_assoc_key(points, ccounts, gvertex, n, p);
  key_row[r] = k;
  atomicMin(&key_pix[p], (unsigned long long)k);
}
__global__ void __launch_bounds__(256) gs_rows_pick_kernel(
    const int64_t* __restrict__ rows, int64_t n_rows, int W, const uint64_t* __restrict__ key_row,
    const uint64_t* __restrict__ key_pix, int32_t* __restrict__ best_pix) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n_rows) return;
  const int64_t n = rows[4 * r + 1], p = rows[4 * r + 2] * W + rows[4 * r + 3];
  if (key_row[r] == key_pix[p]) atomicMin(reinterpret_cast<unsigned*>(&best_pix[p]), (unsigned)n);
}

struct PredBest {
  const int32_t* best;
  __device__ bool operator()(int64_t p) const { return best[p] >= 0; }
};
struct EmitBestRow {
  const int32_t* best;
  int W;
  int64_t b;
  int64_t* rows;
  __device__ void operator()(int64_t p, int64_t pos) const {
    rows[4 * pos] = b;
    rows[4 * pos + 1] = best[p];
    rows[4 * pos + 2] = p / W;
    rows[4 * pos + 3] = p % W;
  }
};

extern "C" int gs_best_table_i64(const int32_t* best_pix, int H, int W, int64_t b,
                                 int64_t* rows_out, int64_t* count_out, void* scratch, void* stream) {
  GS_REQUIRE(H > 0 && W > 0 && best_pix && count_out && scratch, "bad arguments");
  return gs_compact((int64_t)H * W, PredBest{best_pix}, EmitBestRow{best_pix, W, b, rows_out}, count_out, 0,
                    -1, scratch, gs_stream(stream));
}

static inline unsigned gs_blocks(int64_t n) { return (unsigned)gs_ceil_div(n > 0 ? n : 1, 256); }

extern "C" int gs_best_unique_rows_f32(const int64_t* rows, int64_t n_rows, const float* points,
                                       const float* ccounts, const float* gvertex, int H, int W,
                                       int64_t b, int32_t* best_pix, int64_t* rows_out,
                                       int64_t* count_out, void* scratch, void* stream) {
  GS_REQUIRE(n_rows >= 0 && H > 0 && W > 0 && best_pix && count_out && scratch, "bad arguments");
  hipStream_t st = gs_stream(stream);
  const int64_t P = (int64_t)H * W;
  // scratch: [compaction | key_pix u64[P] | key_row u64[n_rows]]
  char* base = reinterpret_cast<char*>(scratch) + gs_cp_scratch_bytes(P > n_rows ? P : n_rows);
  uint64_t* key_pix = reinterpret_cast<uint64_t*>(base);
  uint64_t* key_row = reinterpret_cast<uint64_t*>(base + gs_align(8 * (size_t)P));
  hipLaunchKernelGGL(gs_assoc_init_kernel, dim3(gs_blocks(P)), dim3(256), 0, st, key_pix, best_pix, P);
  if (n_rows > 0) {
    hipLaunchKernelGGL(gs_rows_key_kernel, dim3(gs_blocks(n_rows)), dim3(256), 0, st, rows, n_rows, points,
                       ccounts, gvertex, W, key_row, reinterpret_cast<unsigned long long*>(key_pix));
    hipLaunchKernelGGL(gs_rows_pick_kernel, dim3(gs_blocks(n_rows)), dim3(256), 0, st, rows, n_rows, W,
                       key_row, key_pix, best_pix);
  }
  GS_LAUNCH_CHECK();
  if (rows_out) return gs_best_table_i64(best_pix, H, W, b, rows_out, count_out, scratch, stream);
  return GS_OK;
}

// fused path: pix[] straight to best_pix[] (no tables)
__global__ void __launch_bounds__(256) gs_assoc_key_kernel(
    const int32_t* __restrict__ pix, GsCount n_map_c, const float* __restrict__ points,
    const float* __restrict__ normals, const float* __restrict__ ccounts,
    const float* __restrict__ gvertex, const float* __restrict__ gnormal, float dist_th, float dot_th,
    uint64_t* __restrict__ key_pt, unsigned long long* __restrict__ key_pix,
    uint8_t* __restrict__ similar) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= gs_count(n_map_c)) return;
  const int32_t p = pix[n];
  bool sim = false;
  uint64_t k = ~0ull;
  if (p >= 0 && gs_is_similar(points, normals, gvertex, gnormal, n, p, dist_th, dot_th)) {
    sim = true;
    k = gs_assoc_key(points, ccounts, gvertex, n, p);
    atomicMin(&key_pix[p], (unsigned long long)k);
  }
  key_pt[n] = k;
  if (similar) similar[n] = sim ? 1 : 0;
}
__global__ void __launch_bounds__(256) gs_assoc_pick_kernel(
    const int32_t* __restrict__ pix, GsCount n_map_c, const uint64_t* __restrict__ key_pt,
    const uint64_t* __restrict__ key_pix, int32_t* __restrict__ best_pix) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= gs_count(n_map_c)) return;
  const uint64_t k = key_pt[n];
  if (k == ~0ull) return;  // not similar (a real key can never be all ones: ray is not NaN-coded)
  const int32_t p = pix[n];
  if (k == key_pix[p]) atomicMin(reinterpret_cast<unsigned*>(&best_pix[p]), (unsigned)n);
}

static int associate(const int32_t* pix, GsCount n_map_c, const float* points, const float* normals,
                     const float* ccounts, const float* gvertex, const float* gnormal, int H, int W,
                     float dist_th, float dot_th, int32_t* best_pix, uint8_t* similar, void* scratch,
                     void* stream) {
  const int64_t n_map = n_map_c.host;
  GS_REQUIRE(n_map >= 0 && H > 0 && W > 0 && best_pix && scratch, "bad arguments");
  GS_REQUIRE(n_map < 0x7fffffff, "map too large for int32 indices");
  hipStream_t st = gs_stream(stream);
  const int64_t P = (int64_t)H * W;
  char* base = reinterpret_cast<char*>(scratch) + gs_cp_scratch_bytes(P > n_map ? P : n_map);
  uint64_t* key_pix = reinterpret_cast<uint64_t*>(base);
  uint64_t* key_pt = reinterpret_cast<uint64_t*>(base + gs_align(8 * (size_t)P));
  // 3 launches: map rows 4+12+12+4 B read, 8 B key write, 4+8 B re-read; frame gathers 24 B per
  // active point (counted as all points); per pixel 12 B init + 8 B key traffic
  GsProf prof(GS_PROF_ASSOC, 76.0 * (double)n_map + 20.0 * (double)P, st);
  hipLaunchKernelGGL(gs_assoc_init_kernel, dim3(gs_blocks(P)), dim3(256), 0, st, key_pix, best_pix, P);
  if (n_map > 0) {
    GS_REQUIRE(pix && points && normals && ccounts && gvertex && gnormal, "NULL pointer");
    hipLaunchKernelGGL(gs_assoc_key_kernel, dim3(gs_blocks(n_map)), dim3(256), 0, st, pix, n_map_c, points,
                       normals, ccounts, gvertex, gnormal, dist_th, dot_th, key_pt,
                       reinterpret_cast<unsigned long long*>(key_pix), similar);
    hipLaunchKernelGGL(gs_assoc_pick_kernel, dim3(gs_blocks(n_map)), dim3(256), 0, st, pix, n_map_c, key_pt,
                       key_pix, best_pix);
  }
  GS_LAUNCH_CHECK();
  return GS_OK;
}

extern "C" int gs_associate_f32(const int32_t* pix, int64_t n_map, const float* points,
                                const float* normals, const float* ccounts, const float* gvertex,
                                const float* gnormal, int H, int W, float dist_th, float dot_th,
                                int32_t* best_pix, uint8_t* similar, void* scratch, void* stream) {
  return associate(pix, GsCount{n_map, nullptr}, points, normals, ccounts, gvertex, gnormal, H, W, dist_th,
                   dot_th, best_pix, similar, scratch, stream);
}
extern "C" int gs_associate_dc_f32(const int32_t* pix, int64_t n_map_bound, const int64_t* n_map_dev,
                                   const float* points, const float* normals, const float* ccounts,
                                   const float* gvertex, const float* gnormal, int H, int W, float dist_th,
                                   float dot_th, int32_t* best_pix, uint8_t* similar, void* scratch,
                                   void* stream) {
  GS_REQUIRE(n_map_dev, "NULL device count");
  return associate(pix, GsCount{n_map_bound, n_map_dev}, points, normals, ccounts, gvertex, gnormal, H, W,
                   dist_th, dot_th, best_pix, similar, scratch, stream);
}

__global__ void __launch_bounds__(256) gs_rows_to_best_kernel(const int64_t* __restrict__ rows,
                                                              int64_t n_rows, int W,
                                                              int32_t* __restrict__ best_pix) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n_rows) return;
  best_pix[rows[4 * r + 2] * W + rows[4 * r + 3]] = (int32_t)rows[4 * r + 1];
}

extern "C" int gs_rows_to_best_pix(const int64_t* rows, int64_t n_rows, int H, int W,
                                   int32_t* best_pix, void* stream) {
  GS_REQUIRE(n_rows >= 0 && H > 0 && W > 0 && best_pix, "bad arguments");
  hipStream_t st = gs_stream(stream);
  const int64_t P = (int64_t)H * W;
  hipLaunchKernelGGL(gs_fill_i32_kernel, dim3(gs_blocks(P)), dim3(256), 0, st, best_pix, P, -1);
  if (n_rows > 0)
    hipLaunchKernelGGL(gs_rows_to_best_kernel, dim3(gs_blocks(n_rows)), dim3(256), 0, st, rows, n_rows, W,
                       best_pix);
  GS_LAUNCH_CHECK();
  return GS_OK;
}
