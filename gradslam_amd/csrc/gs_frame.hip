// gs_frame.hip — K1: depth -> vertex / normal / alpha / valid, local -> global maps; plus the
// library-wide plumbing (error string, ABI version, scratch sizing, tile-scan kernel).
// HBM-bound: 4 B/px read, 32-56 B/px written; the 3x3 (really 2x2 forward-difference)
// neighbourhood of the normal is served from an LDS depth tile with a one-pixel halo.
#include <stdarg.h>
#include <string.h>

#include "gs_compact.h"

// ------------------------------------------------------------------ plumbing -----------
static thread_local char g_err[512] = "";

void gs_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- optional kernel timing (see gs_common.h) ----
#include <mutex>
#include <vector>
std::atomic<bool> g_gs_prof_on{false};
namespace {
std::mutex g_prof_mu;  // the record table is shared by every host thread that enqueues work
struct ProfRec { hipEvent_t a, b; int kind; double work; bool closed; int launches; };
std::vector<ProfRec> g_prof;
size_t g_prof_used = 0;
double g_prof_ms[GS_PROF_KINDS], g_prof_work[GS_PROF_KINDS];
int64_t g_prof_n[GS_PROF_KINDS];
}  // namespace

int gs_prof_open(int kind, double work, hipStream_t st, int launches) {
  std::lock_guard<std::mutex> lock(g_prof_mu);
  if (!g_gs_prof_on.load() || g_prof_used >= g_prof.size()) return -1;
  ProfRec& r = g_prof[g_prof_used];
  r.kind = kind; r.work = work; r.closed = false; r.launches = launches;
  if (hipEventRecord(r.a, st) != hipSuccess) return -1;
  return (int)g_prof_used++;
}
void gs_prof_close(int slot, hipStream_t st) {
  std::lock_guard<std::mutex> lock(g_prof_mu);
  if ((size_t)slot < g_prof_used && hipEventRecord(g_prof[slot].b, st) == hipSuccess) g_prof[slot].closed = true;
}

extern "C" int gs_profile_begin(int max_records) {
  GS_REQUIRE(max_records > 0 && max_records <= (1 << 20), "max_records out of range");
  std::lock_guard<std::mutex> lock(g_prof_mu);
  while (g_prof.size() < (size_t)max_records) {
    ProfRec r;
    GS_HIP(hipEventCreate(&r.a));
    GS_HIP(hipEventCreate(&r.b));
    r.kind = 0; r.work = 0; r.closed = false; r.launches = 1;
    g_prof.push_back(r);
  }
  g_prof_used = 0;
  for (int k = 0; k < GS_PROF_KINDS; ++k) { g_prof_ms[k] = 0; g_prof_work[k] = 0; g_prof_n[k] = 0; }
  g_gs_prof_on = true;
  return GS_OK;
}
extern "C" int gs_profile_end(void) {
  g_gs_prof_on = false;
  GS_HIP(hipDeviceSynchronize());
  std::lock_guard<std::mutex> lock(g_prof_mu);
  for (size_t i = 0; i < g_prof_used; ++i) {
    const ProfRec& r = g_prof[i];
    if (!r.closed) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
    g_prof_ms[r.kind] += ms; g_prof_work[r.kind] += r.work; g_prof_n[r.kind] += r.launches;
  }
  return GS_OK;
}
extern "C" int gs_profile_read(int kind, double* ms_total, int64_t* launches, double* work_total) {
  GS_REQUIRE(kind >= 0 && kind < GS_PROF_KINDS && ms_total && launches && work_total, "bad arguments");
  std::lock_guard<std::mutex> lock(g_prof_mu);
  *ms_total = g_prof_ms[kind]; *launches = g_prof_n[kind]; *work_total = g_prof_work[kind];
  return GS_OK;
}

extern "C" int gs_abi_version(void) { return GS_ABI_VERSION; }
extern "C" const char* gs_last_error(void) { return g_err; }

extern "C" int64_t gs_scratch_bytes(int64_t n_map, int64_t n_pix) {
  int64_t n = n_map > n_pix ? n_map : n_pix;
  // compaction scratch + per-pixel and per-point 64-bit association keys
  return (int64_t)(gs_cp_scratch_bytes(n) + gs_align(8 * (size_t)(n_pix > 0 ? n_pix : 1)) +
                   gs_align(8 * (size_t)(n_map > 0 ? n_map : 1)) + 4096);
}

__global__ void gs_cp_scan_tiles_kernel(const int32_t* __restrict__ tile_counts, int64_t ntiles,
                                        int64_t* __restrict__ tile_offsets,
                                        int64_t* __restrict__ count_out, GsCount base_count) {
  __shared__ int smem[1024 / GS_WAVE + 1];
  __shared__ int64_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int64_t t0 = 0; t0 < ntiles; t0 += 1024) {
    int64_t t = t0 + threadIdx.x;
    int c = (t < ntiles) ? tile_counts[t] : 0;
    int total;
    int excl = gs_block_excl_scan<1024>(c, smem, &total);
    if (t < ntiles) tile_offsets[t] = carry + excl;
    __syncthreads();
    if (threadIdx.x == 0) carry += total;
    __syncthreads();
  }
  if (threadIdx.x == 0 && count_out) count_out[0] = gs_count(base_count) + carry;
}

// ------------------------------------------------------------------ K1 -----------------
constexpr int FM_TW = 64;  // tile width  (one wave per row)
constexpr int FM_TH = 8;   // tile height (each of the 4 waves handles 2 rows)
constexpr int FM_LW = FM_TW + 2;
constexpr int FM_LH = FM_TH + 2;

// local vertex of pixel (h, w) from its depth: einsum(Kinv, (u, v, 1)) as FMA chain, then
// * depth, then * valid (structures/rgbdimages.py:662-679).
GS_DEV void fm_vertex(float d, int h, int w, const GsKinv& k, float& x, float& y, float& z) {
  float u = (float)w, v = (float)h;
  float ax = k.k00 * u;
  ax = gs_fma(0.0f, v, ax);
  ax = gs_fma(k.k02, 1.0f, ax);
  float ay = 0.0f * u;
  ay = gs_fma(k.k11, v, ay);
  ay = gs_fma(k.k12, 1.0f, ay);
  float az = 0.0f * u;
  az = gs_fma(0.0f, v, az);
  az = gs_fma(1.0f, 1.0f, az);
  float validf = d > 0.0f ? 1.0f : 0.0f;
  x = (ax * d) * validf;
  y = (ay * d) * validf;
  z = (az * d) * validf;
}

// a 12-byte row assembled in registers and stored by ONE instruction: three dword stores per row were the pattern that
// cost the merge kernel 40 % before it was fixed
struct GsRow3 {
  float x, y, z;
};
GS_DEV void frame_maps_body(const float* __restrict__ depth, const float* __restrict__ K16, int H, int W,
                            float two_sigma_sq, float* __restrict__ vertex, float* __restrict__ normal,
                            float* __restrict__ alpha, uint8_t* __restrict__ valid,
                            uint64_t* __restrict__ key_pix = nullptr, int32_t* __restrict__ best_pix = nullptr) {
  __shared__ float tile[FM_LH][FM_LW];
  const int w_base = blockIdx.x * FM_TW, h_base = blockIdx.y * FM_TH;
  // stage depth tile with a one-pixel halo on every side (coordinates clamped into the image;
  // clamped duplicates are never used for a pixel that exists)
  for (int i = threadIdx.x; i < FM_LH * FM_LW; i += 256) {
    int lh = i / FM_LW, lw = i - lh * FM_LW;
    int gh = h_base + lh - 1, gw = w_base + lw - 1;
    gh = gh < 0 ? 0 : (gh > H - 1 ? H - 1 : gh);
    gw = gw < 0 ? 0 : (gw > W - 1 ? W - 1 : gw);
    tile[lh][lw] = depth[(size_t)gh * W + gw];
  }
  __syncthreads();
  const GsKinv k = gs_kinv(K16);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int w = w_base + lane;
  if (w >= W) return;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int lh = wave * 2 + r;
    const int h = h_base + lh;
    if (h >= H) continue;
    const size_t p = (size_t)h * W + w;
    const float d = tile[lh + 1][lane + 1];
    float vx, vy, vz;
    fm_vertex(d, h, w, k, vx, vy, vz);
    if (vertex) reinterpret_cast<GsRow3*>(vertex)[p] = GsRow3{vx, vy, vz};   // one 12-byte store per row (dwordx3)
    if (valid) valid[p] = d > 0.0f ? 1 : 0;
    if (alpha) alpha[p] = gs_alpha_of(vx, vy, vz, two_sigma_sq, 1e-7f);
    if (key_pix) { key_pix[p] = ~0ull; best_pix[p] = -1; }   // (GsPixelTables: the map update's per-pixel tables)
    if (normal) {
      // forward differences; the last column / row reuse the previous difference
      // (structures/rgbdimages.py:724-731)
      const int w0 = (w < W - 1) ? w : W - 2;
      const int h0 = (h < H - 1) ? h : H - 2;
      const int lw0 = w0 - w_base + 1, lh0 = h0 - h_base + 1;
      float a0x, a0y, a0z, a1x, a1y, a1z, b0x, b0y, b0z, b1x, b1y, b1z;
      fm_vertex(tile[lh + 1][lw0], h, w0, k, a0x, a0y, a0z);
      fm_vertex(tile[lh + 1][lw0 + 1], h, w0 + 1, k, a1x, a1y, a1z);
      fm_vertex(tile[lh0][lane + 1], h0, w, k, b0x, b0y, b0z);
      fm_vertex(tile[lh0 + 1][lane + 1], h0 + 1, w, k, b1x, b1y, b1z);
      const float dhx = a1x - a0x, dhy = a1y - a0y, dhz = a1z - a0z;
      const float dvx = b1x - b0x, dvy = b1y - b0y, dvz = b1z - b0z;
      // torch.cross: fma(a1, b2, -(a2*b1))
      const float nx = gs_fma(dhy, dvz, -(dhz * dvy));
      const float ny = gs_fma(dhz, dvx, -(dhx * dvz));
      const float nz = gs_fma(dhx, dvy, -(dhy * dvx));
      const float nrm = gs_norm3(nx, ny, nz);
      const float den = (nrm == 0.0f) ? 1.0f : nrm;
      const float validf = d > 0.0f ? 1.0f : 0.0f;
      reinterpret_cast<GsRow3*>(normal)[p] = GsRow3{(nx / den) * validf, (ny / den) * validf, (nz / den) * validf};
    }
  }
}

__global__ void __launch_bounds__(256) gs_frame_maps_kernel(
    const float* __restrict__ depth, const float* __restrict__ K16, int H, int W, float two_sigma_sq,
    float* __restrict__ vertex, float* __restrict__ normal, float* __restrict__ alpha,
    uint8_t* __restrict__ valid) {
  frame_maps_body(depth, K16, H, W, two_sigma_sq, vertex, normal, alpha, valid);
}
// blockIdx.z = frame of a contiguous (n_frames, H, W) stack; slices z >= n_frames (if any) zero-fill the regions of
// `job` (independent bytes: they ride along instead of costing a launch of their own)
__global__ void __launch_bounds__(256) gs_frame_maps_batch_kernel(
    const float* __restrict__ depth, int64_t stride_seq, int64_t stride_frame, const float* __restrict__ K16,
    int frames_per_K, int H, int W, float two_sigma_sq, float* __restrict__ vertex, float* __restrict__ normal,
    float* __restrict__ alpha, const int n_frames, const GsClearJob job, const GsPixelTables tabs) {
  if ((int)blockIdx.z >= n_frames) {
    const size_t blk = ((size_t)(blockIdx.z - n_frames) * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const size_t n16 = job.bytes / 16, per = (n16 + 256 * GS_CLEAR_ITEMS - 1) / (256 * GS_CLEAR_ITEMS);
    const size_t r = blk / per;
    if (r >= (size_t)job.n) return;
    float4* dst = reinterpret_cast<float4*>(job.ptr[r]);
    const size_t base = (blk % per) * 256 * GS_CLEAR_ITEMS + threadIdx.x;
#pragma unroll
    for (int u = 0; u < GS_CLEAR_ITEMS; ++u) {
      const size_t i = base + (size_t)u * 256;
      if (i < n16) dst[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    return;
  }
  const size_t f = blockIdx.z, P = (size_t)H * W;
  const size_t b = f / frames_per_K, l = f % frames_per_K;
  const bool tb = (int)f < tabs.n;
  if (tb && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    tabs.any_flag[f][0] = 0;
    tabs.any_flag[f][1] = 0;   // ("some pixel has two rows with the same key", gs_fuse.hip)
    if (f == 0 && tabs.call_flag) *tabs.call_flag = 0;
  }
  frame_maps_body(depth + b * stride_seq + l * stride_frame, K16 + 16 * b, H, W, two_sigma_sq, vertex + 3 * f * P,
                  normal ? normal + 3 * f * P : nullptr, alpha ? alpha + f * P : nullptr, nullptr,
                  tb ? tabs.key_pix[f] : nullptr, tb ? tabs.best_pix[f] : nullptr);
}

extern "C" int gs_frame_maps_f32(const float* depth, const float* K16, int H, int W,
                                 float two_sigma_sq, float* vertex, float* normal, float* alpha,
                                 uint8_t* valid, void* stream) {
  GS_REQUIRE(depth && K16, "depth and K16 must not be NULL");
  GS_REQUIRE(H >= 2 && W >= 2, "image must be at least 2x2");
  dim3 grid((unsigned)gs_ceil_div(W, FM_TW), (unsigned)gs_ceil_div(H, FM_TH));
  const double bytes = (double)H * W * (4.0 + (vertex ? 12 : 0) + (normal ? 12 : 0) + (alpha ? 4 : 0) + (valid ? 1 : 0));
  GsProf prof(GS_PROF_FRAME, bytes, gs_stream(stream));
  hipLaunchKernelGGL(gs_frame_maps_kernel, grid, dim3(256), 0, gs_stream(stream), depth, K16, H, W,
                     two_sigma_sq, vertex, normal, alpha, valid);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

int gs_frame_maps_batch_clear(const float* depth, int64_t depth_stride_seq, int64_t depth_stride_frame, const float* K16,
                              int n_frames, int frames_per_K, int H, int W, float two_sigma_sq, float* vertex,
                              float* normal, float* alpha, const GsClearJob* job, void* stream,
                              const GsPixelTables* tables) {
  GS_REQUIRE(depth && K16 && vertex, "depth, K16 and vertex must not be NULL");
  GsPixelTables tabs;
  tabs.n = 0; tabs.call_flag = nullptr;
  if (tables && tables->n > 0) {
    GS_REQUIRE(tables->n <= GS_MAX_BATCH && tables->n <= n_frames, "bad pixel tables");
    tabs = *tables;
  }
  GS_REQUIRE(H >= 2 && W >= 2, "image must be at least 2x2");
  GS_REQUIRE(n_frames > 0 && n_frames <= 65535 && frames_per_K > 0 && n_frames % frames_per_K == 0, "bad frame count");
  GS_REQUIRE(depth_stride_frame >= (int64_t)H * W && depth_stride_seq >= 0, "bad depth strides");
  dim3 grid((unsigned)gs_ceil_div(W, FM_TW), (unsigned)gs_ceil_div(H, FM_TH), (unsigned)n_frames);
  GsClearJob none;
  none.n = 0; none.bytes = 0;
  if (job && job->n > 0 && job->bytes > 0) {
    GS_REQUIRE(job->n <= GS_MAX_BATCH && job->bytes % 16 == 0, "bad clear job");
    const int64_t per = gs_ceil_div((int64_t)(job->bytes / 16), 256 * GS_CLEAR_ITEMS);
    const int64_t slices = gs_ceil_div(per * job->n, (int64_t)grid.x * grid.y);
    GS_REQUIRE(n_frames + slices <= 65535, "clear job too large for the launch");
    grid.z += (unsigned)slices;
  } else {
    job = &none;
  }
  const double bytes = (double)n_frames * H * W * (4.0 + 12.0 + (normal ? 12 : 0) + (alpha ? 4 : 0)) +
                       (double)tabs.n * H * W * 12.0;
  GsProf prof(GS_PROF_FRAME, bytes, gs_stream(stream));
  hipLaunchKernelGGL(gs_frame_maps_batch_kernel, grid, dim3(256), 0, gs_stream(stream), depth, depth_stride_seq,
                     depth_stride_frame, K16, frames_per_K, H, W, two_sigma_sq, vertex, normal, alpha, n_frames, *job, tabs);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

extern "C" int gs_frame_maps_batch_f32(const float* depth, int64_t depth_stride_seq, int64_t depth_stride_frame,
                                       const float* K16, int n_frames, int frames_per_K, int H, int W,
                                       float two_sigma_sq, float* vertex, float* normal, float* alpha, void* stream) {
  return gs_frame_maps_batch_clear(depth, depth_stride_seq, depth_stride_frame, K16, n_frames, frames_per_K, H, W,
                                   two_sigma_sq, vertex, normal, alpha, nullptr, stream);
}

// local -> global: R v + t re-masked (rgbdimages.py:700-708), R n (rgbdimages.py:760-762)
__global__ void __launch_bounds__(256) gs_global_maps_kernel(
    const float* __restrict__ vertex, const float* __restrict__ normal,
    const float* __restrict__ depth, const float* __restrict__ pose16, int64_t P,
    float* __restrict__ gvertex, float* __restrict__ gnormal) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  float T[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) T[i] = pose16[i];
  if (gvertex) {
    const float v0 = vertex[3 * p], v1 = vertex[3 * p + 1], v2 = vertex[3 * p + 2];
    const float validf = depth[p] > 0.0f ? 1.0f : 0.0f;
    float g0, g1, g2;
    gs_rigid_fma(T, v0, v1, v2, g0, g1, g2);
    gvertex[3 * p] = g0 * validf;
    gvertex[3 * p + 1] = g1 * validf;
    gvertex[3 * p + 2] = g2 * validf;
  }
  if (gnormal) {
    const float n0 = normal[3 * p], n1 = normal[3 * p + 1], n2 = normal[3 * p + 2];
    gnormal[3 * p] = gs_dot3_fma(T[0], T[1], T[2], n0, n1, n2);
    gnormal[3 * p + 1] = gs_dot3_fma(T[4], T[5], T[6], n0, n1, n2);
    gnormal[3 * p + 2] = gs_dot3_fma(T[8], T[9], T[10], n0, n1, n2);
  }
}

extern "C" int gs_global_maps_f32(const float* vertex, const float* normal, const float* depth,
                                  const float* pose16, int H, int W, float* gvertex,
                                  float* gnormal, void* stream) {
  GS_REQUIRE(H > 0 && W > 0, "empty image");
  GS_REQUIRE(!gvertex || (vertex && depth), "gvertex needs vertex and depth");
  GS_REQUIRE(!gnormal || normal, "gnormal needs normal");
  const int64_t P = (int64_t)H * W;
  hipStream_t st = gs_stream(stream);
  if (!pose16) {  // rgbdimages.py:683-685 / :747-749: no poses -> plain copy
    if (gvertex) GS_HIP(hipMemcpyAsync(gvertex, vertex, P * 12, hipMemcpyDeviceToDevice, st));
    if (gnormal) GS_HIP(hipMemcpyAsync(gnormal, normal, P * 12, hipMemcpyDeviceToDevice, st));
    return GS_OK;
  }
  GsProf prof(GS_PROF_FRAME, (double)P * (4.0 + (gvertex ? 24 : 0) + (gnormal ? 24 : 0)), st);
  hipLaunchKernelGGL(gs_global_maps_kernel, dim3((unsigned)gs_ceil_div(P, 256)), dim3(256), 0, st,
                     vertex, normal, depth, pose16, P, gvertex, gnormal);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

__global__ void __launch_bounds__(256) gs_alpha_kernel(const float* __restrict__ pts, int64_t n,
                                                       float two_sigma_sq, float eps,
                                                       float* __restrict__ alpha) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  alpha[i] = gs_alpha_of(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], two_sigma_sq, eps);
}

extern "C" int gs_alpha_f32(const float* points, int64_t n, float two_sigma_sq, float eps,
                            float* alpha, void* stream) {
  GS_REQUIRE(n >= 0, "negative n");
  if (n == 0) return GS_OK;
  GS_REQUIRE(points && alpha, "NULL pointer");
  hipLaunchKernelGGL(gs_alpha_kernel, dim3((unsigned)gs_ceil_div(n, 256)), dim3(256), 0,
                     gs_stream(stream), points, n, two_sigma_sq, eps, alpha);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

// Reverse mode of gs_alpha_f32 (slam/fusionutils.py:69-72 under PyTorch autograd): with S = |p|^2 and a = exp(-S / 2s^2),
// d a / d p = -a p / s^2 and d a / d s = a S / s^3 where the clamp passes the gradient (eps <= a <= 1.01, torch.clamp's
// rule), zero elsewhere.  p_bar[i] per point; the sigma adjoint is a sum over the points: sigma_terms[i] = a_bar a S (the
// caller adds them up and scales by 1 / s^3 -- one tiny reduction off the hot path, fixed order).
__global__ void __launch_bounds__(256) gs_alpha_bwd_kernel(const float* __restrict__ pts, int64_t n, float two_sigma_sq,
                                                           float eps, const float* __restrict__ alpha_bar,
                                                           float* __restrict__ pts_bar, float* __restrict__ sigma_terms) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
  float S = x * x + y * y;
  S = S + z * z;
  const float a = gs_expf_spec((-S) / two_sigma_sq);   // (the forward's operations) the un-clamped value decides whether the clamp passes
  const bool pass = a >= eps && a <= 1.01f;
  const float g = pass ? alpha_bar[i] * a : 0.0f;
  const float k = -2.0f / two_sigma_sq;               // -1 / s^2
  pts_bar[3 * i] = g * k * x;
  pts_bar[3 * i + 1] = g * k * y;
  pts_bar[3 * i + 2] = g * k * z;
  if (sigma_terms) sigma_terms[i] = g * S;
}

extern "C" int gs_alpha_backward_f32(const float* points, int64_t n, float two_sigma_sq, float eps,
                                     const float* alpha_bar, float* points_bar, float* sigma_terms, void* stream) {
  GS_REQUIRE(n >= 0, "negative n");
  if (n == 0) return GS_OK;
  GS_REQUIRE(points && alpha_bar && points_bar, "NULL pointer");
  hipLaunchKernelGGL(gs_alpha_bwd_kernel, dim3((unsigned)gs_ceil_div(n, 256)), dim3(256), 0, gs_stream(stream), points, n,
                     two_sigma_sq, eps, alpha_bar, points_bar, sigma_terms);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

// ------------------------------------------------------------------ K7: backward of K1 ---
// Reverse mode of gs_frame_maps_f32 w.r.t. depth (vertex, normal and alpha paths) and of
// gs_global_maps_f32 w.r.t. the local maps, matching PyTorch autograd through
// structures/rgbdimages.py:643-762 and slam/fusionutils.py:69-72.  Two passes keep it
// deterministic: pass 1 turns the normal adjoint of every pixel into the adjoints of its two
// forward differences, pass 2 GATHERS the (at most six) difference adjoints that touch a vertex.
GS_DEV void fb_ray(int h, int w, const GsKinv& k, float& rx, float& ry) {
  rx = k.k00 * (float)w + k.k02;
  ry = k.k11 * (float)h + k.k12;
}
GS_DEV void fb_vertex(const float* __restrict__ depth, int W, int h, int w, const GsKinv& k, float* v) {
  const float d = depth[(size_t)h * W + w];
  float rx, ry;
  fb_ray(h, w, k, rx, ry);
  const float m = d > 0.0f ? 1.0f : 0.0f;
  v[0] = rx * d * m; v[1] = ry * d * m; v[2] = d * m;
}

__global__ void __launch_bounds__(256) gs_frame_bwd_diff_kernel(const float* __restrict__ depth,
                                                                const float* __restrict__ K16, int H, int W,
                                                                const float* __restrict__ normal_bar,
                                                                float* __restrict__ dhdv_bar) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= (int64_t)H * W) return;
  const int h = (int)(p / W), w = (int)(p % W);
  const GsKinv k = gs_kinv(K16);
  const int w0 = (w < W - 1) ? w : W - 2, h0 = (h < H - 1) ? h : H - 2;
  float a0[3], a1[3], b0[3], b1[3];
  fb_vertex(depth, W, h, w0, k, a0);
  fb_vertex(depth, W, h, w0 + 1, k, a1);
  fb_vertex(depth, W, h0, w, k, b0);
  fb_vertex(depth, W, h0 + 1, w, k, b1);
  const float dh[3] = {a1[0] - a0[0], a1[1] - a0[1], a1[2] - a0[2]};
  const float dv[3] = {b1[0] - b0[0], b1[1] - b0[1], b1[2] - b0[2]};
  const float nr[3] = {dh[1] * dv[2] - dh[2] * dv[1], dh[2] * dv[0] - dh[0] * dv[2], dh[0] * dv[1] - dh[1] * dv[0]};
  const float nrm = sqrtf(nr[0] * nr[0] + nr[1] * nr[1] + nr[2] * nr[2]);
  const float m = depth[p] > 0.0f ? 1.0f : 0.0f;
  const float g[3] = {normal_bar[3 * p] * m, normal_bar[3 * p + 1] * m, normal_bar[3 * p + 2] * m};
  float nb[3];
  if (nrm > 0.0f) {  // n = nr / |nr|
    const float inv = 1.0f / nrm;
    const float nh[3] = {nr[0] * inv, nr[1] * inv, nr[2] * inv};
    const float dot = nh[0] * g[0] + nh[1] * g[1] + nh[2] * g[2];
    for (int c = 0; c < 3; ++c) nb[c] = (g[c] - nh[c] * dot) * inv;
  } else {  // divided by 1 in the forward pass
    for (int c = 0; c < 3; ++c) nb[c] = g[c];
  }
  // nr = dh x dv  ->  dh_bar = dv x nr_bar, dv_bar = nr_bar x dh
  float* o = dhdv_bar + 6 * p;
  o[0] = dv[1] * nb[2] - dv[2] * nb[1];
  o[1] = dv[2] * nb[0] - dv[0] * nb[2];
  o[2] = dv[0] * nb[1] - dv[1] * nb[0];
  o[3] = nb[1] * dh[2] - nb[2] * dh[1];
  o[4] = nb[2] * dh[0] - nb[0] * dh[2];
  o[5] = nb[0] * dh[1] - nb[1] * dh[0];
}

// KBAR: additionally reduce, per block, the adjoints of the four inverse-intrinsics entries the vertex depends on
// (x = (k00 u + k02) d, y = (k11 v + k12) d): [k00, k02, k11, k12]_bar = sum_p vb_x u d, vb_x d, vb_y v d, vb_y d.
template <bool KBAR>
__global__ void __launch_bounds__(256) gs_frame_bwd_depth_kernel(
    const float* __restrict__ depth, const float* __restrict__ K16, int H, int W, float two_sigma_sq,
    const float* __restrict__ vertex_bar, const float* __restrict__ alpha_bar, const float* __restrict__ dhdv_bar,
    float* __restrict__ depth_bar, double* __restrict__ kbar_partials) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool in = p < (int64_t)H * W;
  double kb[4] = {0.0, 0.0, 0.0, 0.0};
  if (in) {
    const int h = (int)(p / W), w = (int)(p % W);
    const GsKinv k = gs_kinv(K16);
    float vb[3] = {0.0f, 0.0f, 0.0f};
    if (vertex_bar)
      for (int c = 0; c < 3; ++c) vb[c] = vertex_bar[3 * p + c];
    const float d = depth[p];
    const float m = d > 0.0f ? 1.0f : 0.0f;
    float rx, ry;
    fb_ray(h, w, k, rx, ry);
    if (alpha_bar) {  // alpha = clamp(exp(-|v|^2 / (2 sigma^2)), 1e-7, 1.01)
      const float v[3] = {rx * d * m, ry * d * m, d * m};
      const float a = gs_alpha_of(v[0], v[1], v[2], two_sigma_sq, 1e-7f);
      if (a > 1e-7f && a < 1.01f) {
        const float f = alpha_bar[p] * a * (-2.0f / two_sigma_sq);
        for (int c = 0; c < 3; ++c) vb[c] += f * v[c];
      }
    }
    if (dhdv_bar) {
      auto DH = [&](int hh, int ww, int c) { return dhdv_bar[6 * ((int64_t)hh * W + ww) + c]; };
      auto DV = [&](int hh, int ww, int c) { return dhdv_bar[6 * ((int64_t)hh * W + ww) + 3 + c]; };
      for (int c = 0; c < 3; ++c) {
        float acc = 0.0f;
        // horizontal difference of pixel (h, w'): a1 = (h, min(w', W-2) + 1), a0 = (h, min(w', W-2))
        if (w >= 1) acc += DH(h, w - 1, c);
        if (w == W - 1) acc += DH(h, W - 1, c);
        if (w <= W - 2) acc -= DH(h, w, c);
        if (w == W - 2) acc -= DH(h, W - 1, c);
        // vertical difference of pixel (h', w)
        if (h >= 1) acc += DV(h - 1, w, c);
        if (h == H - 1) acc += DV(H - 1, w, c);
        if (h <= H - 2) acc -= DV(h, w, c);
        if (h == H - 2) acc -= DV(H - 1, w, c);
        vb[c] += acc;
      }
    }
    if (depth_bar) depth_bar[p] = m * (vb[0] * rx + vb[1] * ry + vb[2]);
    if (KBAR) {
      const double dm = (double)d * (double)m;
      kb[0] = (double)vb[0] * (double)w * dm; kb[1] = (double)vb[0] * dm;
      kb[2] = (double)vb[1] * (double)h * dm; kb[3] = (double)vb[1] * dm;
    }
  }
  if (KBAR) {  // fixed-order block sums (wave tree, then the 4 waves in order)
    __shared__ double red[4][256 / GS_WAVE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const double sum = gs_wave_sum_f64(kb[i]);
      if (lane == 0) red[i][wave] = sum;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
      const int i = threadIdx.x;
      kbar_partials[(int64_t)blockIdx.x * 4 + i] = ((red[i][0] + red[i][1]) + red[i][2]) + red[i][3];
    }
  }
}

// K_bar (4x4) from the block partials: inverse_intrinsics (geometry/projutils.py:437-449) has k00 = 1 / (fx + eps),
// k02 = -cx / (fx + eps) (same for y), so fx_bar = -k00_bar / f'^2 + k02_bar cx / f'^2, cx_bar = -k02_bar / f'.
__global__ void __launch_bounds__(256) gs_frame_bwd_kbar_kernel(const double* __restrict__ partials, int nblocks,
                                                                const float* __restrict__ K16,
                                                                float* __restrict__ K_bar16) {
  __shared__ double red[4][256];
  for (int i = 0; i < 4; ++i) {
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) s += partials[(int64_t)b * 4 + i];
    red[i][threadIdx.x] = s;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  double kb[4];
  for (int i = 0; i < 4; ++i) {
    double s = 0.0;
    for (int t = 0; t < 256; ++t) s += red[i][t];
    kb[i] = s;
  }
  const double eps = 1e-6, fx = (double)K16[0] + eps, fy = (double)K16[5] + eps, cx = K16[2], cy = K16[6];
  for (int i = 0; i < 16; ++i) K_bar16[i] = 0.0f;
  K_bar16[0] = (float)((-kb[0] + kb[1] * cx) / (fx * fx));
  K_bar16[2] = (float)(-kb[1] / fx);
  K_bar16[5] = (float)((-kb[2] + kb[3] * cy) / (fy * fy));
  K_bar16[6] = (float)(-kb[3] / fy);
}

extern "C" int64_t gs_frame_maps_backward_kbar_scratch_bytes(int H, int W) {
  return (int64_t)gs_align(8 * 4 * (size_t)gs_ceil_div((int64_t)H * W, 256));
}

extern "C" int gs_frame_maps_backward_f32(const float* depth, const float* K16, int H, int W, float two_sigma_sq,
                                          const float* vertex_bar, const float* normal_bar, const float* alpha_bar,
                                          float* depth_bar, float* scratch_6hw, float* K_bar16, void* kbar_scratch,
                                          void* stream) {
  GS_REQUIRE(depth && K16 && (depth_bar || K_bar16), "NULL pointer");
  GS_REQUIRE(H >= 2 && W >= 2, "image must be at least 2x2");
  GS_REQUIRE(!normal_bar || scratch_6hw, "normal_bar needs scratch of 6*H*W floats");
  GS_REQUIRE(!K_bar16 || kbar_scratch, "K_bar16 needs gs_frame_maps_backward_kbar_scratch_bytes() of scratch");
  hipStream_t st = gs_stream(stream);
  const unsigned nb = (unsigned)gs_ceil_div((int64_t)H * W, 256);
  if (normal_bar)
    hipLaunchKernelGGL(gs_frame_bwd_diff_kernel, dim3(nb), dim3(256), 0, st, depth, K16, H, W, normal_bar, scratch_6hw);
  if (K_bar16) {
    double* partials = reinterpret_cast<double*>(kbar_scratch);
    hipLaunchKernelGGL((gs_frame_bwd_depth_kernel<true>), dim3(nb), dim3(256), 0, st, depth, K16, H, W, two_sigma_sq,
                       vertex_bar, alpha_bar, normal_bar ? scratch_6hw : nullptr, depth_bar, partials);
    hipLaunchKernelGGL(gs_frame_bwd_kbar_kernel, dim3(1), dim3(256), 0, st, partials, (int)nb, K16, K_bar16);
  } else {
    hipLaunchKernelGGL((gs_frame_bwd_depth_kernel<false>), dim3(nb), dim3(256), 0, st, depth, K16, H, W, two_sigma_sq,
                       vertex_bar, alpha_bar, normal_bar ? scratch_6hw : nullptr, depth_bar, nullptr);
  }
  GS_LAUNCH_CHECK();
  return GS_OK;
}

__global__ void __launch_bounds__(256) gs_global_maps_bwd_kernel(const float* __restrict__ gvertex_bar,
                                                                 const float* __restrict__ gnormal_bar,
                                                                 const float* __restrict__ depth,
                                                                 const float* __restrict__ pose16, int64_t P,
                                                                 float* __restrict__ vertex_bar,
                                                                 float* __restrict__ normal_bar) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  float T[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) T[i] = pose16[i];
  if (vertex_bar) {  // g = (R v + t) * m
    const float m = depth[p] > 0.0f ? 1.0f : 0.0f;
    const float g[3] = {gvertex_bar[3 * p] * m, gvertex_bar[3 * p + 1] * m, gvertex_bar[3 * p + 2] * m};
    for (int c = 0; c < 3; ++c) vertex_bar[3 * p + c] = T[c] * g[0] + T[4 + c] * g[1] + T[8 + c] * g[2];
  }
  if (normal_bar) {  // gn = R n
    const float g[3] = {gnormal_bar[3 * p], gnormal_bar[3 * p + 1], gnormal_bar[3 * p + 2]};
    for (int c = 0; c < 3; ++c) normal_bar[3 * p + c] = T[c] * g[0] + T[4 + c] * g[1] + T[8 + c] * g[2];
  }
}

extern "C" int gs_global_maps_backward_f32(const float* gvertex_bar, const float* gnormal_bar, const float* depth,
                                           const float* pose16, int H, int W, float* vertex_bar, float* normal_bar,
                                           void* stream) {
  GS_REQUIRE(H > 0 && W > 0 && pose16, "bad arguments");
  GS_REQUIRE(!vertex_bar || (gvertex_bar && depth), "vertex_bar needs gvertex_bar and depth");
  GS_REQUIRE(!normal_bar || gnormal_bar, "normal_bar needs gnormal_bar");
  const int64_t P = (int64_t)H * W;
  hipLaunchKernelGGL(gs_global_maps_bwd_kernel, dim3((unsigned)gs_ceil_div(P, 256)), dim3(256), 0, gs_stream(stream),
                     gvertex_bar, gnormal_bar, depth, pose16, P, vertex_bar, normal_bar);
  GS_LAUNCH_CHECK();
  return GS_OK;
}
