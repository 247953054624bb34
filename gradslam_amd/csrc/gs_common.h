// gs_common.h — shared host/device helpers of libgradslam_hip (gfx950 only).
//
// Arithmetic contract: every float32 operation below is written
// out one rounding at a time and the library is compiled with -ffp-contract=off, so the
// compiler never fuses or reassociates; where the reference's CPU kernels use an FMA the
// code says __builtin_fmaf explicitly.  Division and sqrt are IEEE correctly rounded
// (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gradslam_hip.h"

#define GS_ABI_VERSION 2

// sequences per launch of the batched (multi-sequence) kernels; larger batches run in chunks of this size
constexpr int GS_MAX_BATCH = 8;
// Zero-fill riding along with another launch (extra blocks of 256 threads): `n` regions of `bytes` bytes each
// (16-byte aligned, a multiple of 16).  Used by the one-call step: the frame-map launch also clears the grid scratch
// of the localisation that follows, which then needs no clearing blocks of its own.
struct GsClearJob {
  int n;
  size_t bytes;
  char* ptr[GS_MAX_BATCH];
};
constexpr int GS_CLEAR_ITEMS = 8;  // 16-byte stores per thread of a clearing block (32 KB per block)
// Per-pixel tables of the map update that the frame-map launch of the one-call step initialises on the side (its threads
// are per pixel already): key = ~0, winner = -1 for frame f of the launch, the "any match" flag of that sequence and the
// flag of the call = 0.  The update then starts at its projection pass (no per-pixel pass of its own).
struct GsPixelTables {
  int n;   // frames with tables (0: none)
  uint64_t* key_pix[GS_MAX_BATCH];
  int32_t* best_pix[GS_MAX_BATCH];
  int32_t* any_flag[GS_MAX_BATCH];
  int32_t* call_flag;   // may be NULL (not the first chunk of the call)
};
// gs_frame_maps_batch_f32 + the clear job (+ the update's pixel tables) in the same launch (library-internal)
int gs_frame_maps_batch_clear(const float* depth, int64_t depth_stride_seq, int64_t depth_stride_frame, const float* K16,
                              int n_frames, int frames_per_K, int H, int W, float two_sigma_sq, float* vertex,
                              float* normal, float* alpha, const GsClearJob* job, void* stream,
                              const GsPixelTables* tables = nullptr);

// the batched map update behind gs_update_map_fusion_batch_f32 (gs_fuse.hip); tables_ready: see GsPixelTables
int gs_update_map_fusion_batch_impl(const gs_update_seq* seqs_host, int B, int H, int W, float dist_th, float dot_th,
                                    int renorm_all, void* stream, bool tables_ready);
int32_t* gs_update_map_call_flag(void* scratch_of_first_sequence);
void gs_update_map_tables(void* scratch, int32_t** any_flag, uint64_t** key_pix);

// ---------------------------------------------------------------- error plumbing -------
void gs_set_error(const char* fmt, ...);

#define GS_REQUIRE(cond, msg)                  \
  do {                                         \
    if (!(cond)) {                             \
      gs_set_error("%s: %s", __func__, msg);   \
      return GS_ERR_INVALID;                   \
    }                                          \
  } while (0)

#define GS_HIP(call)                                                            \
  do {                                                                          \
    hipError_t e_ = (call);                                                     \
    if (e_ != hipSuccess) {                                                     \
      gs_set_error("%s: %s -> %s", __func__, #call, hipGetErrorString(e_));     \
      return GS_ERR_HIP;                                                        \
    }                                                                           \
  } while (0)

#define GS_LAUNCH_CHECK() GS_HIP(hipGetLastError())

static inline hipStream_t gs_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int64_t gs_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t gs_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// ---------------------------------------------------------------- in-library timing ----
// Optional HIP-event timing of kernel launches on the stream they are launched on (bench.py's
// roofline line).  Disabled by default: a GsProf scope is then two predictable branches.
enum GsProfKind { GS_PROF_KNN = 0, GS_PROF_LINEARIZE = 1, GS_PROF_FRAME = 2, GS_PROF_PROJECT = 3,
                  GS_PROF_ASSOC = 4, GS_PROF_FUSE = 5, GS_PROF_COMPACT = 6, GS_PROF_SOLVE = 7, GS_PROF_ICP_FUSED = 8,
                  GS_PROF_KINDS = 9 };
extern std::atomic<bool> g_gs_prof_on;  // set between gs_profile_begin / _end; records are guarded by a mutex
int gs_prof_open(int kind, double work, hipStream_t st, int launches);
void gs_prof_close(int slot, hipStream_t st);
struct GsProf {
  int slot;
  hipStream_t st;
  // `launches` kernels are enqueued back to back between the two events (one event pair per kernel would
  // put two extra packets between consecutive kernels and inflate the per-launch time of launch-bound loops)
  GsProf(int kind, double work, hipStream_t s, int launches = 1) : slot(-1), st(s) {
    if (g_gs_prof_on) slot = gs_prof_open(kind, work, s, launches);
  }
  ~GsProf() {
    if (slot >= 0) gs_prof_close(slot, st);
  }
};

// ---------------------------------------------------------------- device math ----------
#define GS_DEV __device__ __forceinline__

GS_DEV float gs_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// FMA-chain dot (torch CPU bmm over large batches): a0*b0, then fma, fma.
GS_DEV float gs_dot3_fma(float a0, float a1, float a2, float b0, float b1, float b2) {
  float acc = a0 * b0;
  acc = gs_fma(a1, b1, acc);
  acc = gs_fma(a2, b2, acc);
  return acc;
}
// plain left-to-right dot (tiny matmuls and (a*b).sum(-1)).
GS_DEV float gs_dot3_plain(float a0, float a1, float a2, float b0, float b1, float b2) {
  float p0 = a0 * b0, p1 = a1 * b1, p2 = a2 * b2;
  float s = p0 + p1;
  return s + p2;
}
// tensor.norm(dim=-1) over 3 components: FMA chain then sqrt.
GS_DEV float gs_norm3(float x, float y, float z) {
  float acc = x * x;
  acc = gs_fma(y, y, acc);
  acc = gs_fma(z, z, acc);
  return __builtin_sqrtf(acc);
}

// Specified exp: same operation sequence as oracle/gs_oracle.c:gs_expf_spec.
GS_DEV float gs_expf_spec(float x) {
  if (x < -87.0f) return 0.0f;
  if (x > 88.0f) return __builtin_inff();
  float n = __builtin_rintf(x * 1.44269504088896341f);
  float r = gs_fma(-n, 0.693359375f, x);
  r = gs_fma(-n, -2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = gs_fma(p, r, 1.3981999507e-3f);
  p = gs_fma(p, r, 8.3334519073e-3f);
  p = gs_fma(p, r, 4.1665795894e-2f);
  p = gs_fma(p, r, 1.6666665459e-1f);
  p = gs_fma(p, r, 5.0000001201e-1f);
  float r2 = r * r;
  float y = gs_fma(p, r2, r);
  y = y + 1.0f;
  int32_t bits = __float_as_int(y) + (((int32_t)n) << 23);
  return __int_as_float(bits);
}

GS_DEV float gs_alpha_of(float vx, float vy, float vz, float two_sigma_sq, float eps) {
  float s = vx * vx + vy * vy;
  s = s + vz * vz;
  float a = gs_expf_spec((-s) / two_sigma_sq);
  a = a < eps ? eps : a;
  a = a > 1.01f ? 1.01f : a;
  return a;
}

struct GsKinv {
  float k00, k11, k02, k12;
};
// inverse_intrinsics (geometry/projutils.py:437-449)
GS_DEV GsKinv gs_kinv(const float* __restrict__ K) {
  const float eps = 1e-6f;
  float fx = K[0], fy = K[5], cx = K[2], cy = K[6];
  GsKinv r;
  r.k00 = 1.0f / (fx + eps);
  r.k11 = 1.0f / (fy + eps);
  r.k02 = (-1.0f * cx) / (fx + eps);
  r.k12 = (-1.0f * cy) / (fy + eps);
  return r;
}

// rigid transform of a point, sgemm-style FMA chain + t (geometryutils.py:781-794,
// rgbdimages.py:700-703).
GS_DEV void gs_rigid_fma(const float* __restrict__ T, float p0, float p1, float p2, float& o0,
                         float& o1, float& o2) {
  o0 = gs_dot3_fma(T[0], T[1], T[2], p0, p1, p2) + T[3];
  o1 = gs_dot3_fma(T[4], T[5], T[6], p0, p1, p2) + T[7];
  o2 = gs_dot3_fma(T[8], T[9], T[10], p0, p1, p2) + T[11];
}

// A size that may live on the device: `host` is always a valid UPPER BOUND (used for launch
// geometry and buffer sizes); when `dev` is non-NULL the kernels use min(*dev, host) as the actual
// element count, so that the host never has to read a count back (no sync inside the frame loop).
struct GsCount {
  int64_t host;
  const int64_t* dev;
};
GS_DEV int64_t gs_count(const GsCount c) {
  if (!c.dev) return c.host;
  // the count is the same for every lane: keep it in scalar registers (a vector load would pin two VGPRs for the
  // whole kernel; the fused ICP kernel is register-bound)
  const int64_t raw = *c.dev;
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)raw);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)raw >> 32));
  const int64_t v = (int64_t)(((uint64_t)hi << 32) | lo);
  return v < c.host ? (v < 0 ? 0 : v) : c.host;
}

// ---------------------------------------------------------------- block primitives -----
constexpr int GS_WAVE = 64;

GS_DEV int gs_wave_incl_scan(int v) {
  const int lane = threadIdx.x & (GS_WAVE - 1);
#pragma unroll
  for (int d = 1; d < GS_WAVE; d <<= 1) {
    int t = __shfl_up(v, d, GS_WAVE);
    if (lane >= d) v += t;
  }
  return v;
}

// Exclusive scan of one int per thread over a block of BLOCK threads (BLOCK % 64 == 0).
// Returns the exclusive prefix; *total receives the block sum.  `smem` has BLOCK/64 + 1 ints.
template <int BLOCK>
GS_DEV int gs_block_excl_scan(int v, int* smem, int* total) {
  constexpr int NW = BLOCK / GS_WAVE;
  const int lane = threadIdx.x & (GS_WAVE - 1);
  const int wave = threadIdx.x / GS_WAVE;
  int incl = gs_wave_incl_scan(v);
  if (lane == GS_WAVE - 1) smem[wave] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int w = 0; w < NW; ++w) {
      int c = smem[w];
      smem[w] = run;
      run += c;
    }
    smem[NW] = run;
  }
  __syncthreads();
  int base = smem[wave];
  *total = smem[NW];
  __syncthreads();
  return base + incl - v;
}

GS_DEV double gs_wave_sum_f64(double v) {
#pragma unroll
  for (int d = GS_WAVE / 2; d > 0; d >>= 1) v += __shfl_down(v, d, GS_WAVE);
  return v;
}
