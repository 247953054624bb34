// gs_ingest.hip — dataset -> device ingest: raw sensor frames (uint16 depth, uint8 colour, as decoded
// from the PNGs) to the float32 images RGBDImages consumes, with the resize the reference's loaders
// apply on the host through OpenCV (datasets/tum.py:448-477, datasets/icl.py, datasets/scannet.py:
// depth cv2.INTER_NEAREST then / scaling_factor; colour cv2.INTER_LINEAR then optional / 255).
// The reference does this arithmetic in float64 and casts to float32 last; so do these kernels.
// HBM-bound: 2 B read + 4 B written per depth pixel, <= 12 B read + 12 B written per colour pixel.
#include "gs_common.h"

// cv2.INTER_NEAREST (resizeNN): source index = min(floor(dst * (1 / (dst_size / src_size))), src_size - 1)
GS_DEV int ingest_nn(int d, double inv_scale, int n_src) {
  const int s = (int)floor((double)d * inv_scale);
  return s < n_src - 1 ? s : n_src - 1;
}

__global__ void __launch_bounds__(256) gs_ingest_depth_kernel(const uint16_t* __restrict__ raw, int H0, int W0,
                                                              float* __restrict__ out, int H, int W,
                                                              double ify, double ifx, double scale_div) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= (int64_t)H * W) return;
  const int y = (int)(p / W), x = (int)(p % W);
  const int sy = (H == H0) ? y : ingest_nn(y, ify, H0), sx = (W == W0) ? x : ingest_nn(x, ifx, W0);
  out[p] = (float)((double)raw[(int64_t)sy * W0 + sx] / scale_div);
}

extern "C" int gs_ingest_depth_u16_f32(const uint16_t* raw, int H0, int W0, float* out, int H, int W,
                                       double scale_div, void* stream) {
  GS_REQUIRE(raw && out && H0 > 0 && W0 > 0 && H > 0 && W > 0 && scale_div != 0.0, "bad arguments");
  const double ify = 1.0 / ((double)H / (double)H0), ifx = 1.0 / ((double)W / (double)W0);
  hipLaunchKernelGGL(gs_ingest_depth_kernel, dim3((unsigned)gs_ceil_div((int64_t)H * W, 256)), dim3(256), 0,
                     gs_stream(stream), raw, H0, W0, out, H, W, ify, ifx, scale_div);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

// cv2.INTER_LINEAR on a float64 image (resizeGeneric_, HResizeLinear<double,double,float>, VResizeLinear):
// f = (float)((d + 0.5) * scale - 0.5); s = floor(f); f -= s; clamped at both borders with weight 0;
// the two weights are float32 (1.f - f, f), the sums double: first along x for the two source rows,
// then along y.
GS_DEV void ingest_lin(int d, double scale, int n_src, int& s0, int& s1, float& a0, float& a1) {
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (s < 0) { f = 0.0f; s = 0; }
  if (s >= n_src - 1) { f = 0.0f; s = n_src - 1; }
  s0 = s;
  s1 = s + 1 < n_src ? s + 1 : n_src - 1;
  a0 = 1.0f - f;
  a1 = f;
}

__global__ void __launch_bounds__(256) gs_ingest_color_kernel(const uint8_t* __restrict__ raw, int H0, int W0,
                                                              float* __restrict__ out, int H, int W, double sy_scale,
                                                              double sx_scale, int normalize) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= (int64_t)H * W) return;
  const int y = (int)(p / W), x = (int)(p % W);
  double v[3];
  if (H == H0 && W == W0) {  // cv2.resize to the same size is a copy
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = (double)raw[3 * p + c];
  } else {
    int x0, x1, y0, y1;
    float a0, a1, b0, b1;
    ingest_lin(x, sx_scale, W0, x0, x1, a0, a1);
    ingest_lin(y, sy_scale, H0, y0, y1, b0, b1);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double r0 = (double)raw[3 * ((int64_t)y0 * W0 + x0) + c] * (double)a0 +
                        (double)raw[3 * ((int64_t)y0 * W0 + x1) + c] * (double)a1;
      const double r1 = (double)raw[3 * ((int64_t)y1 * W0 + x0) + c] * (double)a0 +
                        (double)raw[3 * ((int64_t)y1 * W0 + x1) + c] * (double)a1;
      v[c] = r0 * (double)b0 + r1 * (double)b1;
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) out[3 * p + c] = (float)(normalize ? v[c] / 255.0 : v[c]);
}

extern "C" int gs_ingest_color_u8_f32(const uint8_t* raw, int H0, int W0, float* out, int H, int W, int normalize,
                                      void* stream) {
  GS_REQUIRE(raw && out && H0 > 0 && W0 > 0 && H > 0 && W > 0, "bad arguments");
  const double sy = 1.0 / ((double)H / (double)H0), sx = 1.0 / ((double)W / (double)W0);
  hipLaunchKernelGGL(gs_ingest_color_kernel, dim3((unsigned)gs_ceil_div((int64_t)H * W, 256)), dim3(256), 0,
                     gs_stream(stream), raw, H0, W0, out, H, W, sy, sx, normalize);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

// ---- streaming ingest (round 4): n native-size frames (depth and colour) in ONE launch, the arithmetic of the two
// kernels above for H == H0, W == W0.  A thread converts 4 consecutive pixels: 8 B + 12 B read, 16 B + 48 B written, every
// access 8 / 16 bytes wide.  For raw frames that arrive over PCIe while the previous step computes
// (gradslam_amd/datasets/streaming.py).
constexpr int GS_INGEST_BLOCKS = 128;   // 512 waves on 1024 SIMDs, ~400 KB of loads in flight per sweep (PCIe needs ~100 KB)
// One work item = 4 depth pixels (8 B -> 16 B) or 4 colour BYTES (4 B -> 16 B): items [0, n_quads) are depth quads, items
// [n_quads, 4 n_quads) colour words.  A SMALL grid walks them: the launch overlaps the ICP chain of the previous frame,
// whose blocks need every SIMD's register file almost entirely (6 waves x 80 VGPRs of 512); a full-size grid of
// fat waves parked on PCIe loads took the second block's place on their CUs and cost the chain 0.17 ms per step.
__global__ void __launch_bounds__(256) gs_ingest_native4_kernel(const uint16_t* __restrict__ depth_raw,
                                                                const uint8_t* __restrict__ color_raw, int64_t n_quads,
                                                                float* __restrict__ depth_out, float* __restrict__ color_out,
                                                                double scale_div, int normalize) {
  const int64_t first = depth_raw ? 0 : n_quads, last = color_raw ? 4 * n_quads : n_quads;
  for (int64_t i = first + (int64_t)blockIdx.x * 256 + threadIdx.x; i < last; i += (int64_t)gridDim.x * 256) {
    float4 o;
    if (i < n_quads) {
      const uint2 r = reinterpret_cast<const uint2*>(depth_raw)[i];
      o.x = (float)((double)(r.x & 0xffffu) / scale_div);
      o.y = (float)((double)(r.x >> 16) / scale_div);
      o.z = (float)((double)(r.y & 0xffffu) / scale_div);
      o.w = (float)((double)(r.y >> 16) / scale_div);
      reinterpret_cast<float4*>(depth_out)[i] = o;
    } else {
      const int64_t w = i - n_quads;
      const uint32_t c = reinterpret_cast<const uint32_t*>(color_raw)[w];
      const double b0 = (double)(c & 0xffu), b1 = (double)((c >> 8) & 0xffu), b2 = (double)((c >> 16) & 0xffu), b3 = (double)(c >> 24);
      o.x = (float)(normalize ? b0 / 255.0 : b0);
      o.y = (float)(normalize ? b1 / 255.0 : b1);
      o.z = (float)(normalize ? b2 / 255.0 : b2);
      o.w = (float)(normalize ? b3 / 255.0 : b3);
      reinterpret_cast<float4*>(color_out)[w] = o;
    }
  }
}

extern "C" int gs_ingest_frames_native_f32(const uint16_t* depth_raw, const uint8_t* color_raw, int64_t n_frames, int H,
                                           int W, double scale_div, int normalize, float* depth_out, float* color_out,
                                           void* stream) {
  GS_REQUIRE(n_frames > 0 && H > 0 && W > 0 && scale_div != 0.0, "bad arguments");
  GS_REQUIRE((depth_raw == nullptr) == (depth_out == nullptr) && (color_raw == nullptr) == (color_out == nullptr) &&
                 (depth_raw || color_raw), "raw / out pointers must come in pairs");
  const int64_t px = n_frames * (int64_t)H * W;
  GS_REQUIRE(px % 4 == 0, "the pixel count must be a multiple of 4 (use the per-frame entry points otherwise)");
  GS_REQUIRE(((uintptr_t)depth_raw % 8 == 0) && ((uintptr_t)color_raw % 4 == 0) && ((uintptr_t)depth_out % 16 == 0) &&
                 ((uintptr_t)color_out % 16 == 0), "misaligned buffers");
  const int64_t nb = gs_ceil_div(px, 256);   // (one work item per depth quad + three per colour quad)
  hipLaunchKernelGGL(gs_ingest_native4_kernel, dim3((unsigned)(nb < GS_INGEST_BLOCKS ? nb : GS_INGEST_BLOCKS)), dim3(256), 0, gs_stream(stream),
                     depth_raw, color_raw, px / 4, depth_out, color_out, scale_div, normalize);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

// Device-side address of pinned (hipHostMalloc / hipHostRegister) host memory, or an error when the range is not mapped
// into the device's address space.  The streaming ingest reads the raw frames straight out of pinned host memory with it
// (no hipMemcpyAsync: measured, two asynchronous 6 MB copies cost the enqueuing host thread 0.24 ms).
extern "C" int gs_host_device_pointer(const void* host_ptr, void** dev_ptr_out) {
  GS_REQUIRE(host_ptr && dev_ptr_out, "bad arguments");
  void* d = nullptr;
  const hipError_t e = hipHostGetDevicePointer(&d, const_cast<void*>(host_ptr), 0);
  if (e != hipSuccess || !d) {
    (void)hipGetLastError();
    gs_set_error("gs_host_device_pointer: not a device-mapped pinned allocation (%s)", hipGetErrorString(e));
    return GS_ERR_INVALID;
  }
  *dev_ptr_out = d;
  return GS_OK;
}
