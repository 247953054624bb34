// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of the map passes:
// the guide calibrates FETCH_SIZE (x2) for 16-byte-per-lane streams only; the map passes move 12-byte rows (dwordx3), 4-byte
// columns and 16-byte gathers, and their working set (~230 MB for 8 maps) sits mostly in the 256 MiB Infinity Cache.
// Every kernel moves a KNOWN number of bytes; run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate
// passes) and compare (tools/calib/run.sh, tools/calib/summarize.py).
//   hipcc --offload-arch=gfx950 -O3 -o tools/calib/counter_calibration tools/calib/counter_calibration.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// rows of 12 bytes, one row per lane and trip (the layout of points / normals / colours)
__global__ void __launch_bounds__(256) read_rows12(const float* __restrict__ a, int64_t n, float* __restrict__ out) {
  float acc = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    acc += a[3 * i] + a[3 * i + 1] + a[3 * i + 2];
  if (acc == 1.2345e-30f) out[0] = acc;
}
__global__ void __launch_bounds__(256) write_rows12(float* __restrict__ a, int64_t n, float v) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    a[3 * i] = v; a[3 * i + 1] = v; a[3 * i + 2] = v;
  }
}
// the 16-byte stream the guide is calibrated on
__global__ void __launch_bounds__(256) read_rows16(const float4* __restrict__ a, int64_t n, float* __restrict__ out) {
  float acc = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float4 v = a[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 1.2345e-30f) out[0] = acc;
}
__global__ void __launch_bounds__(256) write_rows16(float4* __restrict__ a, int64_t n, float v) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) a[i] = make_float4(v, v, v, v);
}
// 4-byte column (confidence counts, pix[], alpha)
__global__ void __launch_bounds__(256) read_col4(const float* __restrict__ a, int64_t n, float* __restrict__ out) {
  float acc = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc += a[i];
  if (acc == 1.2345e-30f) out[0] = acc;
}
// the merge: read-modify-write of three 12-byte rows and a 4-byte column (80 bytes per row)
__global__ void __launch_bounds__(256) rmw_rows40(float* __restrict__ p, float* __restrict__ q, float* __restrict__ c, float* __restrict__ w, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float cc = w[i], inv = 1.0f / (cc == 0.0f ? 1.0f : cc);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      p[3 * i + k] = (cc * p[3 * i + k]) * inv;
      q[3 * i + k] = (cc * q[3 * i + k]) * inv;
      c[3 * i + k] = (cc * c[3 * i + k]) * inv;
    }
    w[i] = cc;
  }
}
// gathers: lane i reads the 12-byte row idx[i] (the frame-side operands of the association: pixels in data-dependent order)
__global__ void __launch_bounds__(256) gather_rows12(const float* __restrict__ a, const int32_t* __restrict__ idx, int64_t n, float* __restrict__ out) {
  float acc = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t j = idx[i];
    acc += a[3 * j] + a[3 * j + 1] + a[3 * j + 2];
  }
  if (acc == 1.2345e-30f) out[0] = acc;
}

int main(int argc, char** argv) {
  // rows: 230 MB of 12-byte rows ~ the 8 maps' points of the benchmark (inside the Infinity Cache); 1.5 GB beyond it
  const int64_t rows_small = 19000000, rows_big = 125000000;
  const int reps = argc > 1 ? atoi(argv[1]) : 3;
  float *a, *b, *c, *w, *out;
  int32_t* idx;
  CK(hipMalloc(&a, 12 * rows_big + 64)); CK(hipMalloc(&b, 12 * rows_small)); CK(hipMalloc(&c, 12 * rows_small)); CK(hipMalloc(&w, 4 * rows_big));
  CK(hipMalloc(&out, 256)); CK(hipMalloc(&idx, 4 * rows_small));
  CK(hipMemset(a, 0, 12 * rows_big)); CK(hipMemset(b, 0, 12 * rows_small)); CK(hipMemset(c, 0, 12 * rows_small)); CK(hipMemset(w, 0, 4 * rows_big));
  // gather indices: a pixel image of 8 x 307 200 rows visited in a pseudo-random order with locality (runs of 8 neighbours)
  {
    int32_t* h = (int32_t*)malloc(4 * rows_small);
    const int64_t P = 8 * 307200;
    uint64_t s = 88172645463325252ull;
    for (int64_t i = 0; i < rows_small; i += 8) {
      s ^= s << 13; s ^= s >> 7; s ^= s << 17;
      const int64_t base = (int64_t)(s % (uint64_t)(P - 8));
      for (int k = 0; k < 8 && i + k < rows_small; ++k) h[i + k] = (int32_t)(base + k);
    }
    CK(hipMemcpy(idx, h, 4 * rows_small, hipMemcpyHostToDevice));
    free(h);
  }
  const dim3 g(256 * 8), t(256);
  for (int r = 0; r < reps; ++r) {
    hipLaunchKernelGGL(read_rows12, g, t, 0, 0, a, rows_small, out);
    hipLaunchKernelGGL(read_rows12, g, t, 0, 0, a, rows_big, out);
    hipLaunchKernelGGL(write_rows12, g, t, 0, 0, a, rows_small, 1.0f);
    hipLaunchKernelGGL(write_rows12, g, t, 0, 0, a, rows_big, 1.0f);
    hipLaunchKernelGGL(read_rows16, g, t, 0, 0, (const float4*)a, rows_small * 3 / 4, out);
    hipLaunchKernelGGL(read_rows16, g, t, 0, 0, (const float4*)a, rows_big * 3 / 4, out);
    hipLaunchKernelGGL(write_rows16, g, t, 0, 0, (float4*)a, rows_small * 3 / 4, 1.0f);
    hipLaunchKernelGGL(write_rows16, g, t, 0, 0, (float4*)a, rows_big * 3 / 4, 1.0f);
    hipLaunchKernelGGL(read_col4, g, t, 0, 0, w, rows_small, out);
    hipLaunchKernelGGL(read_col4, g, t, 0, 0, w, rows_big, out);
    hipLaunchKernelGGL(rmw_rows40, g, t, 0, 0, a, b, c, w, rows_small / 3);
    hipLaunchKernelGGL(gather_rows12, g, t, 0, 0, a, idx, rows_small, out);
  }
  CK(hipDeviceSynchronize());
  // the true bytes of every launch, in launch order (one line per launch of a repetition)
  printf("# kernel rows true_read_bytes true_write_bytes\n");
  printf("read_rows12 %lld %lld 0\n", (long long)rows_small, (long long)(12 * rows_small));
  printf("read_rows12 %lld %lld 0\n", (long long)rows_big, (long long)(12 * rows_big));
  printf("write_rows12 %lld 0 %lld\n", (long long)rows_small, (long long)(12 * rows_small));
  printf("write_rows12 %lld 0 %lld\n", (long long)rows_big, (long long)(12 * rows_big));
  printf("read_rows16 %lld %lld 0\n", (long long)(rows_small * 3 / 4), (long long)(16 * (rows_small * 3 / 4)));
  printf("read_rows16 %lld %lld 0\n", (long long)(rows_big * 3 / 4), (long long)(16 * (rows_big * 3 / 4)));
  printf("write_rows16 %lld 0 %lld\n", (long long)(rows_small * 3 / 4), (long long)(16 * (rows_small * 3 / 4)));
  printf("write_rows16 %lld 0 %lld\n", (long long)(rows_big * 3 / 4), (long long)(16 * (rows_big * 3 / 4)));
  printf("read_col4 %lld %lld 0\n", (long long)rows_small, (long long)(4 * rows_small));
  printf("read_col4 %lld %lld 0\n", (long long)rows_big, (long long)(4 * rows_big));
  printf("rmw_rows40 %lld %lld %lld\n", (long long)(rows_small / 3), (long long)(40 * (rows_small / 3)), (long long)(40 * (rows_small / 3)));
  printf("gather_rows12 %lld %lld 0\n", (long long)rows_small, (long long)(16 * rows_small));
  return 0;
}
