// chain_probe.hip -- how fast can a chain of DEPENDENT launches go when the dependency is carried by an arrival counter in
// memory instead of the kernel boundary?  (round 6: the ICP solve is 40 such links.)
//   mode 0: one stream, ordinary launches (barrier bit + cache actions at every boundary), plain loads
//   mode 1: one stream, hipExtAnyOrderLaunch (no barrier bit), links wait on the previous link's counter
//   mode 2: two streams, link h on stream h & 1, links wait on the previous link's counter
// Each link: 8 "sequences" (block b serves sequence b % 8), NB blocks of 768 threads per sequence; a block waits until the
// NB blocks of its sequence have arrived at the previous link, sums their NB rows of 28 doubles (L1-bypassing loads), does
// `work` dependent FMAs, stores its own row, drains its stores, arrives.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int NV = 28, TPB = 768;
__device__ inline unsigned ld_u32_sc1(const unsigned* p) {
  unsigned v;
  asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ inline unsigned ld_u32_sys(const unsigned* p) {
  unsigned v;
  asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__global__ void __launch_bounds__(TPB) link(double* rows, unsigned* counters, unsigned* err, int h, int nb, int nseq, int wait, int scope, int work) {
  const int seq = blockIdx.x % nseq, lb = blockIdx.x / nseq;
  __shared__ double S[NV];
  __shared__ int abort_s;
  const double* in = rows + ((size_t)((h + 1) & 1) * nseq + seq) * nb * NV;
  double* out = rows + ((size_t)(h & 1) * nseq + seq) * nb * NV;
  if (threadIdx.x == 0) abort_s = 0;
  if (wait && h > 0 && threadIdx.x == 0) {
    const unsigned* c = counters + (size_t)(h - 1) * 8 + seq;
    unsigned n = 0;
    while ((scope ? ld_u32_sys(c) : ld_u32_sc1(c)) < (unsigned)nb) {
      if (++n > (1u << 20)) { abort_s = 1; atomicExch(err, 1u + h); break; }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  if (abort_s) return;
  {
    __shared__ double part_s[24][NV];
    const int col = threadIdx.x % NV, part = threadIdx.x / NV;
    if (part < 24) {
      double s = 0.0;
      if (h > 0)
        for (int b = part; b < nb; b += 24) {
          const double* p = in + (size_t)b * NV + col;
          double v;
          if (!wait) v = *p;
          else if (scope) v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          else v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          s += v;
        }
      part_s[part][col] = s;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
      double t = 0.0;
      for (int k = 0; k < 24; ++k) t += part_s[k][threadIdx.x];
      S[threadIdx.x] = t;
    }
  }
  __syncthreads();
  double x = S[threadIdx.x % NV] * 1e-3 + 1.0;
  for (int i = 0; i < work; ++i) x = __builtin_fma(x, 0.999, 0.001);
  if (threadIdx.x < NV) {
    if (wait && scope) __hip_atomic_store(out + (size_t)lb * NV + threadIdx.x, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else out[(size_t)lb * NV + threadIdx.x] = x;
  }
  if (wait) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned* c = counters + (size_t)h * 8 + seq;
      if (scope) __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      else { unsigned one = 1u; asm volatile("global_atomic_add %0, %1, off" : : "v"(c), "v"(one) : "memory"); }
    }
  }
}
int main(int argc, char** argv) {
  const int nb = argc > 1 ? atoi(argv[1]) : 50, nseq = argc > 2 ? atoi(argv[2]) : 8, links = argc > 3 ? atoi(argv[3]) : 40, work = argc > 4 ? atoi(argv[4]) : 200;
  double* rows; unsigned* counters; unsigned* err;
  CK(hipMalloc(&rows, sizeof(double) * 2 * 8 * 4096 * NV));
  CK(hipMalloc(&counters, sizeof(unsigned) * 8 * 4096));
  CK(hipMalloc(&err, 4));
  hipStream_t st[2];
  CK(hipStreamCreate(&st[0])); CK(hipStreamCreate(&st[1]));
  hipEvent_t e0, e1, ej;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
  for (int scope = 0; scope < 2; ++scope)
  for (int mode = 0; mode < 3; ++mode) {
    if (mode == 0 && scope == 1) continue;
    float best = 1e30f, sum = 0; unsigned herr = 0;
    const int reps = 12;
    for (int r = 0; r < reps; ++r) {
      CK(hipMemsetAsync(counters, 0, sizeof(unsigned) * 8 * 4096, st[0]));
      CK(hipMemsetAsync(err, 0, 4, st[0]));
      CK(hipStreamSynchronize(st[0]));
      CK(hipEventRecord(e0, st[0]));
      if (mode == 2) { CK(hipEventRecord(ej, st[0])); CK(hipStreamWaitEvent(st[1], ej, 0)); }
      for (int h = 0; h < links; ++h) {
        hipStream_t s = mode == 2 ? st[h & 1] : st[0];
        if (mode == 1) hipExtLaunchKernelGGL(link, dim3(nb * nseq), dim3(TPB), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, rows, counters, err, h, nb, nseq, 1, scope, work);
        else hipLaunchKernelGGL(link, dim3(nb * nseq), dim3(TPB), 0, s, rows, counters, err, h, nb, nseq, mode != 0, scope, work);
      }
      if (mode == 2) { CK(hipEventRecord(ej, st[1])); CK(hipStreamWaitEvent(st[0], ej, 0)); }
      CK(hipEventRecord(e1, st[0]));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 2) { sum += ms; if (ms < best) best = ms; }
      CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
      if (herr) break;
    }
    double chk; CK(hipMemcpy(&chk, rows + ((size_t)((links - 1) & 1) * nseq) * nb * NV, 8, hipMemcpyDeviceToHost));
    printf("nb %d nseq %d links %d work %d scope %s mode %d (%s): %.2f us per link (best %.2f) err %u check %.6f\n", nb, nseq, links, work, scope ? "system" : "xcd-l2", mode,
           mode == 0 ? "barrier launches" : mode == 1 ? "any-order launches + counters" : "two streams + counters", sum / (reps - 2) / links * 1e3, best / links * 1e3, herr, chk);
  }
  return 0;
}
