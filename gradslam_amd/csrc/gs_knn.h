// gs_knn.h — internal interface of the exact 1-NN engine (gs_knn.hip), shared with gs_icp.hip.
#pragma once
#include "gs_common.h"

// best[s] = (float_bits(d2) << 32) | target_index, the minimum over all targets; lowest index on
// ties.  Callers arm best[] with all ones (memset 0xff) before the first search; the consumers
// re-arm it (see gs_icp_linearize_kernel).
GS_DEV unsigned long long knn_pack(float d, uint32_t idx) {
  return ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)idx;
}

// Brute force over every (src, tgt) pair.  Tapply (device, 12+ floats, may be NULL) is applied to
// every source point on load; src_out (may be NULL) receives the transformed points.
int gs_knn_brute_launch(const float* src_in, const float* Tapply, float* src_out, int64_t n_src,
                        const float* tgt, int64_t n_tgt, unsigned long long* best, hipStream_t st);

// Uniform-grid engine: build once per target set, query many times.  Results are IDENTICAL to
// the brute-force search (same distances, same tie-break); queries the grid cannot resolve
// within GS_GRID_RINGS shells fall back to a brute-force pass inside gs_knn_grid_query.
constexpr int GS_GRID_MAXCELL = 1 << 20;
constexpr int GS_GRID_RINGS = 3;
struct GsGridScratch {
  void* base;
  int64_t n_src, n_tgt;
};
size_t gs_knn_grid_scratch_bytes(int64_t n_src, int64_t n_tgt);
int gs_knn_grid_build(const float* tgt, int64_t n_tgt, int64_t n_src, void* grid_scratch, hipStream_t st);
int gs_knn_grid_query(const float* src_in, const float* Tapply, float* src_out, int64_t n_src,
                      const float* tgt, int64_t n_tgt, unsigned long long* best, void* grid_scratch,
                      hipStream_t st);
// Heuristic used by the ICP loop: the grid pays off once the target set is large.
static inline bool gs_knn_use_grid(int64_t n_src, int64_t n_tgt) { return n_tgt >= 2048 && n_src >= 256; }
