// gs_knn_bbox.h -- first pass of a grid build (bounding box of the targets, optional projection of the map rows and
// compaction of the rows that pass the target filter) as a device function: gs_knn.hip wraps it in its kernels, the
// batched localisation (gs_icp_loop.hip) runs it in the launch that also writes the ICP source lattice.
#pragma once
#include "gs_assoc_dev.h"
#include "gs_knn.h"

// Bounding box of the finite targets: block-local min / max, then 6 atomicMax on order-preserving
// codes (min and max are order-independent, so the result is deterministic).  code(v) grows with v;
// the lower corner is kept as max(~code): zero-initialised words mean "nothing seen yet".
GS_DEV unsigned grid_code(float v) {
  const unsigned b = __float_as_uint(v);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
GS_DEV float grid_decode(unsigned c) {
  return __uint_as_float((c & 0x80000000u) ? (c & 0x7fffffffu) : ~c);
}
constexpr int GB_BLOCK = 256;
constexpr int GB_ITEMS = 8;
// Body shared by the single-sequence and the batched kernels (blk = block index within the sequence).
// cam != NULL: pix[] is an OUTPUT (projection of every row under the camera, gs_project_map_f32) and the
// filter is evaluated on the value just computed.
GS_DEV void grid_bbox_body(const float* __restrict__ tgt, const int64_t n_tgt, const GsTargetFilter flt,
                           const GsCamera* cam, int H, float u_hi, float v_hi, int32_t* __restrict__ pix_out,
                           unsigned* __restrict__ bbox, int* __restrict__ unres_count, float4* __restrict__ tlist,
                           const unsigned blk) {
  if (blk == 0 && threadIdx.x == 0) { unres_count[0] = 0; unres_count[1] = 0; }
  if ((int64_t)blk * GB_BLOCK * GB_ITEMS >= n_tgt) return;
  __shared__ float red[6][GB_BLOCK / GS_WAVE];
  __shared__ int scan_s[GB_BLOCK / GS_WAVE + 1];
  __shared__ unsigned base_s;
  int hits = 0;
  unsigned hitmask = 0;  // bit u: item u of this thread passed the filter
  float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  // Loads of all GB_ITEMS rows first (unconditional, on a clamped index), then the arithmetic: with the load inside
  // the per-row control flow the compiler serialises them (load, wait, project, store, next load: 8 dependent round
  // trips per thread).
  const int64_t i0 = (int64_t)blk * GB_ITEMS * GB_BLOCK + threadIdx.x;
  if (cam) {
    float v[GB_ITEMS][3];
#pragma unroll
    for (int u = 0; u < GB_ITEMS; ++u) {
      const int64_t i = i0 + (int64_t)u * GB_BLOCK, ic = i < n_tgt ? i : n_tgt - 1;
      v[u][0] = tgt[3 * ic]; v[u][1] = tgt[3 * ic + 1]; v[u][2] = tgt[3 * ic + 2];
    }
#pragma unroll
    for (int u = 0; u < GB_ITEMS; ++u) {
      const int64_t i = i0 + (int64_t)u * GB_BLOCK;
      if (i < n_tgt) {
        int ph = 0, pw = 0;
        const bool in = gs_project_point_hw(*cam, v[u][0], v[u][1], v[u][2], H, flt.W, u_hi, v_hi, ph, pw);
        pix_out[i] = in ? (int32_t)(ph * flt.W + pw) : -1;
        // lattice test on (h, w) directly: dividing the flat index by run-time W and ds again was most of this pass's
        // instructions, and the pass is VALU-bound (SQ counters: 77 VALU instructions per row at 4 cycles per wave64)
        if (in && gs_on_lattice(ph, pw, flt.ds)) {
          ++hits;
          hitmask |= 1u << u;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const float w = v[u][k];
            if (w > -3.0e38f && w < 3.0e38f) {  // finite
              lo[k] = w < lo[k] ? w : lo[k];
              hi[k] = w > hi[k] ? w : hi[k];
            }
          }
        }
      }
    }
  } else {
    bool t[GB_ITEMS];
    if (flt.pix) {
      int32_t px[GB_ITEMS];
#pragma unroll
      for (int u = 0; u < GB_ITEMS; ++u) {
        const int64_t i = i0 + (int64_t)u * GB_BLOCK;
        px[u] = flt.pix[i < n_tgt ? i : n_tgt - 1];
      }
#pragma unroll
      for (int u = 0; u < GB_ITEMS; ++u) {
        const int32_t p = px[u];
        t[u] = (i0 + (int64_t)u * GB_BLOCK < n_tgt) && p >= 0 && ((p / flt.W) % flt.ds == 0) && ((p % flt.W) % flt.ds == 0);
      }
    } else {
#pragma unroll
      for (int u = 0; u < GB_ITEMS; ++u) t[u] = i0 + (int64_t)u * GB_BLOCK < n_tgt;
    }
#pragma unroll
    for (int u = 0; u < GB_ITEMS; ++u) {
      if (t[u]) {
        const int64_t i = i0 + (int64_t)u * GB_BLOCK;
        ++hits;
        hitmask |= 1u << u;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float w = tgt[3 * i + k];
          if (w > -3.0e38f && w < 3.0e38f) {  // finite
            lo[k] = w < lo[k] ? w : lo[k];
            hi[k] = w > hi[k] ? w : hi[k];
          }
        }
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float a = lo[k], b = hi[k];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      const float a2 = __shfl_down(a, d, GS_WAVE), b2 = __shfl_down(b, d, GS_WAVE);
      a = a2 < a ? a2 : a;
      b = b2 > b ? b2 : b;
    }
    if (lane == 0) { red[k][wave] = a; red[3 + k][wave] = b; }
  }
  const bool filtered = flt.pix != nullptr || cam != nullptr;
  __syncthreads();
  if (filtered) {
    // the rows that passed the filter (a few per cent of a map) are compacted into tlist: the count and scatter
    // passes then walk that list instead of the map.  One atomic per block hands out the slots (bbox[6] is also the
    // number of targets the cell-size heuristic needs); the order of the list does not matter (see the scatter).
    int total;
    int pos = gs_block_excl_scan<GB_BLOCK>(hits, scan_s, &total);
    if (threadIdx.x == 0 && total) base_s = atomicAdd(&bbox[6], (unsigned)total);
    __syncthreads();
    if (hitmask) {
      const unsigned base = base_s;
#pragma unroll
      for (int u = 0; u < GB_ITEMS; ++u) {
        if (hitmask & (1u << u)) {
          const int64_t i = ((int64_t)blk * GB_ITEMS + u) * GB_BLOCK + threadIdx.x;
          tlist[base + (unsigned)pos] = make_float4(tgt[3 * i], tgt[3 * i + 1], tgt[3 * i + 2], __int_as_float((int)i));
          ++pos;
        }
      }
    }
  }
  if (threadIdx.x < 3) {
    const int k = threadIdx.x;
    float a = red[k][0], b = red[3 + k][0];
    for (int w = 1; w < GB_BLOCK / GS_WAVE; ++w) {
      a = red[k][w] < a ? red[k][w] : a;
      b = red[3 + k][w] > b ? red[3 + k][w] : b;
    }
    if (a <= b) {  // at least one finite coordinate on this axis in this block
      atomicMax(&bbox[k], ~grid_code(a));
      atomicMax(&bbox[3 + k], grid_code(b));
    }
  }
}

// ---- batched build: block `bid` of B * ceil(n_max / (GB_BLOCK * GB_ITEMS)) works for sequence bid % B ----
// The camera is derived by every thread from wave-uniform loads (scalar loads; the 24 coefficients then are scalar
// operands of the projection arithmetic: the pass is VALU-bound, and reading them from LDS per row cost issue slots).
GS_DEV void gridb_bbox_block(const GsGridBatch& gb, const unsigned bid, float u_hi, float v_hi) {
  const GsGridSeq& q = gb.s[bid % gb.B];
  const unsigned blk = bid / gb.B;
  const GsTargetFilter flt{q.pix, gb.W, gb.ds};
  if (q.pose16) {
    const GsCamera cam = gs_camera(q.pose16, q.K16);
    grid_bbox_body(q.tgt, gs_count(q.n_tgt), flt, &cam, gb.H, u_hi, v_hi, q.pix, q.m.bbox, q.m.unres_count, q.m.tlist,
                   blk);
  } else {
    grid_bbox_body(q.tgt, gs_count(q.n_tgt), flt, nullptr, 0, 0.0f, 0.0f, nullptr, q.m.bbox, q.m.unres_count, q.m.tlist,
                   blk);
  }
}
static inline unsigned gs_knn_gridb_bbox_blocks(const GsGridBatch& gb) {
  int64_t n_max = 1;
  for (int b = 0; b < gb.B; ++b) n_max = gb.s[b].n_tgt.host > n_max ? gb.s[b].n_tgt.host : n_max;
  return (unsigned)gb.B * (unsigned)gs_ceil_div(n_max, GB_BLOCK * GB_ITEMS);
}
