// gs_knn.hip — K3: exact 1-nearest-neighbour search (replaces chamferdist.knn_points as called at
// odometry/icputils.py:200; squared L2 d = fma(dz,dz,fma(dy,dy,dx*dx)), lowest index on ties).
//
// Two engines with bit-identical results:
//
//  * brute force: fp32-VALU bound (Ns x Nt pair distances, 6 VALU ops + 3 for the running arg-min
//    per pair).  A block stages its target chunk in LDS and reads it with wave-uniform (broadcast)
//    ds_read_b128; every lane keeps KNN_SPT source points in registers.  (src tile, tgt chunk)
//    pairs are spread over a 2-D grid; chunk results meet in a 64-bit atomicMin on
//    (distance bits << 32 | index): order independent, ties resolve to the lowest index exactly
//    like a sequential scan.
//
//  * uniform grid (used by the ICP loop, where the SAME target set is searched 40 times per
//    frame): targets are counting-sorted into cells of a 3-D grid once; a query visits Chebyshev
//    shells of cells around its own cell and stops as soon as the best distance found is provably
//    smaller than anything an unvisited cell can hold.  Every candidate goes through the same
//    distance arithmetic and the same (distance, index) ordering as the brute-force scan, so the
//    result is the brute-force result, not an approximation.  Queries that are not resolved
//    within GS_GRID_RINGS shells (points far from every target) are collected and finished by the
//    brute-force kernel.  Work drops from Ns*Nt pair distances to ~10^2 per query.
#include "gs_knn.h"

#include <stdlib.h>

#include "gs_assoc_dev.h"
#include "gs_knn_bbox.h"

constexpr int KNN_BLOCK = 256;
constexpr int KNN_TCHUNK = 512;

// ---------------------------------------------------------------- brute force ----------
// LISTED: source indices come from list[0 .. *count) (the grid engine's unresolved queries) and
// source tiles are strided over gridDim.x; otherwise tile = blockIdx.x over all n_src points.
template <bool LISTED, int KNN_SPT>
__global__ void __launch_bounds__(KNN_BLOCK) gs_knn1_kernel(
    const float* __restrict__ src_in, const float* __restrict__ Tapply, float* __restrict__ src_out,
    int64_t n_src, const float* __restrict__ tgt, int64_t n_tgt, unsigned long long* __restrict__ best,
    const int* __restrict__ list, const int* __restrict__ count) {
  constexpr int KNN_STILE = KNN_BLOCK * KNN_SPT;
  __shared__ float4 tl[KNN_TCHUNK];
  const int64_t n_q = LISTED ? (int64_t)(*count) : n_src;
  if (LISTED && n_q == 0) return;
  const int64_t j0 = (int64_t)blockIdx.y * KNN_TCHUNK;
  const int cnt = (int)((n_tgt - j0) < KNN_TCHUNK ? (n_tgt - j0) : KNN_TCHUNK);
  if (cnt <= 0) return;
  for (int i = threadIdx.x; i < cnt; i += KNN_BLOCK) {
    const float* t = tgt + 3 * (j0 + i);
    tl[i] = make_float4(t[0], t[1], t[2], 0.0f);
  }
  float T[12];
  if (Tapply) {
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = Tapply[i];
  }
  __syncthreads();
  for (int64_t tile = blockIdx.x; tile * KNN_STILE < n_q; tile += gridDim.x) {
    float sx[KNN_SPT], sy[KNN_SPT], sz[KNN_SPT], bd[KNN_SPT];
    int bi[KNN_SPT];
    int64_t sidx[KNN_SPT];
#pragma unroll
    for (int k = 0; k < KNN_SPT; ++k) {
      const int64_t q = tile * KNN_STILE + k * KNN_BLOCK + threadIdx.x;
      int64_t s = -1;
      if (q < n_q) s = LISTED ? (int64_t)list[q] : q;
      float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f;
      if (s >= 0) {
        p0 = src_in[3 * s];
        p1 = src_in[3 * s + 1];
        p2 = src_in[3 * s + 2];
        if (Tapply) {
          float q0, q1, q2;
          gs_rigid_fma(T, p0, p1, p2, q0, q1, q2);
          p0 = q0; p1 = q1; p2 = q2;
        }
        if (!LISTED && src_out && blockIdx.y == 0) {
          src_out[3 * s] = p0;
          src_out[3 * s + 1] = p1;
          src_out[3 * s + 2] = p2;
        }
      }
      sidx[k] = s;
      sx[k] = p0; sy[k] = p1; sz[k] = p2;
      bd[k] = __builtin_inff();
      bi[k] = 0;
    }
#pragma unroll 4
    for (int j = 0; j < cnt; ++j) {
      const float4 t = tl[j];
#pragma unroll
      for (int k = 0; k < KNN_SPT; ++k) {
        const float dx = sx[k] - t.x, dy = sy[k] - t.y, dz = sz[k] - t.z;
        float d = dx * dx;
        d = gs_fma(dy, dy, d);
        d = gs_fma(dz, dz, d);
        const bool lt = d < bd[k];
        bd[k] = lt ? d : bd[k];
        bi[k] = lt ? j : bi[k];
      }
    }
#pragma unroll
    for (int k = 0; k < KNN_SPT; ++k)
      if (sidx[k] >= 0) atomicMin(&best[sidx[k]], knn_pack(bd[k], (uint32_t)(j0 + bi[k])));
    if (!LISTED) break;
  }
}

int gs_knn_brute_launch(const float* src_in, const float* Tapply, float* src_out, int64_t n_src,
                        const float* tgt, int64_t n_tgt, unsigned long long* best, hipStream_t st) {
  // source points per thread: 8 amortises the LDS broadcast and loop overhead over more pairs on
  // large problems; 4 keeps more workgroups in flight on small ones (GRADSLAM_HIP_KNN_SPT overrides)
  static int spt_env = -1;
  if (spt_env < 0) {
    const char* e = getenv("GRADSLAM_HIP_KNN_SPT");
    spt_env = e ? atoi(e) : 0;
  }
  const int spt = spt_env == 4 || spt_env == 8 ? spt_env : (n_src * n_tgt >= (int64_t)1 << 27 ? 8 : 4);
  GsProf prof(GS_PROF_KNN, (double)n_src * (double)n_tgt, st);  // work unit: pair distances
  if (spt == 8) {
    dim3 grid((unsigned)gs_ceil_div(n_src, KNN_BLOCK * 8), (unsigned)gs_ceil_div(n_tgt, KNN_TCHUNK));
    hipLaunchKernelGGL((gs_knn1_kernel<false, 8>), grid, dim3(KNN_BLOCK), 0, st, src_in, Tapply, src_out, n_src, tgt,
                       n_tgt, best, nullptr, nullptr);
  } else {
    dim3 grid((unsigned)gs_ceil_div(n_src, KNN_BLOCK * 4), (unsigned)gs_ceil_div(n_tgt, KNN_TCHUNK));
    hipLaunchKernelGGL((gs_knn1_kernel<false, 4>), grid, dim3(KNN_BLOCK), 0, st, src_in, Tapply, src_out, n_src, tgt,
                       n_tgt, best, nullptr, nullptr);
  }
  return GS_OK;
}

// ---------------------------------------------------------------- uniform grid ---------
size_t gs_knn_grid_scratch_bytes(int64_t n_src, int64_t n_tgt) {
  return 512 + 2 * gs_align(4 * (size_t)(GS_GRID_MAXCELL + 1)) + gs_align(4 * (size_t)(GS_GRID_MAXCELL / GS_GRID_TILE + 2)) +
         3 * gs_align(16 * (size_t)(n_tgt > 0 ? n_tgt : 1)) + gs_align(4 * (size_t)(n_src > 0 ? n_src : 1)) + 256;
}

__global__ void __launch_bounds__(GB_BLOCK) gs_grid_bbox_kernel(const float* __restrict__ tgt, GsCount n_tgt_c,
                                                                 const GsTargetFilter flt,
                                                                 unsigned* __restrict__ bbox,
                                                                 int* __restrict__ unres_count,
                                                                 float4* __restrict__ tlist) {
  grid_bbox_body(tgt, gs_count(n_tgt_c), flt, nullptr, 0, 0.0f, 0.0f, nullptr, bbox, unres_count, tlist, blockIdx.x);
}

// Cell size from the bounding box.  Heuristic: targets are a sampled surface (spacing ~ sqrt(area / n))
// or, failing that, a volume (spacing ~ cbrt(V / n)); the cell edge is the larger of 1.5 surface
// spacings and 0.5 volume spacings (3 shells then still reach 1.75 volume spacings), grown until the
// grid fits cells_cap cells.  Any positive cell size is CORRECT; the choice only affects speed.
// Pure function of (bbox, n_tgt): every block that evaluates it gets the same grid.
GS_DEV GsGrid grid_from_bbox(const unsigned* __restrict__ bbox, int64_t n_tgt, int cells_cap) {
  float o[3], m[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const unsigned cl = bbox[k], ch = bbox[3 + k];
    float a = 0.0f, b = 0.0f;  // no finite coordinate on this axis
    if (cl != 0u && ch != 0u) { a = grid_decode(~cl); b = grid_decode(ch); }
    o[k] = a; m[k] = b;
  }
  const float tiny = 1e-6f;
  const float ex = (m[0] - o[0]) + tiny, ey = (m[1] - o[1]) + tiny, ez = (m[2] - o[2]) + tiny;
  const float n = (float)(n_tgt > 0 ? n_tgt : 1);
  const float area = ex * ey + ey * ez + ex * ez;
  float c = 1.5f * sqrtf(area / n);
  const float cv = 0.5f * cbrtf((ex * ey * ez) / n);
  c = cv > c ? cv : c;
  const float emax = ex > ey ? (ex > ez ? ex : ez) : (ey > ez ? ey : ez);
  c = c > emax * (1.0f / 1024.0f) ? c : emax * (1.0f / 1024.0f);
  c = c > 1e-6f ? c : 1e-6f;
  int nx, ny, nz;
  for (;;) {
    nx = (int)(ex / c) + 1; ny = (int)(ey / c) + 1; nz = (int)(ez / c) + 1;
    if ((double)nx * (double)ny * (double)nz <= (double)cells_cap) break;
    c *= 1.26f;
  }
  GsGrid g;
  g.ox = o[0]; g.oy = o[1]; g.oz = o[2];
  g.mx = m[0]; g.my = m[1]; g.mz = m[2];
  g.c = c; g.inv_c = 1.0f / c;
  g.nx = nx; g.ny = ny; g.nz = nz; g.ncell = nx * ny * nz;
  return g;
}

// Every block derives the grid header from the bounding box (block 0 publishes it for the kernels
// that follow), then counts its targets per cell -- and per tile of GS_GRID_TILE cells, which is what the scan pass
// needs of the other tiles (no separate tile-sum launch): block-local histogram in LDS, one global atomic per tile the
// block touched (integer sums: the order does not matter).
GS_DEV void grid_count_body(const float* __restrict__ tgt, const int64_t n_tgt, const GsTargetFilter flt,
                            const unsigned* __restrict__ bbox, GsGrid* __restrict__ gp, int* __restrict__ cell_count,
                            int* __restrict__ tile_sums, int cells_cap, const float4* __restrict__ tlist,
                            const unsigned blk, const unsigned nblk) {
  __shared__ GsGrid gsh;
  __shared__ int th[GS_GRID_MAXCELL / GS_GRID_TILE + 1];
  const int64_t n_items = flt.pix ? (int64_t)bbox[6] : n_tgt;   // filtered build: the compacted list of the bbox pass
  if ((int64_t)blk * 256 >= n_items && blk != 0) return;
  if (threadIdx.x == 0) {
    gsh = grid_from_bbox(bbox, n_items, cells_cap);
    if (blk == 0) *gp = gsh;
  }
  __syncthreads();
  const GsGrid g = gsh;
  const int ntile = g.ncell / GS_GRID_TILE + 1;
  for (int t = threadIdx.x; t < ntile; t += 256) th[t] = 0;
  __syncthreads();
  if (flt.pix) {
    for (int64_t i = (int64_t)blk * 256 + threadIdx.x; i < n_items; i += (int64_t)nblk * 256) {
      const float4 t = tlist[i];
      const int cid = grid_cell(g, t.x, t.y, t.z);
      atomicAdd(&cell_count[cid], 1);
      atomicAdd(&th[cid / GS_GRID_TILE], 1);
    }
  } else {
    const int64_t i = (int64_t)blk * 256 + threadIdx.x;
    if (i < n_tgt) {
      const int cid = grid_cell(g, tgt[3 * i], tgt[3 * i + 1], tgt[3 * i + 2]);
      atomicAdd(&cell_count[cid], 1);
      atomicAdd(&th[cid / GS_GRID_TILE], 1);
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < ntile; t += 256) {
    const int v = th[t];
    if (v) atomicAdd(&tile_sums[t], v);
  }
}
__global__ void __launch_bounds__(256) gs_grid_count_kernel(const float* __restrict__ tgt, GsCount n_tgt_c,
                                                            const GsTargetFilter flt,
                                                            const unsigned* __restrict__ bbox,
                                                            GsGrid* __restrict__ gp,
                                                            int* __restrict__ cell_count, int* __restrict__ tile_sums,
                                                            int cells_cap, const float4* __restrict__ tlist) {
  grid_count_body(tgt, gs_count(n_tgt_c), flt, bbox, gp, cell_count, tile_sums, cells_cap, tlist, blockIdx.x, gridDim.x);
}

// exclusive scan of cell_count[0 .. ncell] (ncell + 1 entries, the last one is the end sentinel)
GS_DEV void grid_scan_body(const int* __restrict__ cell_count, const GsGrid* __restrict__ gp,
                           const int* __restrict__ tile_sums, int* __restrict__ cell_start, const unsigned blk) {
  __shared__ int smem[256 / GS_WAVE + 1];
  const int n = gp->ncell + 1;
  if ((int)(blk * GS_GRID_TILE) >= n) return;
  // prefix of the tiles before this one
  int pre = 0;
  for (int t = threadIdx.x; t < (int)blk; t += 256) pre += tile_sums[t];
  int tile_prefix;
  (void)gs_block_excl_scan<256>(pre, smem, &tile_prefix);
  const int base = blk * GS_GRID_TILE + threadIdx.x * 4;
  int v[4], c = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[i] = (base + i < n) ? cell_count[base + i] : 0;
    c += v[i];
  }
  int total;
  int run = tile_prefix + gs_block_excl_scan<256>(c, smem, &total);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (base + i < n) cell_start[base + i] = run;
    run += v[i];
  }
}
__global__ void __launch_bounds__(256) gs_grid_scan_kernel(const int* __restrict__ cell_count,
                                                           const GsGrid* __restrict__ gp,
                                                           const int* __restrict__ tile_sums,
                                                           int* __restrict__ cell_start) {
  grid_scan_body(cell_count, gp, tile_sums, cell_start, blockIdx.x);
}

GS_DEV void grid_scatter_body(const float* __restrict__ tgt, const int64_t n_tgt, const GsTargetFilter flt,
                              const GsGrid* __restrict__ gp, const int* __restrict__ cell_start,
                              int* __restrict__ cell_count, float4* __restrict__ sorted,
                              const float4* __restrict__ tlist, const unsigned* __restrict__ bbox,
                              const float* __restrict__ nrm, float4* __restrict__ sorted_n, const unsigned blk,
                              const unsigned nblk) {
  if (flt.pix) {
    const int64_t n_list = (int64_t)bbox[6];
    if ((int64_t)blk * 256 >= n_list) return;
    const GsGrid g = *gp;
    for (int64_t i = (int64_t)blk * 256 + threadIdx.x; i < n_list; i += (int64_t)nblk * 256) {
      const float4 t = tlist[i];
      const int cid = grid_cell(g, t.x, t.y, t.z);
      const int slot = cell_start[cid] + atomicSub(&cell_count[cid], 1) - 1;
      sorted[slot] = t;
      if (nrm) {  // the normal travels with the point: the ICP kernels never gather from the map arrays
        const int64_t r = (int64_t)__float_as_int(t.w);
        sorted_n[slot] = make_float4(nrm[3 * r], nrm[3 * r + 1], nrm[3 * r + 2], 0.0f);
      }
    }
    return;
  }
  const int64_t i = (int64_t)blk * 256 + threadIdx.x;
  if (i >= n_tgt) return;
  const GsGrid g = *gp;
  const float x = tgt[3 * i], y = tgt[3 * i + 1], z = tgt[3 * i + 2];
  const int cid = grid_cell(g, x, y, z);
  // slots of a cell are handed out back to front; the order inside a cell is irrelevant because
  // queries order candidates by (distance, original index).  Leaves cell_count all zero again.
  const int slot = cell_start[cid] + atomicSub(&cell_count[cid], 1) - 1;
  sorted[slot] = make_float4(x, y, z, __int_as_float((int)i));
  if (nrm) sorted_n[slot] = make_float4(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2], 0.0f);
}
__global__ void __launch_bounds__(256) gs_grid_scatter_kernel(const float* __restrict__ tgt, GsCount n_tgt_c,
                                                              const GsTargetFilter flt,
                                                              const GsGrid* __restrict__ gp,
                                                              const int* __restrict__ cell_start,
                                                              int* __restrict__ cell_count,
                                                              float4* __restrict__ sorted,
                                                              const float4* __restrict__ tlist,
                                                              const unsigned* __restrict__ bbox,
                                                              const float* __restrict__ nrm,
                                                              float4* __restrict__ sorted_n) {
  grid_scatter_body(tgt, gs_count(n_tgt_c), flt, gp, cell_start, cell_count, sorted, tlist, bbox, nrm, sorted_n, blockIdx.x,
                    gridDim.x);
}

// ---- batched build: block b of a launch works for sequence b % B on its block b / B ----
__global__ void __launch_bounds__(GB_BLOCK) gs_gridb_bbox_kernel(const GsGridBatch gb, float u_hi, float v_hi) {
  gridb_bbox_block(gb, blockIdx.x, u_hi, v_hi);
}
__global__ void __launch_bounds__(256) gs_gridb_count_kernel(const GsGridBatch gb) {
  const GsGridSeq& q = gb.s[blockIdx.x % gb.B];
  grid_count_body(q.tgt, gs_count(q.n_tgt), GsTargetFilter{q.pix, gb.W, gb.ds}, q.m.bbox, q.m.g, q.m.cell_count,
                  q.m.tile_sums, gb.cells_cap, q.m.tlist, blockIdx.x / gb.B, gridDim.x / gb.B);
}
__global__ void __launch_bounds__(256) gs_gridb_scan_kernel(const GsGridBatch gb) {
  const GsGridSeq& q = gb.s[blockIdx.x % gb.B];
  grid_scan_body(q.m.cell_count, q.m.g, q.m.tile_sums, q.m.cell_start, blockIdx.x / gb.B);
}
__global__ void __launch_bounds__(256) gs_gridb_scatter_kernel(const GsGridBatch gb) {
  const GsGridSeq& q = gb.s[blockIdx.x % gb.B];
  grid_scatter_body(q.tgt, gs_count(q.n_tgt), GsTargetFilter{q.pix, gb.W, gb.ds}, q.m.g, q.m.cell_start, q.m.cell_count,
                    q.m.sorted, q.m.tlist, q.m.bbox, q.nrm, q.m.sorted_n, blockIdx.x / gb.B, gridDim.x / gb.B);
}

// Cells the grid of a build may use (what is cleared and scanned per build): the target density the
// cell size has to follow grows with the query lattice (the targets are map surfels seen on the same
// lattice, several per pixel in a mature map), so the budget is tied to n_src; a host-exact n_tgt
// (API calls) may raise it.  1M cells for a 640x480 frame, 4M for 1296x968.
static int grid_cells_cap(int64_t n_src, int64_t n_tgt_exact) {
  int64_t want = 48 * n_src;
  if (16 * n_tgt_exact > want) want = 16 * n_tgt_exact;
  int cells_cap = 1 << 20;
  while (cells_cap < GS_GRID_MAXCELL && cells_cap < want) cells_cap <<= 1;
  return cells_cap;
}
int gs_knn_grid_cells_cap(int64_t n_src) { return grid_cells_cap(n_src, 0); }
size_t gs_knn_grid_clear_bytes(const GridMem& m, int cells_cap) {
  // from the 256-byte aligned start of the scratch and rounded up to 256 bytes (the header is rewritten by the
  // count kernel and the bytes of cell_start the round-up may touch are rewritten by the scan)
  return gs_align((size_t)(reinterpret_cast<char*>(m.cell_count) - reinterpret_cast<char*>(m.g)) +
                  4 * (size_t)(cells_cap + 1));
}

int gs_knn_grid_build_batch(const GsGridBatch& gb, hipStream_t st, bool bbox_done) {
  int64_t n_max = 1;
  for (int b = 0; b < gb.B; ++b) n_max = gb.s[b].n_tgt.host > n_max ? gb.s[b].n_tgt.host : n_max;
  const unsigned B = (unsigned)gb.B;
  const float u_hi = (float)((double)gb.W - 0.999), v_hi = (float)((double)gb.H - 0.999);
  if (!bbox_done)   // (else the caller ran gridb_bbox_block for gs_knn_gridb_bbox_blocks(gb) blocks in a launch of its own)
    hipLaunchKernelGGL(gs_gridb_bbox_kernel, dim3(gs_knn_gridb_bbox_blocks(gb)), dim3(GB_BLOCK), 0, st, gb, u_hi, v_hi);
  // filtered builds walk the compacted target list (a few entries per lattice slot) with a grid-stride loop
  bool listed = true;
  for (int b = 0; b < gb.B; ++b) listed = listed && gb.s[b].pix != nullptr;
  const int64_t slots = (int64_t)gs_ceil_div(gb.H, gb.ds) * gs_ceil_div(gb.W, gb.ds);
  unsigned nb_pts = (unsigned)gs_ceil_div(n_max, 256);
  if (listed && gb.H > 0 && (unsigned)gs_ceil_div(3 * slots, 256) < nb_pts) nb_pts = (unsigned)gs_ceil_div(3 * slots, 256);
  hipLaunchKernelGGL(gs_gridb_count_kernel, dim3(B * nb_pts), dim3(256), 0, st, gb);
  const unsigned ntile = (unsigned)gs_ceil_div(gb.cells_cap + 1, GS_GRID_TILE);
  hipLaunchKernelGGL(gs_gridb_scan_kernel, dim3(B * ntile), dim3(256), 0, st, gb);
  hipLaunchKernelGGL(gs_gridb_scatter_kernel, dim3(B * nb_pts), dim3(256), 0, st, gb);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { gs_set_error("gs_knn_grid_build_batch: %s", hipGetErrorString(e)); return GS_ERR_HIP; }
  return GS_OK;
}

int gs_knn_grid_build(const float* tgt, GsCount n_tgt_c, int64_t n_src, void* grid_scratch, hipStream_t st,
                      GsTargetFilter flt, const float* nrm) {
  const int64_t n_tgt = n_tgt_c.host;  // upper bound: launch geometry and scratch layout
  GridMem m = grid_carve(grid_scratch, n_src, n_tgt);
  const int cells_cap = grid_cells_cap(n_src, n_tgt_c.dev ? 0 : n_tgt);
  GsProf prof(GS_PROF_COMPACT, 28.0 * (double)n_tgt + 8.0 * (double)cells_cap, st);
  // one memset: bbox codes + unresolved counters + cell counts (contiguous in the scratch layout; 256-byte
  // aligned and rounded so that the runtime issues ONE fill kernel, not a head + body pair)
  const size_t clear = gs_knn_grid_clear_bytes(m, cells_cap);
  hipError_t e = hipMemsetAsync(m.g, 0, clear, st);
  if (e != hipSuccess) { gs_set_error("gs_knn_grid_build: %s", hipGetErrorString(e)); return GS_ERR_HIP; }
  hipLaunchKernelGGL(gs_grid_bbox_kernel, dim3((unsigned)gs_ceil_div(n_tgt > 0 ? n_tgt : 1, GB_BLOCK * GB_ITEMS)),
                     dim3(GB_BLOCK), 0, st, tgt, n_tgt_c, flt, m.bbox, m.unres_count, m.tlist);
  hipLaunchKernelGGL(gs_grid_count_kernel, dim3((unsigned)gs_ceil_div(n_tgt > 0 ? n_tgt : 1, 256)), dim3(256), 0, st,
                     tgt, n_tgt_c, flt, m.bbox, m.g, m.cell_count, m.tile_sums, cells_cap, m.tlist);
  const unsigned ntile = (unsigned)gs_ceil_div(cells_cap + 1, GS_GRID_TILE);
  hipLaunchKernelGGL(gs_grid_scan_kernel, dim3(ntile), dim3(256), 0, st, m.cell_count, m.g, m.tile_sums,
                     m.cell_start);
  hipLaunchKernelGGL(gs_grid_scatter_kernel, dim3((unsigned)gs_ceil_div(n_tgt, 256)), dim3(256), 0, st, tgt,
                     n_tgt_c, flt, m.g, m.cell_start, m.cell_count, m.sorted, m.tlist, m.bbox, nrm, m.sorted_n);
  return GS_OK;
}

constexpr int GQ_BLOCK = 256;

__global__ void __launch_bounds__(GQ_BLOCK) gs_grid_query_kernel(
    const float* __restrict__ src_in, const float* __restrict__ Tapply, float* __restrict__ src_out,
    int64_t n_src, const GsGrid* __restrict__ gp, const int* __restrict__ cell_start,
    const float4* __restrict__ sorted, unsigned long long* __restrict__ best, int* __restrict__ unres_count,
    int* __restrict__ unres_next, int* __restrict__ unres_list) {
  const int lane = threadIdx.x & (GQ_G - 1);
  const int64_t s = ((int64_t)blockIdx.x * GQ_BLOCK + threadIdx.x) / GQ_G;
  if (blockIdx.x == 0 && threadIdx.x == 0) *unres_next = 0;  // arm the counter of the NEXT query (ping-pong)
  if (s >= n_src) return;  // whole GQ_G-lane groups leave together
  const GsGrid g = *gp;
  float qx = src_in[3 * s], qy = src_in[3 * s + 1], qz = src_in[3 * s + 2];
  if (Tapply) {
    float T[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = Tapply[i];
    float t0, t1, t2;
    gs_rigid_fma(T, qx, qy, qz, t0, t1, t2);
    qx = t0; qy = t1; qz = t2;
  }
  if (src_out && lane == 0) {
    src_out[3 * s] = qx;
    src_out[3 * s + 1] = qy;
    src_out[3 * s + 2] = qz;
  }
  bool done;
  const unsigned long long key = grid_search_group(g, cell_start, sorted, qx, qy, qz, lane, &done);
  if (lane == 0) {
    if (done) {
      best[s] = key;
    } else {
      best[s] = ~0ull;  // finished by the brute-force pass
      unres_list[atomicAdd(unres_count, 1)] = (int)s;
    }
  }
}

int gs_knn_grid_query(const float* src_in, const float* Tapply, float* src_out, int64_t n_src,
                      const float* tgt, int64_t n_tgt, unsigned long long* best, void* grid_scratch,
                      hipStream_t st) {
  static unsigned seq = 0;  // ping-pong index of the unresolved counter (one host thread per GPU)
  GridMem m = grid_carve(grid_scratch, n_src, n_tgt);
  int* cnt = m.unres_count + (seq & 1);
  int* nxt = m.unres_count + ((seq + 1) & 1);
  ++seq;
  GsProf prof(GS_PROF_KNN, (double)n_src * (double)n_tgt, st);  // brute-force-equivalent pairs
  hipLaunchKernelGGL(gs_grid_query_kernel, dim3((unsigned)gs_ceil_div(n_src * GQ_G, GQ_BLOCK)), dim3(GQ_BLOCK), 0, st,
                     src_in, Tapply, src_out, n_src, m.g, m.cell_start, m.sorted, best, cnt, nxt, m.unres_list);
  // unresolved queries (device-side count; blocks exit at once when it is zero)
  unsigned gx = (unsigned)gs_ceil_div(n_src, KNN_BLOCK * 4);
  gx = gx > 4 ? 4 : gx;
  dim3 grid(gx, (unsigned)gs_ceil_div(n_tgt, KNN_TCHUNK));
  hipLaunchKernelGGL((gs_knn1_kernel<true, 4>), grid, dim3(KNN_BLOCK), 0, st, src_in, Tapply, nullptr, n_src, tgt,
                     n_tgt, best, m.unres_list, cnt);
  return GS_OK;
}

// ---------------------------------------------------------------- C-ABI ----------------
__global__ void __launch_bounds__(256) gs_knn_unpack_kernel(const unsigned long long* __restrict__ best,
                                                            int64_t n, int64_t n_tgt, int64_t* __restrict__ idx,
                                                            float* __restrict__ d2) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const unsigned long long b = best[i];
  int64_t j = (int64_t)(b & 0xffffffffull);
  if (j >= n_tgt) j = 0;  // only when every distance was NaN
  idx[i] = j;
  if (d2) d2[i] = __uint_as_float((uint32_t)(b >> 32));
}

extern "C" int gs_knn1_f32(const float* src, int64_t n_src, const float* tgt, int64_t n_tgt,
                           int64_t* out_idx, float* out_d2, uint64_t* best_scratch, void* stream) {
  GS_REQUIRE(n_src > 0 && n_tgt > 0, "empty point set");
  GS_REQUIRE(n_tgt < 0x7fffffffll && n_src < 0x7fffffffll, "too many points");
  GS_REQUIRE(src && tgt && out_idx && best_scratch, "NULL pointer");
  hipStream_t st = gs_stream(stream);
  GS_HIP(hipMemsetAsync(best_scratch, 0xff, 8 * (size_t)n_src, st));
  gs_knn_brute_launch(src, nullptr, nullptr, n_src, tgt, n_tgt, reinterpret_cast<unsigned long long*>(best_scratch), st);
  hipLaunchKernelGGL(gs_knn_unpack_kernel, dim3((unsigned)gs_ceil_div(n_src, 256)), dim3(256), 0, st,
                     reinterpret_cast<const unsigned long long*>(best_scratch), n_src, n_tgt, out_idx, out_d2);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

extern "C" int64_t gs_knn1_grid_scratch_bytes(int64_t n_src, int64_t n_tgt) {
  return (int64_t)(gs_align(8 * (size_t)(n_src > 0 ? n_src : 1)) + gs_knn_grid_scratch_bytes(n_src, n_tgt));
}

extern "C" int gs_knn1_grid_f32(const float* src, int64_t n_src, const float* tgt, int64_t n_tgt,
                                int64_t* out_idx, float* out_d2, void* scratch, int64_t* unresolved_out,
                                void* stream) {
  GS_REQUIRE(n_src > 0 && n_tgt > 0, "empty point set");
  GS_REQUIRE(n_tgt < 0x7fffffffll && n_src < 0x7fffffffll, "too many points");
  GS_REQUIRE(src && tgt && out_idx && scratch, "NULL pointer");
  hipStream_t st = gs_stream(stream);
  unsigned long long* best = reinterpret_cast<unsigned long long*>(scratch);
  void* gscratch = reinterpret_cast<char*>(scratch) + gs_align(8 * (size_t)n_src);
  GS_HIP(hipMemsetAsync(best, 0xff, 8 * (size_t)n_src, st));
  int rc = gs_knn_grid_build(tgt, GsCount{n_tgt, nullptr}, n_src, gscratch, st);
  if (rc != GS_OK) return rc;
  GridMem m = grid_carve(gscratch, n_src, n_tgt);
  rc = gs_knn_grid_query(src, nullptr, nullptr, n_src, tgt, n_tgt, best, gscratch, st);
  if (rc != GS_OK) return rc;
  hipLaunchKernelGGL(gs_knn_unpack_kernel, dim3((unsigned)gs_ceil_div(n_src, 256)), dim3(256), 0, st, best, n_src,
                     n_tgt, out_idx, out_d2);
  if (unresolved_out) {  // diagnostic: how many queries needed the brute-force pass (sum of both counters:
    // exactly one of them was used by this query, the other one is zero)
    int h[2] = {0, 0};
    GS_HIP(hipMemcpyAsync(h, m.unres_count, 8, hipMemcpyDeviceToHost, st));
    GS_HIP(hipStreamSynchronize(st));
    *unresolved_out = (int64_t)h[0] + (int64_t)h[1];
  }
  GS_LAUNCH_CHECK();
  return GS_OK;
}
