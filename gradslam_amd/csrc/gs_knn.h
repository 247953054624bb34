// gs_knn.h — internal interface of the exact 1-NN engine (gs_knn.hip), shared with gs_icp.hip
// (the fused search + linearise kernels of the ICP loop inline the device-side search).
#pragma once
#include "gs_common.h"

// best[s] = (float_bits(d2) << 32) | target_index, the minimum over all targets; lowest index on
// ties.  Callers arm best[] with all ones (memset 0xff) before a brute-force search.
GS_DEV unsigned long long knn_pack(float d, uint32_t idx) {
  return ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)idx;
}

// Brute force over every (src, tgt) pair.  Tapply (device, 12+ floats, may be NULL) is applied to
// every source point on load; src_out (may be NULL) receives the transformed points.
int gs_knn_brute_launch(const float* src_in, const float* Tapply, float* src_out, int64_t n_src,
                        const float* tgt, int64_t n_tgt, unsigned long long* best, hipStream_t st);

// ---------------------------------------------------------------- uniform grid ---------
// Build once per target set, query many times.  Results are IDENTICAL to the brute-force search
// (same distances, same tie-break); queries the grid cannot resolve within GS_GRID_RINGS shells
// are finished by brute force (a separate pass in gs_knn_grid_query, in-block in the fused ICP
// kernels).
constexpr int GS_GRID_MAXCELL = 1 << 22;
constexpr int GS_GRID_RINGS = 3;
constexpr int GS_GRID_TILE = 1024;

struct GsGrid {
  float ox, oy, oz;  // bounding box of the (finite) targets
  float mx, my, mz;
  float c, inv_c;    // cell edge
  int nx, ny, nz, ncell;
};

// Which rows of the target array take part: all of them (pix == NULL), or only the map rows whose projection
// lands on the [::ds, ::ds] pixel lattice (the selection of gs_select_targets_f32, evaluated in place: the
// ICP target set of the SLAM loop is then never gathered into a compact array; candidate keys carry the
// map row index, whose order is the order of the compacted set, so ties break identically).
struct GsTargetFilter {
  const int32_t* pix;
  int W, ds;
};
GS_DEV bool gs_is_target(const GsTargetFilter& f, int64_t n) {
  if (!f.pix) return true;
  const int32_t p = f.pix[n];
  return p >= 0 && ((p / f.W) % f.ds == 0) && ((p % f.W) % f.ds == 0);
}

struct GridMem {
  GsGrid* g;
  unsigned* bbox;    // [6] order-preserving codes of the bounding box (max of ~code(lo), code(hi)) + [6] = number
                     // of rows that passed the target filter; zeroed per build
  int* unres_count;  // [2], ping-pong between consecutive queries
  int* cell_count;   // [MAXCELL + 1]
  int* cell_start;   // [MAXCELL + 1]
  int* tile_sums;    // [MAXCELL / 1024 + 2]
  float4* sorted;    // [n_tgt] (x, y, z, original index bits), grouped by cell
  float4* sorted_n;  // [n_tgt] normal of the target in the same slot of `sorted` (builds that are given normals)
  float4* tlist;     // [n_tgt] filtered builds: the rows that passed the filter, compacted by the bbox pass (any order)
  int* unres_list;   // [n_src]
};

static inline GridMem grid_carve(void* scratch, int64_t n_src, int64_t n_tgt) {
  char* p = reinterpret_cast<char*>(scratch);
  GridMem m;
  m.g = reinterpret_cast<GsGrid*>(p); p += 128;
  m.bbox = reinterpret_cast<unsigned*>(p); p += 128;
  m.unres_count = reinterpret_cast<int*>(p); p += 256;
  // (the tile sums sit in front of the cell counts: both are accumulated by the count pass and must be zero before it,
  // so they are part of the one cleared range, gs_knn_grid_clear_bytes)
  m.tile_sums = reinterpret_cast<int*>(p); p += gs_align(4 * (size_t)(GS_GRID_MAXCELL / GS_GRID_TILE + 2));
  m.cell_count = reinterpret_cast<int*>(p); p += gs_align(4 * (size_t)(GS_GRID_MAXCELL + 1));
  m.cell_start = reinterpret_cast<int*>(p); p += gs_align(4 * (size_t)(GS_GRID_MAXCELL + 1));
  m.sorted = reinterpret_cast<float4*>(p); p += gs_align(16 * (size_t)(n_tgt > 0 ? n_tgt : 1));
  m.sorted_n = reinterpret_cast<float4*>(p); p += gs_align(16 * (size_t)(n_tgt > 0 ? n_tgt : 1));
  m.tlist = reinterpret_cast<float4*>(p); p += gs_align(16 * (size_t)(n_tgt > 0 ? n_tgt : 1));
  m.unres_list = reinterpret_cast<int*>(p);
  (void)n_src;
  return m;
}

size_t gs_knn_grid_scratch_bytes(int64_t n_src, int64_t n_tgt);

// ---- batched build: B independent target sets per launch (block b of a launch works for sequence b % B, so
// that with B = 8 every sequence stays on one XCD and its binned targets in that XCD's L2) ----
struct GsGridSeq {
  const float* tgt;      // target rows (the whole map when a filter is given)
  GsCount n_tgt;
  int32_t* pix;          // target filter (with W, ds of the batch); when pose16 != NULL it is WRITTEN first:
  const float* pose16;   //   pix[n] = projection of row n under (pose16, K16) (gs_project_map_f32)
  const float* K16;
  const float* nrm;      // normals of the target rows (NULL: none): binned next to the points (m.sorted_n)
  GridMem m;             // grid_carve(scratch, n_src, n_tgt.host)
};
struct GsGridBatch {
  int B, H, W, ds;
  int cells_cap;
  GsGridSeq s[GS_MAX_BATCH];
};
// cells_cap of a build for n_src queries (device-side target counts)
int gs_knn_grid_cells_cap(int64_t n_src);
// bytes from the start of the grid scratch that must be ZERO before a build (header, bbox, counters, cell counts)
size_t gs_knn_grid_clear_bytes(const GridMem& m, int cells_cap);
// projection (optional) + bbox, count (+ tile sums), scan, scatter: 4 launches for all B sequences; the caller has
// cleared gs_knn_grid_clear_bytes() of every scratch
// bbox_done: the caller already ran the bbox pass (gridb_bbox_block of gs_knn_bbox.h) in a launch of its own
int gs_knn_grid_build_batch(const GsGridBatch& gb, hipStream_t st, bool bbox_done = false);

int gs_knn_grid_build(const float* tgt, GsCount n_tgt, int64_t n_src, void* grid_scratch, hipStream_t st,
                      GsTargetFilter filter = GsTargetFilter{nullptr, 1, 1}, const float* nrm = nullptr);
int gs_knn_grid_query(const float* src_in, const float* Tapply, float* src_out, int64_t n_src,
                      const float* tgt, int64_t n_tgt, unsigned long long* best, void* grid_scratch,
                      hipStream_t st);
// Heuristic used by the ICP loop: the grid pays off once the target set is large.
static inline bool gs_knn_use_grid(int64_t n_src, int64_t n_tgt) { return n_tgt >= 2048 && n_src >= 256; }

GS_DEV int grid_axis(float v, float o, float inv_c, int n) {
  const float f = (v - o) * inv_c;
  // NaN fails both comparisons and lands in cell 0
  return f >= 0.0f ? (f < (float)n ? (int)f : n - 1) : 0;
}
GS_DEV int grid_cell(const GsGrid& g, float x, float y, float z) {
  const int ix = grid_axis(x, g.ox, g.inv_c, g.nx);
  const int iy = grid_axis(y, g.oy, g.inv_c, g.ny);
  const int iz = grid_axis(z, g.oz, g.inv_c, g.nz);
  return (iz * g.ny + iy) * g.nx + ix;
}

// A query is served by a group of G lanes (G = 8, 4 or 2; 64 / G queries per wave): the lanes first fetch the
// [begin, end) bounds of the row segments of a shell in parallel, then stride together over every
// segment, and finally min-reduce their packed (distance bits << 32 | index) keys -- the same
// ordering as the brute-force engine's 64-bit atomicMin, so the result does not depend on G.  This turns ~100
// serial dependent gathers per query into ~15 wave-wide ones (the search is latency-, not bandwidth-bound).
// Which G is fastest depends on how many queries have to share the chip: ONE 640x480 sequence per GPU (19 200
// queries on 256 CUs) is served best by 8 lanes per query (measured 12.6 / 10.7 / 11.5 us per ICP half-iteration
// with 16 / 8 / 4 lanes: every dependent launch pays for the waves it has to start); with 8 sequences per GPU a
// sequence owns one XCD (32 CUs x 24 waves), where only 2 lanes per query keep all its queries in flight at once.
constexpr int GQ_G = 8;  // group size of the API-level query kernel and of one-sequence-per-GPU solves
constexpr int GL_RING_NF = 2;   // gathers in flight per lane of a cube scan (four cost the 2- and 4-lane list variants of the
                                // half-iteration kernel a spill)

GS_DEV unsigned long long grid_key(float qx, float qy, float qz, const float4 p) {
  const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
  float d = dx * dx;
  d = gs_fma(dy, dy, d);
  d = gs_fma(dz, dz, d);
  // a NaN distance never wins (brute force: `d < best` is false for NaN)
  return d == d ? knn_pack(d, (uint32_t)__float_as_int(p.w)) : ~0ull;
}

template <int G>
GS_DEV unsigned long long grid_group_min(unsigned long long key) {
#pragma unroll
  for (int d = G / 2; d > 0; d >>= 1) {
    const unsigned long long o = __shfl_xor(key, d, G);
    key = o < key ? o : key;
  }
  return key;
}

// Search of one query by a group of G lanes (all lanes of the group call with the same query), in two parts:
//   grid_search_stage0: the 2x2x2 block of cells whose centre is nearest to the query -- resolves almost every ICP
//                       query (its neighbour sits within a fraction of a cell);
//   grid_search_rings : cubes of Chebyshev radius 1 .. GS_GRID_RINGS around the query's cell for the rest (a few % of the
//                       queries of a frame whose border looks at surface the map has not seen: distances of 0.5 - 2
//                       cells); the fused ICP kernels run this part with wider groups on the collected leftovers so
//                       that a handful of far queries does not hold up whole waves.
// Both return the packed best key (identical in all lanes); *resolved tells whether it is provably the global
// minimum.  grid_search_group = stage 0, then the rings if needed, with the same group.
struct GsQueryCell {
  float px, py, pz;  // projection of the query onto the bounding box
  int cx, cy, cz;
};
GS_DEV GsQueryCell grid_query_cell(const GsGrid& g, float qx, float qy, float qz) {
  // cell of the query's projection onto the bounding box (the projection onto a convex set never
  // increases the distance to points inside it, so shell bounds around it stay valid)
  GsQueryCell c;
  // (compare + select rather than fminf / fmaxf: those canonicalise the block-uniform box corners into six VGPRs
  // that stay live through the whole search; queries are never NaN here, the callers skip such points)
  c.px = qx < g.ox ? g.ox : (qx > g.mx ? g.mx : qx);
  c.py = qy < g.oy ? g.oy : (qy > g.my ? g.my : qy);
  c.pz = qz < g.oz ? g.oz : (qz > g.mz ? g.mz : qz);
  c.cx = grid_axis(c.px, g.ox, g.inv_c, g.nx); c.cy = grid_axis(c.py, g.oy, g.inv_c, g.ny);
  c.cz = grid_axis(c.pz, g.oz, g.inv_c, g.nz);
  return c;
}

template <int G>
GS_DEV unsigned long long grid_search_stage0(const GsGrid& g, const int* __restrict__ cell_start,
                                             const float4* __restrict__ sorted, float qx, float qy, float qz, int lane,
                                             bool* resolved, int* win, const float rball = __builtin_inff()) {
  const GsQueryCell qc = grid_query_cell(g, qx, qy, qz);
  const float px = qc.px, py = qc.py, pz = qc.pz;
  const int cx = qc.cx, cy = qc.cy, cz = qc.cz;
  unsigned long long key = ~0ull;
  int bt = -1;  // flat position (in the candidate list) of this lane's best candidate
  // The 2x2x2 block of cells whose centre is nearest to the (projected) query.  Every target
  // outside that box is at least as far as the nearest box face that has cells behind it (>= half a cell
  // by construction); ICP queries sit within a fraction of a cell of their neighbour, so most searches
  // end here with 8 cells (4 row segments) instead of 27.
  const float fx = (px - g.ox) * g.inv_c - (float)cx, fy = (py - g.oy) * g.inv_c - (float)cy,
              fz = (pz - g.oz) * g.inv_c - (float)cz;
  const int x0 = (fx < 0.5f) ? cx - 1 : cx, y0 = (fy < 0.5f) ? cy - 1 : cy, z0 = (fz < 0.5f) ? cz - 1 : cz;
  // distance (in cells) from the query to the nearest box face with cells behind it, per axis
  const float BIG = 3.0e38f;
  const float ax = fminf(x0 >= 1 ? fx + (float)(cx - x0) : BIG, x0 + 2 < g.nx ? (float)(x0 + 2 - cx) - fx : BIG);
  const float ay = fminf(y0 >= 1 ? fy + (float)(cy - y0) : BIG, y0 + 2 < g.ny ? (float)(y0 + 2 - cy) - fy : BIG);
  const float az = fminf(z0 >= 1 ? fz + (float)(cz - z0) : BIG, z0 + 2 < g.nz ? (float)(z0 + 2 - cz) - fz : BIG);
  const float amin = fminf(ax, fminf(ay, az));
  // rball: a radius within which the caller KNOWS a target lies (the previous neighbour of this source point, whose
  // distance can only have grown by the displacement of the query since).  The nearest target is then inside that
  // ball, and when the ball stays inside the box only the cells it touches can hold it: about half of the eight.
  // Margins: 0.01 % on the radius, 0.002 cells on the cell assignment (float rounding is orders of magnitude below).
  const float rc = rball * g.inv_c * 1.0001f + 0.002f;
  const bool prune = rc < amin - 0.001f;  // false for rball = inf / NaN
  bool ulx = true, uhx = true, uly = true, uhy = true, ulz = true, uhz = true;
  if (prune) {
    const float tx = fx + (float)(cx - x0), ty = fy + (float)(cy - y0), tz = fz + (float)(cz - z0);  // in [0.5, 1.5)
    ulx = tx - rc < 1.0f; uhx = tx + rc >= 1.0f;
    uly = ty - rc < 1.0f; uhy = ty + rc >= 1.0f;
    ulz = tz - rc < 1.0f; uhz = tz + rc >= 1.0f;
  }
  // The 4 row segments of the box (2 cells each) as ONE flat candidate list.  Every lane of the group fetches all
  // 8 segment bounds itself (the lanes' addresses coincide, so this costs one round trip and no cross-lane
  // traffic), then lane l takes the flat positions l, l + G, ...: position -> (segment, offset) is three
  // compares on registers.  No shuffles until the final min.
  int sb0 = 0, sb1 = 0, sb2 = 0, sb3 = 0, e1 = 0, e2 = 0, e3 = 0, total = 0;
  {
    // (the cell of the query itself is never dropped: p sits in it, so at least one of each pair holds)
    const int xa = (x0 >= 0 && ulx) ? x0 : x0 + 1, xb = (x0 + 1 < g.nx && uhx) ? x0 + 1 : x0;
    const bool zl = z0 >= 0 && ulz, zh = z0 + 1 < g.nz && uhz, yl = y0 >= 0 && uly, yh = y0 + 1 < g.ny && uhy;
    const int r0 = (z0 * g.ny + y0) * g.nx, r1 = r0 + g.nx, r2 = r0 + g.ny * g.nx, r3 = r2 + g.nx;
    int se0 = 0, se1 = 0, se2 = 0, se3 = 0;
    if (zl && yl) { sb0 = cell_start[r0 + xa]; se0 = cell_start[r0 + xb + 1]; }
    if (zl && yh) { sb1 = cell_start[r1 + xa]; se1 = cell_start[r1 + xb + 1]; }
    if (zh && yl) { sb2 = cell_start[r2 + xa]; se2 = cell_start[r2 + xb + 1]; }
    if (zh && yh) { sb3 = cell_start[r3 + xa]; se3 = cell_start[r3 + xb + 1]; }
    e1 = se0 - sb0;
    e2 = e1 + (se1 - sb1);
    e3 = e2 + (se2 - sb2);
    total = e3 + (se3 - sb3);
  }
  for (int t0 = lane; t0 < total; t0 += 4 * G) {  // four independent gathers in flight per lane
    float4 p[4];
    bool in[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = t0 + u * G;
      in[u] = t < total;
      const int tt = in[u] ? t : 0;  // total > 0 here: position 0 is valid
      const int ix = tt < e1 ? sb0 + tt : (tt < e2 ? sb1 + (tt - e1) : (tt < e3 ? sb2 + (tt - e2) : sb3 + (tt - e3)));
      p[u] = sorted[ix];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned long long k2 = in[u] ? grid_key(qx, qy, qz, p[u]) : ~0ull;
      const bool better = k2 < key;
      key = better ? k2 : key;
      bt = better ? t0 + u * G : bt;
    }
  }
  {
    // keys are unique (they carry the target index): at most one lane of the group holds the minimum; it reports
    // the slot of that candidate in `sorted`
    const unsigned long long kmin = grid_group_min<G>(key);
    const int bs = bt < e1 ? sb0 + bt : (bt < e2 ? sb1 + (bt - e1) : (bt < e3 ? sb2 + (bt - e2) : sb3 + (bt - e3)));
    *win = (key == kmin && bt >= 0) ? bs : -1;
    key = kmin;
  }
  // 0.1 % of a cell is orders of magnitude above the float rounding of the cell assignment
  const float rb = (amin < 1.0e30f) ? (amin - 0.001f) * g.c : BIG;
  const float bd = __uint_as_float((uint32_t)(key >> 32));  // NaN while nothing was found
  // pruned scan: the ball lies inside the box and holds a target, every cell it touches was scanned
  *resolved = prune ? (bd == bd) : (rb > 0.0f && (rb >= 1.0e30f ? bd == bd : bd <= rb * rb));
  return key;
}

// Cubes of Chebyshev radius 1 .. GS_GRID_RINGS around the query's cell: lane l of the group scans the rows
// l, l + G, ... of the cube (2k+1 cells each) on its own.  Deliberately simple (a few registers, no cross-lane traffic
// but the final min): it serves the few queries per frame whose neighbour is farther than half a cell.
template <int G>
GS_DEV unsigned long long grid_search_rings(const GsGrid& g, const int* __restrict__ cell_start,
                                            const float4* __restrict__ sorted, float qx, float qy, float qz, int lane,
                                            unsigned long long key, bool* resolved, int* win,
                                            const int kmax = GS_GRID_RINGS, int* kdone = nullptr) {
  const GsQueryCell qc = grid_query_cell(g, qx, qy, qz);
  bool done = false;
  int bs = -1;  // slot of a candidate of THIS call that beats the incoming key (-1: the incoming key stands)
  int k = 1;
  for (; k <= kmax && !done; ++k) {
    const int side = 2 * k + 1, nrow = side * side;
    const int xa = qc.cx - k < 0 ? 0 : qc.cx - k, xb = qc.cx + k >= g.nx ? g.nx - 1 : qc.cx + k;
    for (int r = lane; r < nrow; r += G) {
      const int zz = qc.cz + r / side - k, yy = qc.cy + r % side - k;
      if (zz < 0 || zz >= g.nz || yy < 0 || yy >= g.ny) continue;
      const int row = (zz * g.ny + yy) * g.nx;
      const int je = cell_start[row + xb + 1];
      for (int j = cell_start[row + xa]; j < je; j += GL_RING_NF) {   // (as grid_search_rings_top: gathers in flight)
        float4 p[GL_RING_NF];
#pragma unroll
        for (int u = 0; u < GL_RING_NF; ++u) p[u] = sorted[j + u < je ? j + u : je - 1];
#pragma unroll
        for (int u = 0; u < GL_RING_NF; ++u) {
          const unsigned long long k2 = j + u < je ? grid_key(qx, qy, qz, p[u]) : ~0ull;
          if (k2 < key) { key = k2; bs = j + u; }
        }
      }
    }
    {
      const unsigned long long own = key;
      key = grid_group_min<G>(key);
      if (own != key) bs = -1;
    }
    // every unvisited target is farther than k cells from the projected query; 0.1 % of a cell is
    // orders of magnitude above the float rounding of the cell assignment
    const float rb = (float)k * g.c * 0.999f;
    const float bd = __uint_as_float((uint32_t)(key >> 32));  // NaN while nothing was found
    done = bd <= rb * rb;
  }
  *resolved = done;
  *win = bs;
  if (kdone) *kdone = k - 1;   // radius of the last cube scanned
  return key;
}

// ---- candidate lists of far queries (the fused ICP kernels search the SAME target set 2 x numiters times from
// slowly moving query positions).  A query whose neighbour the 2x2x2 stage cannot prove -- it sits several cells from
// every target: a frame border that looks at surface the map has not seen -- pays a cube scan or a block-wide pass over
// all binned targets per search.  Once resolved it gets a list: ALL targets within R of its position q0 (<= 8 slots of
// `sorted`).  A later search from q is exact on the list alone when  sqrt(best list distance) + |q - q0| < R:  every
// target at least as close to q as the list's best is within R of q0, hence on the list (ties included), and the key
// order is that of every other engine.
constexpr int GS_FAR_SLOTS = 64;
constexpr float GS_FAR_RADD = 1.0f;    // R = distance of the neighbour + GS_FAR_RADD cells (reduced until the list fits)

// the list of one query, checked by a group of G lanes (GS_FAR_SLOTS / G entries each, one 16-byte load of slot numbers
// per lane): returns the minimum key, *proven = exactness
template <int G>
GS_DEV unsigned long long far_list_search(const float4 c0R, const uint32_t* __restrict__ slots,
                                          const float4* __restrict__ sorted, float qx, float qy, float qz, int lane,
                                          bool* proven, int* win) {
  static_assert(GS_FAR_SLOTS == 4 * G, "four list entries per lane");
  const uint4 sl = reinterpret_cast<const uint4*>(slots)[lane];
  const uint32_t s4[4] = {sl.x, sl.y, sl.z, sl.w};
  float4 a[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) a[u] = sorted[s4[u] != ~0u ? s4[u] : 0u];   // (the four gathers are in flight together)
  unsigned long long k = ~0ull;
  int bs = -1;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const unsigned long long k2 = s4[u] != ~0u ? grid_key(qx, qy, qz, a[u]) : ~0ull;
    if (k2 < k) { k = k2; bs = (int)s4[u]; }
  }
  const unsigned long long kmin = grid_group_min<G>(k);
  *win = (k == kmin) ? bs : -1;
  const float bd = __uint_as_float((uint32_t)(kmin >> 32));
  const float ex = qx - c0R.x, ey = qy - c0R.y, ez = qz - c0R.z;
  const float delta = sqrtf(ex * ex + ey * ey + ez * ez);
  *proven = sqrtf(bd) + delta < c0R.w * 0.9999f;   // false for NaN, for an empty list and for R <= 0
  return kmin;
}

// the same check with two gathers in flight per lane instead of four (the wide lists of round 5 are checked inside
// kernels that sit at their register limit; the check serves a handful of points per launch)
template <int G>
GS_DEV unsigned long long wide_list_search(const float4 c0R, const uint32_t* __restrict__ slots,
                                           const float4* __restrict__ sorted, float qx, float qy, float qz, int lane,
                                           bool* proven, int* win) {
  static_assert(GS_FAR_SLOTS == 4 * G, "four list entries per lane");
  unsigned long long k = ~0ull;
  int bs = -1;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const uint2 sl = reinterpret_cast<const uint2*>(slots)[2 * lane + h];
    const float4 a0 = sorted[sl.x != ~0u ? sl.x : 0u], a1 = sorted[sl.y != ~0u ? sl.y : 0u];
    const unsigned long long k0 = sl.x != ~0u ? grid_key(qx, qy, qz, a0) : ~0ull;
    const unsigned long long k1 = sl.y != ~0u ? grid_key(qx, qy, qz, a1) : ~0ull;
    if (k0 < k) { k = k0; bs = (int)sl.x; }
    if (k1 < k) { k = k1; bs = (int)sl.y; }
  }
  const unsigned long long kmin = grid_group_min<G>(k);
  *win = (k == kmin) ? bs : -1;
  const float bd = __uint_as_float((uint32_t)(kmin >> 32));
  const float ex = qx - c0R.x, ey = qy - c0R.y, ez = qz - c0R.z;
  const float delta = sqrtf(ex * ex + ey * ey + ez * ez);
  *proven = sqrtf(bd) + delta < c0R.w * 0.9999f;   // false for NaN, for an empty list and for R <= 0
  return kmin;
}

// list of a query that grid_search_rings served with the cube of radius kdone: every target within
// R = min(d1 + radd, 0.999 kE cells) of the query, kE = kdone or kdone + 1 (the larger cube when the smaller one leaves
// less than radd of room); collected by the G lanes into `stage` (GS_FAR_SLOTS x 32 bit + a counter, LDS of the group).
// Returns R, or 0 when no radius down to d1 + radd / 4 gives a list that fits.
template <int G>
GS_DEV float far_emit_cube(const GsGrid& g, const int* __restrict__ cell_start, const float4* __restrict__ sorted,
                           float qx, float qy, float qz, int lane, const float d1, const int kdone, uint32_t* stage,
                           int* stage_n) {
  const GsQueryCell qc = grid_query_cell(g, qx, qy, qz);
  float radd = GS_FAR_RADD * g.c, R = 0.0f;
  const int kE = (d1 + radd > (float)kdone * g.c * 0.999f) ? kdone + 1 : kdone;
  const int xa = qc.cx - kE < 0 ? 0 : qc.cx - kE, xb = qc.cx + kE >= g.nx ? g.nx - 1 : qc.cx + kE;
  const int side = 2 * kE + 1, nrow = side * side;
  const float rcube = (float)kE * g.c * 0.999f;
  for (int attempt = 0; attempt < 3; ++attempt, radd *= 0.5f) {
    float Rt = d1 + radd;
    Rt = Rt < rcube ? Rt : rcube;
    const float R2 = Rt * Rt;
    if (lane == 0) *stage_n = 0;
    for (int u = lane; u < GS_FAR_SLOTS; u += G) stage[u] = ~0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int r = lane; r < nrow; r += G) {
      const int zz = qc.cz + r / side - kE, yy = qc.cy + r % side - kE;
      if (zz < 0 || zz >= g.nz || yy < 0 || yy >= g.ny) continue;
      const int row = (zz * g.ny + yy) * g.nx;
      const int je = cell_start[row + xb + 1];
      for (int j = cell_start[row + xa]; j < je; ++j) {
        const float4 c = sorted[j];
        const float dx = qx - c.x, dy = qy - c.y, dz = qz - c.z;
        float d = dx * dx;
        d = gs_fma(dy, dy, d);
        d = gs_fma(dz, dz, d);
        if (d < R2) {
          const int pos = atomicAdd(stage_n, 1);
          if (pos < GS_FAR_SLOTS) stage[pos] = (uint32_t)j;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int n = *stage_n;   // the same for all lanes of the group (same wave: LDS accesses are ordered)
    __builtin_amdgcn_wave_barrier();
    if (n <= GS_FAR_SLOTS) { R = Rt; break; }
  }
  return R;
}

// ---- candidate lists of ORDINARY queries (round 4).  The 2 x numiters searches of a solve look at the SAME binned
// targets from positions that move by millimetres, then by micrometres.  A search that scans a whole region (the
// un-pruned 2x2x2 block, or a cube of cells) lets every lane remember, besides the minimum, the KT nearest candidates it
// has seen and the distance of the next one (GlTop: a few compares and selects per candidate).  Behind the scan
//     list = the nearest candidates of the lanes (M slots of `sorted`),
//     R    = min(distance of the nearest candidate that did NOT make the list, bound of the scanned region):
// every target closer than R to the search position q0 is on the list.  A later search from q is EXACT on the list alone
// when
//     sqrt(best list distance) + |q - q0| < 0.9999 R:
// every target outside the list is at least R from q0, hence farther than R - |q - q0| from q, hence farther than the
// list's best; ties are inside the list, and the packed (distance bits, index) order is that of every other engine.
// The half-iteration kernels fetch the listed points while their prologue waits for the partial rows of the previous
// launch, so a launch whose lists all prove has no search on its critical path.  A list that fails is replaced where the
// point is now by the 16-lane scan that serves it (the same pass).  tools/icp_list_sim.py is the CPU study behind it.
constexpr int GL_SLOTS = 8;            // list slots of a source point in memory (32 bytes)
constexpr int GL_STAT_LAUNCHES = 64;   // launches of a solve with their own failure counters (diagnostics)
constexpr float GL_MIN_ROOM = 0.1f;    // cells between the neighbour and the bound of a cube scan, at least (else the next cube)
// entries per lane of a G-lane group (M = G x entries <= GL_SLOTS)
// (2 lanes x 2, 4 x 1, 8 x 1: one more entry per lane makes the first-half kernel spill next to its scalar stage --
// which wants ~60 registers for one lane's float64 code -- and any spill costs a dependent launch microseconds)
template <int G> constexpr int gl_k() { return G == 2 ? 2 : 1; }

// the KT nearest candidates a lane has seen (squared distances ascending, slots of `sorted`) and the squared distance of
// the next nearest one (d[KT], +inf while fewer were seen)
template <int KT>
struct GlTop {
  float d[KT + 1];
  int s[KT];
};
template <int KT>
GS_DEV void gl_top_reset(GlTop<KT>& t) {
#pragma unroll
  for (int k = 0; k <= KT; ++k) t.d[k] = __builtin_inff();
#pragma unroll
  for (int k = 0; k < KT; ++k) t.s[k] = -1;
}
// (branch-free insertion; a NaN distance fails every compare and changes nothing)
template <int KT>
GS_DEV void gl_top_push(GlTop<KT>& t, const float d, const int slot) {
#pragma unroll
  for (int k = KT; k >= 0; --k) {   // from the far end: entry k takes entry k - 1 when d goes in front of it
    const bool before_prev = k > 0 && d < t.d[k - 1];
    const bool before_this = d < t.d[k];
    if (k < KT) t.s[k] = before_prev ? t.s[k - 1] : (before_this ? slot : t.s[k]);
    t.d[k] = before_prev ? t.d[k - 1] : (before_this ? d : t.d[k]);
  }
}
template <int G>
GS_DEV float gl_group_minf(float v) {
#pragma unroll
  for (int d = G / 2; d > 0; d >>= 1) {
    const float o = __shfl_xor(v, d, G);
    v = o < v ? o : v;
  }
  return v;
}

// grid_search_stage0 without a search bound (the whole 2x2x2 block is scanned, so that it bounds the list) and with the
// lanes' trackers.  *rcov2 = squared distance within which the block holds every target (+inf: no cells behind any face).
template <int G, int KT, int NFI = (G >= 16 ? 1 : 4)>   // NFI: gathers in flight per lane
GS_DEV unsigned long long grid_search_stage0_top(const GsGrid& g, const int* __restrict__ cell_start,
                                                 const float4* __restrict__ sorted, float qx, float qy, float qz, int lane,
                                                 bool* resolved, int* win, GlTop<KT>& top, float* rcov2) {
  const GsQueryCell qc = grid_query_cell(g, qx, qy, qz);
  const float px = qc.px, py = qc.py, pz = qc.pz;
  const int cx = qc.cx, cy = qc.cy, cz = qc.cz;
  unsigned long long key = ~0ull;
  int bt = -1;
  const float fx = (px - g.ox) * g.inv_c - (float)cx, fy = (py - g.oy) * g.inv_c - (float)cy,
              fz = (pz - g.oz) * g.inv_c - (float)cz;
  const int x0 = (fx < 0.5f) ? cx - 1 : cx, y0 = (fy < 0.5f) ? cy - 1 : cy, z0 = (fz < 0.5f) ? cz - 1 : cz;
  const float BIG = 3.0e38f;
  const float ax = fminf(x0 >= 1 ? fx + (float)(cx - x0) : BIG, x0 + 2 < g.nx ? (float)(x0 + 2 - cx) - fx : BIG);
  const float ay = fminf(y0 >= 1 ? fy + (float)(cy - y0) : BIG, y0 + 2 < g.ny ? (float)(y0 + 2 - cy) - fy : BIG);
  const float az = fminf(z0 >= 1 ? fz + (float)(cz - z0) : BIG, z0 + 2 < g.nz ? (float)(z0 + 2 - cz) - fz : BIG);
  const float amin = fminf(ax, fminf(ay, az));
  int sb0 = 0, sb1 = 0, sb2 = 0, sb3 = 0, e1 = 0, e2 = 0, e3 = 0, total = 0;
  {
    const int xa = x0 >= 0 ? x0 : x0 + 1, xb = x0 + 1 < g.nx ? x0 + 1 : x0;
    const bool zl = z0 >= 0, zh = z0 + 1 < g.nz, yl = y0 >= 0, yh = y0 + 1 < g.ny;
    const int r0 = (z0 * g.ny + y0) * g.nx, r1 = r0 + g.nx, r2 = r0 + g.ny * g.nx, r3 = r2 + g.nx;
    int se0 = 0, se1 = 0, se2 = 0, se3 = 0;
    if (zl && yl) { sb0 = cell_start[r0 + xa]; se0 = cell_start[r0 + xb + 1]; }
    if (zl && yh) { sb1 = cell_start[r1 + xa]; se1 = cell_start[r1 + xb + 1]; }
    if (zh && yl) { sb2 = cell_start[r2 + xa]; se2 = cell_start[r2 + xb + 1]; }
    if (zh && yh) { sb3 = cell_start[r3 + xa]; se3 = cell_start[r3 + xb + 1]; }
    e1 = se0 - sb0;
    e2 = e1 + (se1 - sb1);
    e3 = e2 + (se2 - sb2);
    total = e3 + (se3 - sb3);
  }
  constexpr int NF = NFI;   // gathers in flight per lane (the wide groups of the re-search pass: one or two -- registers)
  for (int t0 = lane; t0 < total; t0 += NF * G) {
    float4 p[NF];
    int ix[NF];
#pragma unroll
    for (int u = 0; u < NF; ++u) {
      const int t = t0 + u * G, tt = t < total ? t : 0;  // total > 0 here: position 0 is valid
      ix[u] = tt < e1 ? sb0 + tt : (tt < e2 ? sb1 + (tt - e1) : (tt < e3 ? sb2 + (tt - e2) : sb3 + (tt - e3)));
      p[u] = sorted[ix[u]];
    }
#pragma unroll
    for (int u = 0; u < NF; ++u) {
      const bool in = t0 + u * G < total;
      const unsigned long long k2 = in ? grid_key(qx, qy, qz, p[u]) : ~0ull;
      const bool better = k2 < key;
      key = better ? k2 : key;
      bt = better ? ix[u] : bt;
      // (the distance the key carries; NaN for a masked or NaN candidate: ignored by the tracker)
      gl_top_push<KT>(top, __uint_as_float((uint32_t)(k2 >> 32)), ix[u]);
    }
  }
  {
    const unsigned long long kmin = grid_group_min<G>(key);
    *win = (key == kmin && bt >= 0) ? bt : -1;
    key = kmin;
  }
  const float rb = (amin < 1.0e30f) ? (amin - 0.001f) * g.c : BIG;
  const float bd = __uint_as_float((uint32_t)(key >> 32));  // NaN while nothing was found
  *resolved = rb > 0.0f && (rb >= 1.0e30f ? bd == bd : bd <= rb * rb);
  *rcov2 = rb >= 1.0e30f ? __builtin_inff() : (rb > 0.0f ? rb * rb : 0.0f);
  return key;
}

// grid_search_rings with the lanes' trackers (reset per cube: a cube re-scans the cells of the one before).  A cube that
// proves the neighbour but leaves it less than GL_MIN_ROOM cells of room is followed by the next one, for the list's
// sake.  *rcov2 = squared radius within which the last cube scanned holds every target (0: not resolved).
template <int G, int KT>
GS_DEV unsigned long long grid_search_rings_top(const GsGrid& g, const int* __restrict__ cell_start,
                                                const float4* __restrict__ sorted, float qx, float qy, float qz, int lane,
                                                unsigned long long key, bool* resolved, int* win, const int kmax,
                                                GlTop<KT>& top, float* rcov2) {
  const GsQueryCell qc = grid_query_cell(g, qx, qy, qz);
  bool done = false;
  int bs = -1;
  float cov2 = 0.0f;
  for (int k = 1; k <= kmax + 1; ++k) {
    if (k > kmax && !done) break;
    gl_top_reset<KT>(top);
    const int side = 2 * k + 1, nrow = side * side;
    const int xa = qc.cx - k < 0 ? 0 : qc.cx - k, xb = qc.cx + k >= g.nx ? g.nx - 1 : qc.cx + k;
    for (int r = lane; r < nrow; r += G) {
      const int zz = qc.cz + r / side - k, yy = qc.cy + r % side - k;
      if (zz < 0 || zz >= g.nz || yy < 0 || yy >= g.ny) continue;
      const int row = (zz * g.ny + yy) * g.nx;
      const int je = cell_start[row + xb + 1];
      // (GL_RING_NF gathers in flight per lane: one candidate per round trip made a cube scan 7 - 16 us in a mature map, and a
      // launch is as slow as its slowest block -- profiles/r05_c_failing_lookahead_timeline.txt)
      for (int j = cell_start[row + xa]; j < je; j += GL_RING_NF) {
        float4 p[GL_RING_NF];
#pragma unroll
        for (int u = 0; u < GL_RING_NF; ++u) p[u] = sorted[j + u < je ? j + u : je - 1];
#pragma unroll
        for (int u = 0; u < GL_RING_NF; ++u) {
          const unsigned long long k2 = j + u < je ? grid_key(qx, qy, qz, p[u]) : ~0ull;
          if (k2 < key) { key = k2; bs = j + u; }
          gl_top_push<KT>(top, __uint_as_float((uint32_t)(k2 >> 32)), j + u);   // (NaN bits of a masked entry: ignored)
        }
      }
    }
    {
      const unsigned long long own = key;
      key = grid_group_min<G>(key);
      if (own != key) bs = -1;
    }
    const float rb = (float)k * g.c * 0.999f;
    const float bd = __uint_as_float((uint32_t)(key >> 32));  // NaN while nothing was found
    if (done) { cov2 = rb * rb; break; }                      // the extra cube: room for the list
    done = bd <= rb * rb;
    if (done) {
      cov2 = rb * rb;
      if (sqrtf(bd) + GL_MIN_ROOM * g.c <= rb) break;
    }
  }
  *resolved = done;
  *win = bs;
  *rcov2 = done ? cov2 : 0.0f;
  return key;
}

// The list of a point searched by its own G lanes (the readers: lane l keeps entries l * KT ..): every lane writes the
// candidates it remembers; R from what the lanes left out and the bound of the scan.  All lanes of the group call it.
template <int G, int KT>
GS_DEV void gl_write_lanes(const GlTop<KT>& top, const float rcov2, float qx, float qy, float qz, int lane,
                           uint32_t* __restrict__ slots, float4* __restrict__ lq) {
  const float out2 = gl_group_minf<G>(top.d[KT]);
  const float R2 = out2 < rcov2 ? out2 : rcov2;
#pragma unroll
  for (int k = 0; k < KT; ++k) slots[lane * KT + k] = top.s[k] >= 0 ? (uint32_t)top.s[k] : ~0u;
  if (lane == 0) *lq = make_float4(qx, qy, qz, sqrtf(R2));
}

// The list of a point searched by a group of GB lanes for readers that keep M slots: the M nearest of everything the
// lanes remember (M rounds of a group minimum; the winner hands over its head), R from the nearest candidate left out -- a
// lane's own next one or the best head that stays behind -- and the bound of the scan.  All lanes of the group call it.
template <int GB, int KT>
GS_DEV void gl_select_write(GlTop<KT> top, const float rcov2, float qx, float qy, float qz, int lane, const int M,
                            uint32_t* __restrict__ slots, float4* __restrict__ lq) {
  float out2 = gl_group_minf<GB>(top.d[KT]);
  for (int e = 0; e < GL_SLOTS; ++e) {
    if (e > M) {   // (M is the same for every lane of the kernel) nothing more to look at: clear the slot
      if (lane == 0) slots[e] = ~0u;
      continue;
    }
    const unsigned long long mine = ((unsigned long long)__float_as_uint(top.d[0]) << 32) | (unsigned)lane;  // d >= 0
    const unsigned long long best = grid_group_min<GB>(mine);
    const float bd = __uint_as_float((uint32_t)(best >> 32));
    if (e >= M) {   // the nearest one that stays behind bounds the list (and the unused slots are cleared)
      if (e == M) out2 = bd < out2 ? bd : out2;
      if (lane == 0) slots[e] = ~0u;
      continue;
    }
    if (best == mine) {
      slots[e] = (bd < __builtin_inff() && top.s[0] >= 0) ? (uint32_t)top.s[0] : ~0u;
#pragma unroll
      for (int k = 0; k + 1 < KT; ++k) { top.d[k] = top.d[k + 1]; top.s[k] = top.s[k + 1]; }
      top.d[KT - 1] = __builtin_inff();   // (what came behind the remembered ones is in out2 already)
      top.s[KT - 1] = -1;
    }
  }
  if (M >= GL_SLOTS) {   // no slot left to look at the next head in the loop
    const float nb = gl_group_minf<GB>(top.d[0]);
    out2 = nb < out2 ? nb : out2;
  }
  const float R2 = out2 < rcov2 ? out2 : rcov2;
  if (lane == 0) *lq = make_float4(qx, qy, qz, sqrtf(R2));
}

template <int G = GQ_G>
GS_DEV unsigned long long grid_search_group(const GsGrid& g, const int* __restrict__ cell_start,
                                        const float4* __restrict__ sorted, float qx, float qy, float qz, int lane,
                                        bool* resolved) {
  int win;
  unsigned long long key = grid_search_stage0<G>(g, cell_start, sorted, qx, qy, qz, lane, resolved, &win);
  if (!*resolved) key = grid_search_rings<G>(g, cell_start, sorted, qx, qy, qz, lane, key, resolved, &win);
  return key;
}

// Block-wide brute force over the BINNED targets (sorted[0 .. n) = every target that passed the filter, as (x, y, z,
// index bits)): what the fused ICP kernels fall back to for queries the cube scans leave open -- with a target filter
// the raw array holds the whole map, the binned copy only the targets.  Same distance arithmetic and key order as the
// grid search and the brute-force engine, hence the same result.
// Up to BQ queries are served in ONE pass over the binned targets (every thread of the block calls it with the same
// arguments).  Far queries come in clusters (a frame border that looks at surface the map has not seen lands in one or
// two blocks), and a pass per query made such launches 5x longer; a pass serves BQ of them for the price of one.
// ids[0 .. nq): slots of the queries in qs / key_out / bslot_out (LDS arrays of the caller).
template <int BLOCK, int BQ>
GS_DEV void block_brute_min_sorted_multi(const float (*qs)[3], const int* ids, int nq,
                                         const float4* __restrict__ sorted, int n, unsigned long long* key_out,
                                         int* bslot_out, const bool code_global = false) {
  __shared__ unsigned long long red_m[BLOCK / GS_WAVE][BQ];
  float q[BQ][3];
  unsigned long long key[BQ];
  int bs[BQ];
#pragma unroll
  for (int i = 0; i < BQ; ++i) {
    const int id = ids[i < nq ? i : 0];
    q[i][0] = qs[id][0]; q[i][1] = qs[id][1]; q[i][2] = qs[id][2];
    key[i] = ~0ull;
    bs[i] = -1;
  }
  for (int j = threadIdx.x; j < n; j += BLOCK) {
    const float4 p = sorted[j];
#pragma unroll
    for (int i = 0; i < BQ; ++i) {
      const unsigned long long k2 = grid_key(q[i][0], q[i][1], q[i][2], p);
      if (k2 < key[i]) { key[i] = k2; bs[i] = j; }
    }
  }
#pragma unroll
  for (int i = 0; i < BQ; ++i) {
    unsigned long long k = key[i];
#pragma unroll
    for (int d = GS_WAVE / 2; d > 0; d >>= 1) {
      const unsigned long long o = __shfl_xor(k, d, GS_WAVE);
      k = o < k ? o : k;
    }
    if ((threadIdx.x & (GS_WAVE - 1)) == 0) red_m[threadIdx.x / GS_WAVE][i] = k;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < BQ; ++i) {
    if (i < nq) {
      unsigned long long k = red_m[0][i];
#pragma unroll
      for (int w = 1; w < BLOCK / GS_WAVE; ++w) k = red_m[w][i] < k ? red_m[w][i] : k;
      const int id = ids[i];
      // the one thread that holds the winning candidate (code_global: the tile engine's code of a global slot)
      if (key[i] == k && bs[i] >= 0) bslot_out[id] = code_global ? -2 - bs[i] : bs[i];
      if (threadIdx.x == 0) key_out[id] = k;
    }
  }
  __syncthreads();
}

// ---- WIDE lists of hard queries (round 5).  A source point several cells from every target -- a frame border that looks
// at a surface under a grazing angle: neighbouring lattice pixels are 15 - 30 cm apart there, profiles/r05_b_late_frames.txt
// -- is served by the cube scans (~9 us for a 16-lane group) or by a pass of its whole block over all binned targets
// (~36 us: one CU pulling a megabyte), in EVERY launch of a solve, and the launch is as slow as that block.  Its ordinary
// list (4 slots) rarely proves anything: a far query sees a line of targets at nearly the same distance.  So whatever
// serves such a point also leaves a WIDE list: GS_FAR_SLOTS slots of `sorted`, four per lane of the 16-lane group that
// checks it first thing in the left-over pass (far_list_search; the exactness argument is that of every list: every target
// within R of the position q0 the list was made at is on it).
//   * from a cube scan: everything the 16 lanes remember (2 candidates each) that lies within R = min(nearest candidate a
//     lane dropped, bound of the cube);
//   * from the block-wide pass: every thread remembers its nearest candidate and the distance of its second nearest;
//     R = the smallest of those second distances over the block -- a target within R that is not its thread's nearest
//     would make that thread's second distance smaller than R -- and the list = the threads' nearest candidates within R
//     (with 768 threads the first two of the globally nearest that share a thread are ~35 apart: the list holds the ~35
//     nearest targets; nested smaller radii in case it does not fit).
// lane-major layout: lane l of the checking group reads slots 4 l .. 4 l + 3.
template <int GB, int KT>
GS_DEV void far_write_from_top(const GlTop<KT>& top, const float rcov2, float qx, float qy, float qz, int lane,
                               uint32_t* __restrict__ slots, float4* __restrict__ cq) {
  static_assert(GS_FAR_SLOTS == 4 * GB && KT <= 4, "four slots per lane of the checking group");
  const float out2 = gl_group_minf<GB>(top.d[KT]);
  const float R2 = out2 < rcov2 ? out2 : rcov2;
  static_assert(KT == 2, "two remembered candidates per lane");
  const uint32_t w0 = (top.s[0] >= 0 && top.d[0] < R2) ? (uint32_t)top.s[0] : ~0u;
  const uint32_t w1 = (top.s[1] >= 0 && top.d[1] < R2) ? (uint32_t)top.s[1] : ~0u;
  reinterpret_cast<uint4*>(slots)[lane] = make_uint4(w0, w1, ~0u, ~0u);
  if (lane == 0) *cq = make_float4(qx, qy, qz, R2 > 0.0f && R2 < 3.0e38f ? sqrtf(R2) : 0.0f);
}

// block_brute_min_sorted_multi that also leaves the wide list of every query (see above).  gbase + id = the source point
// of the query in slot id.  Every thread of the block calls it with the same arguments.
template <int BLOCK, int BQ>
GS_DEV void block_brute_min_list_multi(const float (*qs)[3], const int* ids, int nq, const float4* __restrict__ sorted,
                                       int n, unsigned long long* key_out, int* bslot_out, const int64_t gbase,
                                       float4* __restrict__ far_cq, uint32_t* __restrict__ far_c) {
  __shared__ unsigned long long red_m[BLOCK / GS_WAVE][BQ];
  __shared__ float red_d[BLOCK / GS_WAVE][BQ];
  __shared__ uint32_t lst[3][BQ][GS_FAR_SLOTS];
  __shared__ int cnt[3][BQ];
  float q[BQ][3], d2[BQ];
  unsigned long long key[BQ];
  int bs[BQ];
#pragma unroll
  for (int i = 0; i < BQ; ++i) {
    const int id = ids[i < nq ? i : 0];
    q[i][0] = qs[id][0]; q[i][1] = qs[id][1]; q[i][2] = qs[id][2];
    key[i] = ~0ull;
    bs[i] = -1;
    d2[i] = __builtin_inff();
  }
  if (threadIdx.x < 3 * BQ) cnt[threadIdx.x / BQ][threadIdx.x % BQ] = 0;
  for (int t = threadIdx.x; t < 3 * BQ * GS_FAR_SLOTS; t += BLOCK) (&lst[0][0][0])[t] = ~0u;
  for (int j = threadIdx.x; j < n; j += BLOCK) {
    const float4 p = sorted[j];
#pragma unroll
    for (int i = 0; i < BQ; ++i) {
      const unsigned long long k2 = grid_key(q[i][0], q[i][1], q[i][2], p);
      const bool better = k2 < key[i];
      // the distance that does NOT become (or stay) the thread's nearest: NaN bits (no candidate yet, NaN distance) lose
      const float od = __uint_as_float((uint32_t)((better ? key[i] : k2) >> 32));
      d2[i] = od < d2[i] ? od : d2[i];
      if (better) { key[i] = k2; bs[i] = j; }
    }
  }
#pragma unroll
  for (int i = 0; i < BQ; ++i) {
    unsigned long long k = key[i];
    float m = d2[i];
#pragma unroll
    for (int d = GS_WAVE / 2; d > 0; d >>= 1) {
      const unsigned long long o = __shfl_xor(k, d, GS_WAVE);
      k = o < k ? o : k;
      const float om = __shfl_xor(m, d, GS_WAVE);
      m = om < m ? om : m;
    }
    if ((threadIdx.x & (GS_WAVE - 1)) == 0) { red_m[threadIdx.x / GS_WAVE][i] = k; red_d[threadIdx.x / GS_WAVE][i] = m; }
  }
  __syncthreads();
  float Ra[BQ], Rb[BQ], Rc[BQ];   // nested radii, squared
#pragma unroll
  for (int i = 0; i < BQ; ++i) {
    unsigned long long k = red_m[0][i];
    float m = red_d[0][i];
#pragma unroll
    for (int w = 1; w < BLOCK / GS_WAVE; ++w) {
      k = red_m[w][i] < k ? red_m[w][i] : k;
      m = red_d[w][i] < m ? red_d[w][i] : m;
    }
    const float bd = __uint_as_float((uint32_t)(k >> 32));   // NaN: nothing found
    const float d1 = sqrtf(bd), Rm = sqrtf(m);               // (m = +inf: fewer targets than threads -- every one is remembered)
    const bool open = i < nq && bd == bd && m > bd;
    const float room = (m < 3.0e38f ? Rm : d1 + 1.0f) - d1;  // (all targets remembered: any radius is covered; one metre)
    const float tb = d1 + 0.5f * room, tc = d1 + 0.25f * room;
    Ra[i] = open ? (m < 3.0e38f ? m : (d1 + 1.0f) * (d1 + 1.0f)) : -1.0f;
    Rb[i] = open ? tb * tb : -1.0f;
    Rc[i] = open ? tc * tc : -1.0f;
    if (i < nq) {
      const int id = ids[i];
      if (key[i] == k && bs[i] >= 0) bslot_out[id] = bs[i];   // the one thread that holds the winning candidate
      if (threadIdx.x == 0) key_out[id] = k;
    }
    // this thread's nearest candidate goes on the lists whose radius holds it
    const float md = __uint_as_float((uint32_t)(key[i] >> 32));
    if (bs[i] >= 0 && md < Ra[i]) {
      const int pa = atomicAdd(&cnt[0][i], 1);
      if (pa < GS_FAR_SLOTS) lst[0][i][pa] = (uint32_t)bs[i];
      if (md < Rb[i]) {
        const int pb = atomicAdd(&cnt[1][i], 1);
        if (pb < GS_FAR_SLOTS) lst[1][i][pb] = (uint32_t)bs[i];
        if (md < Rc[i]) {
          const int pc = atomicAdd(&cnt[2][i], 1);
          if (pc < GS_FAR_SLOTS) lst[2][i][pc] = (uint32_t)bs[i];
        }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < BQ; ++i) {
    if (i < nq) {   // block-uniform
      const int r = cnt[0][i] <= GS_FAR_SLOTS ? 0 : (cnt[1][i] <= GS_FAR_SLOTS ? 1 : (cnt[2][i] <= GS_FAR_SLOTS ? 2 : -1));
      const bool ok = r >= 0 && Ra[i] > 0.0f;
      const int64_t sq = gbase + ids[i];
      if (ok)
        for (int t = threadIdx.x; t < GS_FAR_SLOTS; t += BLOCK)
          far_c[sq * GS_FAR_SLOTS + t] = r == 0 ? lst[0][i][t] : (r == 1 ? lst[1][i][t] : lst[2][i][t]);
      if (threadIdx.x == 0) {
        // (the radius again from its definition -- the reductions are still in LDS: selecting among the register arrays
        // by r would put them in scratch; d1 + room is sqrt(Ra) up to rounding, which the 0.01 % of the proof covers)
        unsigned long long k = red_m[0][i];
        float m = red_d[0][i];
        for (int w = 1; w < BLOCK / GS_WAVE; ++w) {
          k = red_m[w][i] < k ? red_m[w][i] : k;
          m = red_d[w][i] < m ? red_d[w][i] : m;
        }
        const float d1 = sqrtf(__uint_as_float((uint32_t)(k >> 32)));
        const float room = (m < 3.0e38f ? sqrtf(m) : d1 + 1.0f) - d1;
        const float Rsel = d1 + (r == 0 ? 1.0f : (r == 1 ? 0.5f : 0.25f)) * room;
        far_cq[sq] = make_float4(q[i][0], q[i][1], q[i][2], ok ? Rsel : 0.0f);
      }
    }
  }
  __syncthreads();
}

// Lists for queries only a pass over all binned targets serves (ids: their slots in the LDS arrays of the caller,
// d1_in[id] = distance of the neighbour the search found): ONE more pass collects, per query, every target within
// d1 + radd0, d1 + radd0 / 2 and d1 + radd0 / 4 into three lists (the radii are nested; a far query facing a surface
// has many targets at nearly the same distance, and how many is not known beforehand); the largest radius whose list
// fits wins.  Writes far_cq[gidx[id]] = (query, R or 0), the GS_FAR_SLOTS slots of far_c[gidx[id]] and flag_out[id] =
// the list is valid.  Every thread of the block calls it with the same arguments.
template <int BLOCK, int BQ>
GS_DEV void block_brute_collect_multi(const float (*qs)[3], const int* ids, int nq, const float4* __restrict__ sorted,
                                      int n, const float* d1_in, const float radd0, const int* gidx,
                                      float4* __restrict__ far_cq, uint32_t* __restrict__ far_c, uint8_t* flag_out) {
  __shared__ uint32_t lst[3][BQ][GS_FAR_SLOTS];
  __shared__ int cnt[3][BQ];
  float q[BQ][3], Ra[BQ], Rb[BQ], Rc[BQ];   // the three radii, squared (-1: no list for this query)
#pragma unroll
  for (int i = 0; i < BQ; ++i) {
    const int id = ids[i < nq ? i : 0];
    q[i][0] = qs[id][0]; q[i][1] = qs[id][1]; q[i][2] = qs[id][2];
    const float d1 = d1_in[id];
    const bool open = i < nq && d1 == d1;   // (NaN: nothing was found, no list)
    const float ta = d1 + radd0, tb = d1 + 0.5f * radd0, tc = d1 + 0.25f * radd0;
    Ra[i] = open ? ta * ta : -1.0f;
    Rb[i] = open ? tb * tb : -1.0f;
    Rc[i] = open ? tc * tc : -1.0f;
  }
  if (threadIdx.x < 3 * BQ) cnt[threadIdx.x / BQ][threadIdx.x % BQ] = 0;
  for (int t = threadIdx.x; t < 3 * BQ * GS_FAR_SLOTS; t += BLOCK) (&lst[0][0][0])[t] = ~0u;
  __syncthreads();
  for (int j = threadIdx.x; j < n; j += BLOCK) {
    const float4 p = sorted[j];
#pragma unroll
    for (int i = 0; i < BQ; ++i) {
      const float dx = q[i][0] - p.x, dy = q[i][1] - p.y, dz = q[i][2] - p.z;
      float d = dx * dx;
      d = gs_fma(dy, dy, d);
      d = gs_fma(dz, dz, d);
      if (d < Ra[i]) {
        const int pa = atomicAdd(&cnt[0][i], 1);
        if (pa < GS_FAR_SLOTS) lst[0][i][pa] = (uint32_t)j;
        if (d < Rb[i]) {
          const int pb = atomicAdd(&cnt[1][i], 1);
          if (pb < GS_FAR_SLOTS) lst[1][i][pb] = (uint32_t)j;
          if (d < Rc[i]) {
            const int pc = atomicAdd(&cnt[2][i], 1);
            if (pc < GS_FAR_SLOTS) lst[2][i][pc] = (uint32_t)j;
          }
        }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < BQ; ++i) {
    if (i < nq) {   // block-uniform
      const int r = cnt[0][i] <= GS_FAR_SLOTS ? 0 : (cnt[1][i] <= GS_FAR_SLOTS ? 1 : (cnt[2][i] <= GS_FAR_SLOTS ? 2 : -1));
      const bool ok = r >= 0 && Ra[i] > 0.0f;
      if (ok)
        for (int t = threadIdx.x; t < GS_FAR_SLOTS; t += BLOCK)
          far_c[(int64_t)gidx[ids[i]] * GS_FAR_SLOTS + t] = r == 0 ? lst[0][i][t] : (r == 1 ? lst[1][i][t] : lst[2][i][t]);
      if (threadIdx.x == 0) {
        // (the radius again from its definition: selecting among the register arrays by r would put them in scratch)
        const float Rsel = d1_in[ids[i]] + (r == 0 ? 1.0f : (r == 1 ? 0.5f : 0.25f)) * radd0;
        far_cq[gidx[ids[i]]] = make_float4(q[i][0], q[i][1], q[i][2], ok ? Rsel : 0.0f);
        flag_out[ids[i]] = ok ? 1 : 0;   // (LDS of the caller)
      }
    }
  }
  __syncthreads();
}

