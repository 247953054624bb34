// gs_icp_tile.h — the lattice engine of the device-resident LM loop (included by gs_icp_loop.hip only).
//
// The ICP source of the SLAM loop (slam/icpslam.py:238-242, odometry/icputils.py:623-669) is the live frame's
// [::ds, ::ds] pixel LATTICE, and the 2 x numiters exact 1-NN searches of a solve (odometry/icputils.py:200) all run
// against the SAME binned target set with queries that move by a fraction of a grid cell.  So a workgroup owns a
// 2-D tile of IT_TW x IT_TH lattice slots -- a compact patch of surface -- and everything its searches can touch is a
// small box of grid cells that is known before the first search:
//
//   gs_it_slab_build_kernel (once per frame): per tile, the bounding box of its queries' cells at the initial pose,
//     grown by IT_MARGIN cells, and a private copy ("slab") of that box of the global grid: the cell table rebased to
//     16-bit offsets + the binned target points and normals of those cells, contiguous in memory.
//   gs_icp_tile_half_kernel (2 x numiters per frame): the slab is copied into LDS by coalesced loads that are in
//     flight while the prologue (row sums + scalar stage of the previous half-iteration) runs; the 2x2x2 stage of the
//     search, the match and the Gauss-Newton row then read LDS only.  No dependent global gathers are left on the
//     critical path of a launch (the row-unit kernel spent 6.8 of its 17.4 us there at 8 sequences per GPU).
//
// Exactness: the slab is a verbatim sub-box of the global grid, and the search applies the global engine's bound
// (grid_search_stage0) to the same cells and candidates, so every query gets the brute-force neighbour.  A query
// whose 2x2x2 block leaves the tile's box (it moved more than IT_MARGIN - 1 cells), or whose tile has no slab (box or
// target count above the LDS budget), is served from the global arrays by the same code path as before.
// Sums: one partial row per TILE (fixed order inside the tile: 24 groups of 16 queries, then the 24 sub-sums in
// order); the next launch adds the tile rows in tile order.  The result does not depend on the batch size or on
// which block runs which tile.
#pragma once
#include <stddef.h>

constexpr int IT_TW = 16, IT_TH = 24;        // lattice tile of one workgroup
constexpr int IT_NQ = IT_TW * IT_TH;         // 384 queries
constexpr int IT_G = 2;                      // lanes per query
constexpr int IT_BLOCK = IT_NQ * IT_G;       // 768 threads, 2 workgroups per CU
constexpr int IT_PTS_CAP = 2560;             // binned targets a slab can hold (LDS: 16 bytes each, the points; the normals stay in global memory)
constexpr int IT_CELLS_CAP = 16384;          // entries of a slab's cell table (cells + 1 end sentinel; global memory only)
constexpr int IT_ROWS_CAP = 2048;            // (z, y) rows of a slab's box
constexpr int IT_MARGIN = 2;                 // cells added around the tile's query cells (1 for the 2x2x2 block + 1 of motion)
constexpr int IT_RG = 24, IT_RPG = IT_NQ / IT_RG;  // row groups of the tile sum x queries per group
constexpr int IT_PTS_EARLY = 1024;           // slab slots requested before the slab header has arrived
constexpr int IT_LDS_CELLS = IT_NQ * 8 * 2;   // cell-table entries that fit the LDS space of the Gauss-Newton rows (12 KB)
constexpr int IT_HG = 16;                    // lanes per query of the leftover searches
constexpr int IT_LOCAL_RINGS = 3;            // cube radius the leftover searches go to on the slab before the global grid
static_assert(IT_RG * LIN_NV <= IT_BLOCK && IT_NQ % IT_RG == 0, "tile sum shape");
static_assert(IT_CELLS_CAP * 2 % 64 == 0 && IT_PTS_CAP % GS_WAVE == 0 && IT_PTS_CAP < 0xffff, "slab shape");

struct ItSlabHdr {   // 64 bytes in front of every slab
  int mode;          // 0: no query in the tile, 1: slab valid, 2: no slab (searches use the global arrays)
  int bx0, by0, bz0; // the box in global cell coordinates
  int nbx, nby, nbz;
  int ncell, npts;
  int pad[7];
};
constexpr size_t IT_OFF_CELLS = 64;
constexpr size_t IT_OFF_PTS = IT_OFF_CELLS + 2 * (size_t)IT_CELLS_CAP;
constexpr size_t IT_OFF_NRM = IT_OFF_PTS + 16 * (size_t)IT_PTS_CAP;
constexpr size_t IT_SLAB_BYTES = IT_OFF_NRM + 16 * (size_t)IT_PTS_CAP;
static_assert(sizeof(ItSlabHdr) == 64 && IT_SLAB_BYTES % 64 == 0, "slab layout");

static inline int it_tiles_x(int Wl) { return (Wl + IT_TW - 1) / IT_TW; }
static inline int it_tiles(int Hl, int Wl) { return it_tiles_x(Wl) * ((Hl + IT_TH - 1) / IT_TH); }
static inline size_t it_slab_bytes(int Hl, int Wl) { return gs_align(IT_SLAB_BYTES * (size_t)it_tiles(Hl, Wl)); }

// ---------------------------------------------------------------- slab build ------------
struct ItBuildSeq {
  const float* lattice;      // [Hl * Wl][3] queries at the initial pose (NaN = empty slot)
  const GsGrid* gp;
  const int* cell_start;
  const float4* sorted;
  const float4* sorted_n;
  char* slabs;
};
struct ItBuildBatch {
  int B, Wl, Hl, tiles_x;
  ItBuildSeq s[GS_MAX_BATCH];
};

__global__ void __launch_bounds__(IT_NQ) gs_it_slab_build_kernel(const ItBuildBatch bb) {
  const ItBuildSeq& q = bb.s[blockIdx.x % bb.B];
  const int tile = blockIdx.x / bb.B;
  char* slab = q.slabs + IT_SLAB_BYTES * (size_t)tile;
  ItSlabHdr* hdr = reinterpret_cast<ItSlabHdr*>(slab);
  uint16_t* cells = reinterpret_cast<uint16_t*>(slab + IT_OFF_CELLS);
  float4* pts = reinterpret_cast<float4*>(slab + IT_OFF_PTS);
  float4* nrm = reinterpret_cast<float4*>(slab + IT_OFF_NRM);

  __shared__ int red[6][IT_NQ / GS_WAVE];
  __shared__ int rowbase[IT_ROWS_CAP + 1];   // first slab slot of every (z, y) row of the box
  __shared__ int rowsrc[IT_ROWS_CAP];        // global slot of the first target of the row
  __shared__ int scan_s[IT_NQ / GS_WAVE + 1];
  __shared__ int box_s[8];

  const GsGrid g = *q.gp;
  const int lx = (tile % bb.tiles_x) * IT_TW + (int)threadIdx.x % IT_TW;
  const int ly = (tile / bb.tiles_x) * IT_TH + (int)threadIdx.x / IT_TW;
  int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {-1, -1, -1};
  if (lx < bb.Wl && ly < bb.Hl) {
    const int64_t s = (int64_t)ly * bb.Wl + lx;
    const float p0 = q.lattice[3 * s], p1 = q.lattice[3 * s + 1], p2 = q.lattice[3 * s + 2];
    if (p0 == p0) {
      const GsQueryCell c = grid_query_cell(g, p0, p1, p2);
      lo[0] = hi[0] = c.cx; lo[1] = hi[1] = c.cy; lo[2] = hi[2] = c.cz;
    }
  }
  const int lane = threadIdx.x & (GS_WAVE - 1), wave = threadIdx.x / GS_WAVE;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    int a = lo[k], b = hi[k];
#pragma unroll
    for (int d = GS_WAVE / 2; d > 0; d >>= 1) {
      const int a2 = __shfl_down(a, d, GS_WAVE), b2 = __shfl_down(b, d, GS_WAVE);
      a = a2 < a ? a2 : a;
      b = b2 > b ? b2 : b;
    }
    if (lane == 0) { red[k][wave] = a; red[3 + k][wave] = b; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int a[3], b[3];
    for (int k = 0; k < 3; ++k) {
      a[k] = red[k][0]; b[k] = red[3 + k][0];
      for (int w = 1; w < IT_NQ / GS_WAVE; ++w) {
        a[k] = red[k][w] < a[k] ? red[k][w] : a[k];
        b[k] = red[3 + k][w] > b[k] ? red[3 + k][w] : b[k];
      }
    }
    const int n[3] = {g.nx, g.ny, g.nz};
    int mode = b[0] < 0 ? 0 : 1;
    for (int k = 0; k < 3; ++k) {
      a[k] = a[k] - IT_MARGIN < 0 ? 0 : a[k] - IT_MARGIN;
      b[k] = b[k] + IT_MARGIN >= n[k] ? n[k] - 1 : b[k] + IT_MARGIN;
      box_s[k] = a[k];
      box_s[3 + k] = mode ? b[k] - a[k] + 1 : 0;
    }
    if (mode && ((long long)box_s[3] * box_s[4] * box_s[5] + 1 > IT_CELLS_CAP || box_s[4] * box_s[5] > IT_ROWS_CAP)) mode = 2;
    box_s[6] = mode;
  }
  __syncthreads();
  const int bx0 = box_s[0], by0 = box_s[1], bz0 = box_s[2], nbx = box_s[3], nby = box_s[4], nbz = box_s[5];
  int mode = box_s[6];
  if (mode != 1) {
    if (threadIdx.x == 0) {
      ItSlabHdr h = {mode, bx0, by0, bz0, nbx, nby, nbz, 0, 0, {0, 0, 0, 0, 0, 0, 0}};
      *hdr = h;
    }
    return;
  }
  // rows of the box: row r = (z - bz0) * nby + (y - by0) is the global slot range [cell_start[row + bx0],
  // cell_start[row + bx0 + nbx)); thread t takes the contiguous rows [t * per, (t + 1) * per)
  const int nrows = nby * nbz, ncell = nrows * nbx;
  const int per = (nrows + IT_NQ - 1) / IT_NQ;
  int mine = 0;
  for (int k = 0; k < per; ++k) {
    const int r = (int)threadIdx.x * per + k;
    if (r < nrows) {
      const int grow = ((bz0 + r / nby) * g.ny + (by0 + r % nby)) * g.nx + bx0;
      const int b0 = q.cell_start[grow], e0 = q.cell_start[grow + nbx];
      rowsrc[r] = b0;
      rowbase[r] = e0 - b0;   // count for now
      mine += e0 - b0;
    }
  }
  int total;
  int run = gs_block_excl_scan<IT_NQ>(mine, scan_s, &total);
  for (int k = 0; k < per; ++k) {
    const int r = (int)threadIdx.x * per + k;
    if (r < nrows) {
      const int c = rowbase[r];
      rowbase[r] = run;
      run += c;
    }
  }
  if (threadIdx.x == 0) rowbase[nrows] = total;
  __syncthreads();
  if (total > IT_PTS_CAP) mode = 2;
  if (threadIdx.x == 0) {
    ItSlabHdr h = {mode, bx0, by0, bz0, nbx, nby, nbz, ncell, mode == 1 ? total : 0, {0, 0, 0, 0, 0, 0, 0}};
    *hdr = h;
  }
  if (mode != 1) return;
  // cell table: slab slot of the first target of every cell of the box (+ end sentinel)
  for (int lc = threadIdx.x; lc <= ncell; lc += IT_NQ) {
    int v = total;
    if (lc < ncell) {
      const int r = lc / nbx, x = lc % nbx;
      const int grow = ((bz0 + r / nby) * g.ny + (by0 + r % nby)) * g.nx + bx0;
      v = rowbase[r] + (q.cell_start[grow + x] - rowsrc[r]);
    }
    cells[lc] = (uint16_t)v;
  }
  // targets: slab slot i belongs to the row r with rowbase[r] <= i < rowbase[r + 1] (binary search over the rows)
  for (int i = threadIdx.x; i < total; i += IT_NQ) {
    int a = 0, b = nrows;  // invariant: rowbase[a] <= i < rowbase[b]
    while (b - a > 1) {
      const int m = (a + b) >> 1;
      if (rowbase[m] <= i) a = m; else b = m;
    }
    const int src = rowsrc[a] + (i - rowbase[a]);
    pts[i] = q.sorted[src];
    nrm[i] = q.sorted_n[src];
  }
}

// ---------------------------------------------------------------- half-iteration ---------
struct ItSeq {
  const float* src_in;       // lattice [Hl * Wl][3]
  float* src_out;
  const float* tgt;          // map rows (only for the degenerate "no target at all" row)
  const float* tn;
  const GsGrid* gp;
  const int* cell_start;
  const float4* sorted;
  const float4* sorted_n;
  const char* slabs;
  float* d2prev;             // tiles without a slab: squared distance of every query's previous neighbour (search bound)
  uint32_t* cand;            // [n_lat][4] candidate lists: 2 lanes x 4 slab slots (16 bit each, 0xffff = none)
  float4* cq;                // [n_lat] (query position the list was built at, exactness radius R; R <= 0: no list)
  float4* cn;                // [n_lat] normal of the previous match (x, y, z, slab slot bits)
  const double* partials_in;
  double* partials_out;
  const IcpSmall* st_in;
  IcpSmall* st_out;
  float* trace;
};
struct ItBatch {
  int B, Wl, Hl, tiles_x, ntiles;
  int force_scan;   // experiments (GRADSLAM_HIP_ICP_FORCE_SCAN): the first query of every tile always takes the scan pass
  ItSeq s[GS_MAX_BATCH];
#ifdef GS_ICP_TIMELINE
  unsigned long long* tl;   // debugging builds: 8 words per block [start, loads issued, prologue done, list pass done,
                            // scan pass done, leftovers done, end (100 MHz ticks), counts]
#endif
};
#ifdef GS_ICP_TIMELINE
#define IT_STAMP(k) do { if (hb.tl && threadIdx.x == 0) hb.tl[8 * (size_t)blockIdx.x + (k)] = wall_clock64(); } while (0)
#define IT_NOTE(k, v) do { if (hb.tl && threadIdx.x == 0) hb.tl[8 * (size_t)blockIdx.x + (k)] = (unsigned long long)(v); } while (0)
#else
#define IT_STAMP(k) do { } while (0)
#define IT_NOTE(k, v) do { } while (0)
#endif

struct ItBox {
  int x0, y0, z0, nx, ny, nz;
};

// What a scan leaves behind for the searches that follow (EMIT): this lane's share of the query's CANDIDATE LIST -- the
// slab slots of every target closer than R to the query position q0 of the scan -- and R itself (<= 0: no list).
// Every target that is not on the list is at least R away from q0 (inside the 2x2x2 block by its computed distance,
// outside by the block's face bound), so a later query position q with |q - q0| = delta has its exact nearest
// neighbour on the list whenever the best list entry is closer than R - delta (it_list_search).
struct ItList {
  uint32_t w[2];   // four 16-bit slots, appended from the low end; 0xffff = empty
  float R;
};
constexpr int IT_LIST_LANE = 4;
constexpr float IT_RADD = 0.6f;    // cube searches (far queries): R = nearest distance + IT_RADD cells, at most the cube's
                                   // bound; halved (up to 3 times) while the list does not fit

// grid_search_stage0 (gs_knn.h) on either the global grid (LOCAL = false; cells = cell_start, pts = sorted) or a
// tile's slab (LOCAL = true; cells = the slab's 16-bit cell table, pts = its points in LDS, box = the slab's box): same
// cells, same candidates, same bound.  *served = false (LOCAL only): the 2x2x2 block of this query is not inside the
// box, nothing was searched.  *win: slot of the best candidate in `pts` (the one lane of the group that holds it; -1
// in the others).  EMIT (LOCAL, G = 2, called with rball = inf so that all 8 cells are scanned): also builds the list.
template <int G, bool LOCAL, bool EMIT, typename CT>
GS_DEV unsigned long long it_stage0(const GsGrid& g, const ItBox& box, const CT* __restrict__ cells,
                                    const float4* __restrict__ pts, float qx, float qy, float qz, int lane,
                                    const float rball, bool* resolved, bool* served, int* win, ItList* lst = nullptr) {
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
  const float rc = rball * g.inv_c * 1.0001f + 0.002f;
  const bool prune = rc < amin - 0.001f;
  bool ulx = true, uhx = true, uly = true, uhy = true, ulz = true, uhz = true;
  if (prune) {
    const float tx = fx + (float)(cx - x0), ty = fy + (float)(cy - y0), tz = fz + (float)(cz - z0);
    ulx = tx - rc < 1.0f; uhx = tx + rc >= 1.0f;
    uly = ty - rc < 1.0f; uhy = ty + rc >= 1.0f;
    ulz = tz - rc < 1.0f; uhz = tz + rc >= 1.0f;
  }
  if (EMIT) { lst->w[0] = lst->w[1] = ~0u; lst->R = 0.0f; }
  int sb0 = 0, sb1 = 0, sb2 = 0, sb3 = 0, e1 = 0, e2 = 0, e3 = 0, total = 0;
  {
    const int xa = (x0 >= 0 && ulx) ? x0 : x0 + 1, xb = (x0 + 1 < g.nx && uhx) ? x0 + 1 : x0;
    const bool zl = z0 >= 0 && ulz, zh = z0 + 1 < g.nz && uhz, yl = y0 >= 0 && uly, yh = y0 + 1 < g.ny && uhy;
    int r0, sy, sz;
    if (LOCAL) {
      const int ya = yl ? y0 : y0 + 1, yb = yh ? y0 + 1 : y0, za = zl ? z0 : z0 + 1, zb = zh ? z0 + 1 : z0;
      const bool inside = xa >= box.x0 && xb < box.x0 + box.nx && ya >= box.y0 && yb < box.y0 + box.ny &&
                          za >= box.z0 && zb < box.z0 + box.nz;
      if (!inside) {
        *served = false;
        *resolved = false;
        *win = -1;
        return ~0ull;
      }
      sy = box.nx; sz = box.ny * box.nx;
      r0 = (z0 - box.z0) * sz + (y0 - box.y0) * sy - box.x0;
    } else {
      sy = g.nx; sz = g.ny * g.nx;
      r0 = (z0 * g.ny + y0) * g.nx;
    }
    *served = true;
    const int r1 = r0 + sy, r2 = r0 + sz, r3 = r2 + sy;
    // all eight bounds are loaded unconditionally (rows that are not used: the address of a row that is -- one of
    // the four always is, the query's own cell is never dropped -- and the result ignored): loads under branches are
    // waited for one branch at a time, four dependent round trips instead of one
    const bool u0 = zl && yl, u1 = zl && yh, u2 = zh && yl, u3 = zh && yh;
    const int rs = u0 ? r0 : (u1 ? r1 : (u2 ? r2 : r3));
    const int a0 = (u0 ? r0 : rs) + xa, a1 = (u1 ? r1 : rs) + xa, a2 = (u2 ? r2 : rs) + xa, a3 = (u3 ? r3 : rs) + xa;
    const int w = xb + 1 - xa;
    const int lb0 = (int)cells[a0], lb1 = (int)cells[a1], lb2 = (int)cells[a2], lb3 = (int)cells[a3];
    const int le0 = (int)cells[a0 + w], le1 = (int)cells[a1 + w], le2 = (int)cells[a2 + w], le3 = (int)cells[a3 + w];
    int se0 = 0, se1 = 0, se2 = 0, se3 = 0;
    if (u0) { sb0 = lb0; se0 = le0; }
    if (u1) { sb1 = lb1; se1 = le1; }
    if (u2) { sb2 = lb2; se2 = le2; }
    if (u3) { sb3 = lb3; se3 = le3; }
    e1 = se0 - sb0;
    e2 = e1 + (se1 - sb1);
    e3 = e2 + (se2 - sb2);
    total = e3 + (se3 - sb3);
  }
  // candidates in flight per lane: 4 gathers from global memory (latency), 2 from LDS (registers)
  constexpr int U = LOCAL ? (EMIT ? 1 : 2) : 4;
  // EMIT: this lane's four nearest candidates (distance ascending, slab slots) and the smallest distance it dropped
  float n0 = __builtin_inff(), n1 = n0, n2 = n0, n3 = n0, ndrop = n0;
  uint32_t s0 = 0xffffu, s1 = 0xffffu, s2 = 0xffffu, s3 = 0xffffu;
  for (int t0 = lane; t0 < total; t0 += U * G) {
    float4 p[U];
    bool in[U];
    int ixs[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = t0 + u * G;
      in[u] = t < total;
      const int tt = in[u] ? t : 0;
      const int ix = tt < e1 ? sb0 + tt : (tt < e2 ? sb1 + (tt - e1) : (tt < e3 ? sb2 + (tt - e2) : sb3 + (tt - e3)));
      ixs[u] = ix;
      p[u] = pts[ix];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned long long k2 = in[u] ? grid_key(qx, qy, qz, p[u]) : ~0ull;
      const bool better = k2 < key;
      key = better ? k2 : key;
      bt = better ? t0 + u * G : bt;
      if (EMIT) {
        // (a NaN distance -- key ~0 -- compares false everywhere: never listed, never the dropped minimum)
        const float dn = k2 != ~0ull ? __uint_as_float((uint32_t)(k2 >> 32)) : __builtin_nanf("");
        const uint32_t sn = (uint32_t)ixs[u];
        const bool l0 = dn < n0, l1 = dn < n1, l2 = dn < n2, l3 = dn < n3;
        ndrop = fminf(ndrop, l3 ? n3 : dn);
        n3 = l2 ? n2 : (l3 ? dn : n3); s3 = l2 ? s2 : (l3 ? sn : s3);
        n2 = l1 ? n1 : (l2 ? dn : n2); s2 = l1 ? s1 : (l2 ? sn : s2);
        n1 = l0 ? n0 : (l1 ? dn : n1); s1 = l0 ? s0 : (l1 ? sn : s1);
        n0 = l0 ? dn : n0;             s0 = l0 ? sn : s0;
      }
    }
  }
  {
    const unsigned long long kmin = grid_group_min<G>(key);
    const int bs = bt < e1 ? sb0 + bt : (bt < e2 ? sb1 + (bt - e1) : (bt < e3 ? sb2 + (bt - e2) : sb3 + (bt - e3)));
    *win = (key == kmin && bt >= 0) ? bs : -1;
    key = kmin;
  }
  const float rb = (amin < 1.0e30f) ? (amin - 0.001f) * g.c : BIG;
  const float bd = __uint_as_float((uint32_t)(key >> 32));
  *resolved = prune ? (bd == bd) : (rb > 0.0f && (rb >= 1.0e30f ? bd == bd : bd <= rb * rb));
  if (EMIT) {
    // The list = the four nearest candidates of either lane.  Every other target is either outside the 2x2x2 block
    // (at least rb away) or a candidate one of the lanes dropped (at least that lane's ndrop away): R = the minimum.
    if (*resolved) {   // (the same for all lanes of the group)
      const float od = __shfl_xor(ndrop, 1, G);
      const float R = fminf(sqrtf(fminf(ndrop, od)), rb);
      lst->w[0] = s0 | (s1 << 16);
      lst->w[1] = s2 | (s3 << 16);
      lst->R = R;
    }
  }
  return key;
}

// The nearest neighbour of q among this lane's list entries (+ the other lane's: group minimum), and whether it is
// PROVABLY the exact nearest neighbour: best distance + |q - q0| < R with a 1e-4 relative margin (float rounding of the
// distances involved is ~1e-7 relative).
GS_DEV unsigned long long it_list_search(const uint32_t* w, const float4 c0R, const float4* __restrict__ pts, float qx,
                                         float qy, float qz, bool* proven, int* win) {
  unsigned long long key = ~0ull;
  int bs = -1;
#pragma unroll
  for (int u = 0; u < IT_LIST_LANE; ++u) {
    const uint32_t sl = (w[u >> 1] >> (16 * (u & 1))) & 0xffffu;
    const bool in = sl != 0xffffu;
    const float4 c = pts[in ? sl : 0u];
    const unsigned long long k2 = in ? grid_key(qx, qy, qz, c) : ~0ull;
    const bool better = k2 < key;
    key = better ? k2 : key;
    bs = better ? (int)sl : bs;
  }
  const unsigned long long kmin = grid_group_min<2>(key);
  *win = (key == kmin && bs >= 0) ? bs : -1;
  const float bd = __uint_as_float((uint32_t)(kmin >> 32));   // NaN: empty list
  const float ex = qx - c0R.x, ey = qy - c0R.y, ez = qz - c0R.z;
  const float delta = sqrtf(ex * ex + ey * ey + ez * ez);
  *proven = sqrtf(bd) + delta < c0R.w * 0.9999f;   // false for NaN and for R <= 0
  return kmin;
}

// The scan of the first search of a solve (every query, all waves busy: throughput, not latency, is what counts):
// exact nearest neighbour + candidate list from a block of cells chosen PER AXIS -- the two cells nearest to the query
// where it sits within IT_SPAN_LO of a cell boundary, three cells (its own and both neighbours) where it sits near the
// middle of its cell.  Every face of that block with cells behind it is then at least 0.65 cells away (the plain 2x2x2
// block: 0.5), which is what leaves a list room to move: R - d1 >= ~0.4 cells for a typical neighbour distance, more
// than a solve moves a query.  Lane l of the pair walks the rows l, l + 2, ... of the block.
constexpr float IT_SPAN_LO = 0.25f;
GS_DEV unsigned long long it_scan_adaptive(const GsGrid& g, const ItBox& box, const uint16_t* __restrict__ cells,
                                           const float4* __restrict__ pts, float qx, float qy, float qz, int lane,
                                           bool* resolved, bool* served, int* win, ItList* lst) {
  const GsQueryCell qc = grid_query_cell(g, qx, qy, qz);
  const float fx = (qc.px - g.ox) * g.inv_c - (float)qc.cx, fy = (qc.py - g.oy) * g.inv_c - (float)qc.cy,
              fz = (qc.pz - g.oz) * g.inv_c - (float)qc.cz;
  const float BIG = 3.0e38f;
  // per axis: first cell, last cell (clipped to the grid) and the distance (cells) to the nearest face with cells behind
  int xa = fx > 1.0f - IT_SPAN_LO ? qc.cx : qc.cx - 1, xb = fx < IT_SPAN_LO ? qc.cx : qc.cx + 1;
  int ya = fy > 1.0f - IT_SPAN_LO ? qc.cy : qc.cy - 1, yb = fy < IT_SPAN_LO ? qc.cy : qc.cy + 1;
  int za = fz > 1.0f - IT_SPAN_LO ? qc.cz : qc.cz - 1, zb = fz < IT_SPAN_LO ? qc.cz : qc.cz + 1;
  const float ax = fminf(xa >= 1 ? fx + (float)(qc.cx - xa) : BIG, xb + 1 < g.nx ? (float)(xb + 1 - qc.cx) - fx : BIG);
  const float ay = fminf(ya >= 1 ? fy + (float)(qc.cy - ya) : BIG, yb + 1 < g.ny ? (float)(yb + 1 - qc.cy) - fy : BIG);
  const float az = fminf(za >= 1 ? fz + (float)(qc.cz - za) : BIG, zb + 1 < g.nz ? (float)(zb + 1 - qc.cz) - fz : BIG);
  const float amin = fminf(ax, fminf(ay, az));
  xa = xa < 0 ? 0 : xa; xb = xb >= g.nx ? g.nx - 1 : xb;
  ya = ya < 0 ? 0 : ya; yb = yb >= g.ny ? g.ny - 1 : yb;
  za = za < 0 ? 0 : za; zb = zb >= g.nz ? g.nz - 1 : zb;
  lst->w[0] = lst->w[1] = ~0u;
  lst->R = 0.0f;
  *win = -1;
  if (xa < box.x0 || xb >= box.x0 + box.nx || ya < box.y0 || yb >= box.y0 + box.ny || za < box.z0 || zb >= box.z0 + box.nz) {
    *served = false;
    *resolved = false;
    return ~0ull;
  }
  *served = true;
  const int ny = yb - ya + 1, nrow = ny * (zb - za + 1);
  unsigned long long key = ~0ull;
  int bs = -1;
  float n0 = __builtin_inff(), n1 = n0, n2 = n0, n3 = n0, ndrop = n0;
  uint32_t s0 = 0xffffu, s1 = 0xffffu, s2 = 0xffffu, s3 = 0xffffu;
  for (int r = lane; r < nrow; r += 2) {
    const int zz = za + r / ny, yy = ya + r % ny;
    const int row = ((zz - box.z0) * box.ny + (yy - box.y0)) * box.nx - box.x0;
    const int je = (int)cells[row + xb + 1];
    for (int j = (int)cells[row + xa]; j < je; ++j) {
      const unsigned long long k2 = grid_key(qx, qy, qz, pts[j]);
      if (k2 < key) { key = k2; bs = j; }
      // (a NaN distance -- key ~0 -- compares false everywhere: never listed, never the dropped minimum)
      const float dn = k2 != ~0ull ? __uint_as_float((uint32_t)(k2 >> 32)) : __builtin_nanf("");
      const uint32_t sn = (uint32_t)j;
      const bool l0 = dn < n0, l1 = dn < n1, l2 = dn < n2, l3 = dn < n3;
      ndrop = fminf(ndrop, l3 ? n3 : dn);
      n3 = l2 ? n2 : (l3 ? dn : n3); s3 = l2 ? s2 : (l3 ? sn : s3);
      n2 = l1 ? n1 : (l2 ? dn : n2); s2 = l1 ? s1 : (l2 ? sn : s2);
      n1 = l0 ? n0 : (l1 ? dn : n1); s1 = l0 ? s0 : (l1 ? sn : s1);
      n0 = l0 ? dn : n0;             s0 = l0 ? sn : s0;
    }
  }
  const unsigned long long kmin = grid_group_min<2>(key);
  *win = (key == kmin && bs >= 0) ? bs : -1;
  const float rb = (amin < 1.0e30f) ? (amin - 0.001f) * g.c : BIG;
  const float bd = __uint_as_float((uint32_t)(kmin >> 32));
  *resolved = rb > 0.0f && (rb >= 1.0e30f ? bd == bd : bd <= rb * rb);
  if (*resolved) {
    const float od = __shfl_xor(ndrop, 1, 2);
    lst->w[0] = s0 | (s1 << 16);
    lst->w[1] = s2 | (s3 << 16);
    lst->R = fminf(sqrtf(fminf(ndrop, od)), rb);
  }
  return kmin;
}

// Cubes of Chebyshev radius 1 .. kmax around the query's cell, on a tile's slab (grid_search_rings of gs_knn.h on the
// slab's cell table and LDS points): serves the queries whose neighbour is farther than the 2x2x2 stage can prove,
// as long as the cube stays inside the slab's box (*inbox = false otherwise: the caller goes to the global grid).
// *kdone = radius of the last cube scanned completely.
template <int G>
GS_DEV unsigned long long it_rings_local(const GsGrid& g, const ItBox& box, const uint16_t* __restrict__ cells,
                                         const float4* __restrict__ pts, float qx, float qy, float qz, int lane,
                                         unsigned long long key, const int kmax, bool* resolved, bool* inbox, int* win,
                                         int* kdone) {
  const GsQueryCell qc = grid_query_cell(g, qx, qy, qz);
  bool done = false;
  int bs = -1, k = 1;
  *inbox = true;
  *kdone = 0;
  for (; k <= kmax && !done; ++k) {
    const int xa = qc.cx - k < 0 ? 0 : qc.cx - k, xb = qc.cx + k >= g.nx ? g.nx - 1 : qc.cx + k;
    const int ya = qc.cy - k < 0 ? 0 : qc.cy - k, yb = qc.cy + k >= g.ny ? g.ny - 1 : qc.cy + k;
    const int za = qc.cz - k < 0 ? 0 : qc.cz - k, zb = qc.cz + k >= g.nz ? g.nz - 1 : qc.cz + k;
    if (xa < box.x0 || xb >= box.x0 + box.nx || ya < box.y0 || yb >= box.y0 + box.ny || za < box.z0 || zb >= box.z0 + box.nz) {
      *inbox = false;
      break;
    }
    const int side = 2 * k + 1, nrow = side * side;
    for (int r = lane; r < nrow; r += G) {
      const int zz = qc.cz + r / side - k, yy = qc.cy + r % side - k;
      if (zz < 0 || zz >= g.nz || yy < 0 || yy >= g.ny) continue;
      const int row = ((zz - box.z0) * box.ny + (yy - box.y0)) * box.nx - box.x0;
      const int je = (int)cells[row + xb + 1];
      for (int j = (int)cells[row + xa]; j < je; ++j) {
        const unsigned long long k2 = grid_key(qx, qy, qz, pts[j]);
        if (k2 < key) { key = k2; bs = j; }
      }
    }
    {
      const unsigned long long own = key;
      key = grid_group_min<G>(key);
      if (own != key) bs = -1;
    }
    const float rb = (float)k * g.c * 0.999f;
    const float bd = __uint_as_float((uint32_t)(key >> 32));
    done = bd <= rb * rb;
    *kdone = k;
  }
  *resolved = done;
  *win = bs;
  return key;
}

// Candidate list of a query served by it_rings_local: every slab target closer than R = min(d1 + radd, 0.999 kE cells)
// to the query, kE = the largest cube (<= kdone + 1) inside the box; up to 8 entries, collected by the G lanes into
// `stage` (8 x 16 bit + a counter, LDS).  Returns R, or 0 when no radius down to d1 + radd / 8 gives a list that fits.
template <int G>
GS_DEV float it_emit_cube(const GsGrid& g, const ItBox& box, const uint16_t* __restrict__ cells,
                          const float4* __restrict__ pts, float qx, float qy, float qz, int lane, const float d1,
                          const int kdone, uint16_t* stage, int* stage_n) {
  const GsQueryCell qc = grid_query_cell(g, qx, qy, qz);
  int kE = kdone + (kdone < 2 ? 1 : 0);   // (beyond 5x5x5 the next cube costs more than the room it adds)
  for (; kE > kdone; --kE) {   // kdone itself is inside the box (it was scanned)
    const int xa = qc.cx - kE < 0 ? 0 : qc.cx - kE, xb = qc.cx + kE >= g.nx ? g.nx - 1 : qc.cx + kE;
    const int ya = qc.cy - kE < 0 ? 0 : qc.cy - kE, yb = qc.cy + kE >= g.ny ? g.ny - 1 : qc.cy + kE;
    const int za = qc.cz - kE < 0 ? 0 : qc.cz - kE, zb = qc.cz + kE >= g.nz ? g.nz - 1 : qc.cz + kE;
    if (!(xa < box.x0 || xb >= box.x0 + box.nx || ya < box.y0 || yb >= box.y0 + box.ny || za < box.z0 || zb >= box.z0 + box.nz)) break;
  }
  const int xa = qc.cx - kE < 0 ? 0 : qc.cx - kE, xb = qc.cx + kE >= g.nx ? g.nx - 1 : qc.cx + kE;
  const int side = 2 * kE + 1, nrow = side * side;
  const float rcube = (float)kE * g.c * 0.999f;
  float radd = IT_RADD * g.c, R = 0.0f;
  for (int attempt = 0; attempt < 4; ++attempt, radd *= 0.5f) {
    float Rt = d1 + radd;
    Rt = Rt < rcube ? Rt : rcube;
    const float R2 = Rt * Rt;
    if (lane == 0) {
      *stage_n = 0;
#pragma unroll
      for (int u = 0; u < 8; ++u) stage[u] = 0xffffu;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int r = lane; r < nrow; r += G) {
      const int zz = qc.cz + r / side - kE, yy = qc.cy + r % side - kE;
      if (zz < 0 || zz >= g.nz || yy < 0 || yy >= g.ny) continue;
      const int row = ((zz - box.z0) * box.ny + (yy - box.y0)) * box.nx - box.x0;
      const int je = (int)cells[row + xb + 1];
      for (int j = (int)cells[row + xa]; j < je; ++j) {
        const float4 c = pts[j];
        const float dx = qx - c.x, dy = qy - c.y, dz = qz - c.z;
        float d = dx * dx;
        d = gs_fma(dy, dy, d);
        d = gs_fma(dz, dz, d);
        if (d < R2) {
          const int pos = atomicAdd(stage_n, 1);
          if (pos < 8) stage[pos] = (uint16_t)j;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int n = *stage_n;   // the same for all lanes of the group (same wave: LDS accesses are ordered)
    __builtin_amdgcn_wave_barrier();
    if (n <= 8) { R = Rt; break; }
  }
  return R;
}

// The same on the GLOBAL grid, for a query served by grid_search_rings (cube of radius kdone scanned): up to 4 global
// slots (32 bit) collected into `stage`; R = min(d1 + radd, 0.999 kdone cells).  Returns R or 0 (no list).
template <int G>
GS_DEV float it_emit_cube_global(const GsGrid& g, const int* __restrict__ cell_start, const float4* __restrict__ sorted,
                                 float qx, float qy, float qz, int lane, const float d1, const int kdone,
                                 uint32_t* stage, int* stage_n) {
  const GsQueryCell qc = grid_query_cell(g, qx, qy, qz);
  const int xa = qc.cx - kdone < 0 ? 0 : qc.cx - kdone, xb = qc.cx + kdone >= g.nx ? g.nx - 1 : qc.cx + kdone;
  const int side = 2 * kdone + 1, nrow = side * side;
  const float rcube = (float)kdone * g.c * 0.999f;
  float radd = IT_RADD * g.c, R = 0.0f;
  for (int attempt = 0; attempt < 4; ++attempt, radd *= 0.5f) {
    float Rt = d1 + radd;
    Rt = Rt < rcube ? Rt : rcube;
    const float R2 = Rt * Rt;
    if (lane == 0) {
      *stage_n = 0;
#pragma unroll
      for (int u = 0; u < 4; ++u) stage[u] = ~0u;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int r = lane; r < nrow; r += G) {
      const int zz = qc.cz + r / side - kdone, yy = qc.cy + r % side - kdone;
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
          if (pos < 4) stage[pos] = (uint32_t)j;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int n = *stage_n;
    __builtin_amdgcn_wave_barrier();
    if (n <= 4) { R = Rt; break; }
  }
  return R;
}

// it_list_search for a list of global slots (2 per lane, 0xffffffff = none): candidates gathered from `sorted`.
GS_DEV unsigned long long it_list_search_global(const uint32_t* w, const float4 c0R, const float4* __restrict__ sorted,
                                                float qx, float qy, float qz, bool* proven, int* win) {
  const bool in0 = w[0] != ~0u, in1 = w[1] != ~0u;
  const float4 a = sorted[in0 ? w[0] : 0u], b = sorted[in1 ? w[1] : 0u];
  const unsigned long long k0 = in0 ? grid_key(qx, qy, qz, a) : ~0ull, k1 = in1 ? grid_key(qx, qy, qz, b) : ~0ull;
  const unsigned long long key = k1 < k0 ? k1 : k0;
  const int bs = k1 < k0 ? (int)w[1] : (k0 != ~0ull ? (int)w[0] : -1);
  const unsigned long long kmin = grid_group_min<2>(key);
  *win = (key == kmin && bs >= 0) ? bs : -1;
  const float bd = __uint_as_float((uint32_t)(kmin >> 32));
  const float ex = qx - c0R.x, ey = qy - c0R.y, ez = qz - c0R.z;
  const float delta = sqrtf(ex * ex + ey * ey + ez * ez);
  *proven = sqrtf(bd) + delta < -c0R.w * 0.9999f;
  return kmin;
}

// icp_sum_rows<IT_BLOCK, IT_CH> / icp_sum_col27<IT_BLOCK> (gs_icp_loop.hip) in two steps, so that the loads of the first
// round are requested at the very start of the kernel, before anything that waits for the slab header: same values
// added in the same order.
constexpr int IT_CH = 4;
GS_DEV void it_rows_first(const double* __restrict__ partials, int nrows, double* a) {
  // branch-free (clamped addresses, every lane loads): a load under a branch makes the compiler wait for it at the
  // join, which would stall the requests that follow
  const int i = (threadIdx.x & 31) < LIN_NV ? (threadIdx.x & 31) : LIN_NV - 1, j = threadIdx.x >> 5;
#pragma unroll
  for (int u = 0; u < IT_CH; ++u) {
    const int b = j * IT_CH + u;
    a[u] = partials[(int64_t)(b < nrows ? b : nrows - 1) * LIN_NV + i];
  }
}
GS_DEV void it_rows_finish(const double* __restrict__ partials, int nrows, const double* a0, double* S, double (*sub)[32]) {
  constexpr int STEP = IT_BLOCK / 32;
  const int i = threadIdx.x & 31, j = threadIdx.x >> 5;
  double s = 0.0;
  if (i < LIN_NV) {
    if (j * IT_CH < nrows) {
#pragma unroll
      for (int u = 0; u < IT_CH; ++u) s += (j * IT_CH + u < nrows) ? a0[u] : 0.0;
    }
    for (int b = j * IT_CH + IT_CH * STEP; b < nrows; b += IT_CH * STEP) {
      const double* base = partials + (int64_t)b * LIN_NV + i;
      const int left = nrows - b;
      double a[IT_CH];
#pragma unroll
      for (int u = 0; u < IT_CH; ++u) a[u] = (u < left) ? base[u * LIN_NV] : 0.0;
#pragma unroll
      for (int u = 0; u < IT_CH; ++u) s += a[u];
    }
  }
  sub[j][i] = s;
  __syncthreads();
  if (threadIdx.x < LIN_NV) {
    double t = 0.0;
    for (int k = 0; k < STEP; ++k) t += sub[k][threadIdx.x];
    S[threadIdx.x] = t;
  }
  __syncthreads();
}
GS_DEV double it_col27_first(const double* __restrict__ partials, int nrows) {
  const int b = (int)threadIdx.x < nrows ? (int)threadIdx.x : nrows - 1;   // (branch-free, see it_rows_first)
  return partials[(int64_t)b * LIN_NV + 27];
}
GS_DEV double it_col27_finish(const double* __restrict__ partials, int nrows, const double first, double* red) {
  double s = 0.0;
  if ((int)threadIdx.x < nrows) s += first;
  for (int b = threadIdx.x + IT_BLOCK; b < nrows; b += IT_BLOCK) s += partials[(int64_t)b * LIN_NV + 27];
  s = gs_wave_sum_f64(s);
  if ((threadIdx.x & (GS_WAVE - 1)) == 0) red[threadIdx.x / GS_WAVE] = s;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < IT_BLOCK / GS_WAVE; ++w) t += red[w];
  __syncthreads();
  return t;
}

// slot codes in bslot_s: >= 0 slot in the LDS slab, <= -2 global slot -2 - code, -1 none
GS_DEV int it_global_code(int slot) { return -2 - slot; }

struct ItLds {
  float4 pts[IT_PTS_CAP];          // slab: binned target points of the tile's box (x, y, z, map row bits)
  float4 nspec[IT_NQ];             // normal of every query's previous match (x, y, z, slab slot bits)
  alignas(16) float qa[IT_NQ][8];  // a0..a5, residual of every query (zero when filtered out); before the rows are
                                   // built: the slab's cell table
  IcpSmall sm;                     // solver state
  double S[32];
  double sub[IT_BLOCK / 32][32];   // prologue: chunk sums of the partial rows; epilogue: the row groups' sub-sums
  unsigned long long keys[IT_NQ];  // best (distance bits, map row) of every query
  int bslot[IT_NQ];                // where its point / normal sit (slot codes above)
  float qs[IT_NQ][3];              // transformed queries
  int scan_q[IT_NQ], hard_q[IT_NQ];   // (the scan list's storage is reused for the queries left to brute force)
  int scan_n, hard_n, unres_n;
  unsigned long long red[IT_BLOCK / GS_WAVE];
};
static_assert(IT_RG * LIN_NV <= (IT_BLOCK / 32) * 32 && offsetof(ItLds, sm) % 16 == 0 && offsetof(ItLds, qa) % 16 == 0 &&
              offsetof(ItLds, sm) + sizeof(IcpSmall) <= 65536, "LDS copy targets");

template <bool FULL>
__global__ void __launch_bounds__(IT_BLOCK, 6) gs_icp_tile_half_kernel(const ItBatch hb, const float dist_thresh,
                                                                       const gs_icp_params prm, const int it,
                                                                       const int rows_in_reduced) {
  const unsigned B = (unsigned)hb.B, blk = blockIdx.x / B, nblk = gridDim.x / B;
  const unsigned X = (GS_XCDS % B == 0) ? GS_XCDS / B : 1u;
  const ItSeq& q = hb.s[blockIdx.x % B];
  const int tile = (int)gs_xcd_block(blk, nblk, X);

  // one LDS block with the targets of the global -> LDS copies first (their LDS address goes through M0: keep them in
  // the low 64 KB whatever the hardware reads of it)
  __shared__ ItLds L;
  IcpSmall& sm = L.sm;
  double* const S = L.S;
  double (*const sub)[32] = L.sub;
  float4* const pts_s = L.pts;
  float4* const nspec_s = L.nspec;
  unsigned long long* const keys_s = L.keys;
  int* const bslot_s = L.bslot;
  float (*const qs)[3] = L.qs;
  float (*const qa_s)[8] = L.qa;
  double (*const sub_s)[LIN_NV] = reinterpret_cast<double (*)[LIN_NV]>(L.sub);
  int* const scan_q = L.scan_q;
  int* const hard_q = L.hard_q;
  int* const unres_q = L.scan_q;
  unsigned long long* const red = L.red;

  const float* __restrict__ src_in = q.src_in;
  float* __restrict__ src_out = q.src_out;
  const int* __restrict__ cell_start = q.cell_start;
  const float4* __restrict__ sorted = q.sorted;
  const float4* __restrict__ sorted_n = q.sorted_n;
  float* __restrict__ d2prev = q.d2prev;
  const double* __restrict__ partials_in = q.partials_in;
  double* __restrict__ partials_out = q.partials_out;

  IT_STAMP(0);
  const int lane = threadIdx.x & (IT_G - 1), slot = threadIdx.x / IT_G;
  const int tx0 = (tile % hb.tiles_x) * IT_TW, ty0 = (tile / hb.tiles_x) * IT_TH;
  const int lx = tx0 + slot % IT_TW, ly = ty0 + slot / IT_TW;
  const bool live = lx < hb.Wl && ly < hb.Hl;
  const int s = ly * hb.Wl + lx;   // (32-bit on purpose: registers)
  const bool bounded = !(FULL && it == 0);  // the first search of a solve has no predecessor
  const int wv = threadIdx.x / GS_WAVE, ln = threadIdx.x & (GS_WAVE - 1);
  const char* slab = q.slabs + IT_SLAB_BYTES * (size_t)tile;
  const float4* gp4 = reinterpret_cast<const float4*>(slab + IT_OFF_PTS);
  const int nrows_in = rows_in_reduced ? 1 : hb.ntiles;
  const GsGrid g = *q.gp;
  const ItSlabHdr hdr = *reinterpret_cast<const ItSlabHdr*>(slab);
  const bool local = hdr.mode == 1;

  // ---- (1) everything that does not depend on the slab header is requested first.
  // Values that stay in registers across the prologue are kept few (the kernel sits at the 80-VGPR limit of two
  // workgroups per CU): the list centre / radius (x, y, z, R) is split over the query's two lanes (ca, cb = x, y in
  // lane 0 and z, R in lane 1, exchanged after the prologue).
  float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f, ca = __builtin_inff(), cb = 0.0f;
  uint32_t lw[2] = {~0u, ~0u};
  if (live) {
    p0 = src_in[3 * s];
    p1 = src_in[3 * s + 1];
    p2 = src_in[3 * s + 2];
    if (bounded) {
      const uint32_t* cw = q.cand + 4 * s + 2 * lane;
      lw[0] = cw[0]; lw[1] = cw[1];
      const float* cf = reinterpret_cast<const float*>(q.cq + s) + 2 * lane;
      ca = cf[0]; cb = cf[1];
    }
  }
  // partial rows of the previous half-iteration (first round of the sums below)
  double rows0[IT_CH], col0 = 0.0;
  if (FULL) {
    if (it > 0) col0 = it_col27_first(partials_in, nrows_in);
  } else {
    it_rows_first(partials_in, nrows_in, rows0);
  }
  // state of the previous half-iteration: 240 bytes, 16 per lane of the second wave, straight into LDS (a register
  // copy would make this wave wait for its loads before the slab and the partial rows are even requested)
  static_assert(sizeof(IcpSmall) % 16 == 0, "state copy");
  if (threadIdx.x >= GS_WAVE && threadIdx.x < GS_WAVE + (int)(sizeof(IcpSmall) / 16))
    it_load_lds16(reinterpret_cast<const float4*>(q.st_in) + (threadIdx.x - GS_WAVE), &sm);
  // the normal of every query's previous match (thread t fetches query t's): the match rarely changes between
  // searches, so the Gauss-Newton rows seldom have to wait for a gather from the slab's normals (which stay in global
  // memory)
  if (bounded && threadIdx.x < IT_NQ) {
    const int tlx = tx0 + (int)threadIdx.x % IT_TW, tly = ty0 + (int)threadIdx.x / IT_TW;
    if (tlx < hb.Wl && tly < hb.Hl) it_load_lds16(q.cn + (tly * hb.Wl + tlx), nspec_s + wv * GS_WAVE);
  }
  // the slab's points: global -> LDS directly (global_load_lds_dwordx4: wave-uniform LDS base + 16 B per lane, no
  // staging registers); the latency hides behind the prologue, the next barrier drains it.  The first IT_PTS_EARLY
  // slots are requested whatever the header says (the bytes exist; a slab is seldom smaller), the rest below.
#pragma unroll
  for (int c0 = 0; c0 < IT_PTS_EARLY / GS_WAVE; c0 += IT_BLOCK / GS_WAVE) {
    const int c = c0 + wv;   // chunk c = slots [64 c, 64 c + 64)
    if (c < IT_PTS_EARLY / GS_WAVE) it_load_lds16(gp4 + c * GS_WAVE + ln, pts_s + c * GS_WAVE);
  }

  // ---- (2) what depends on the header
  {
    const int npts = local ? hdr.npts : 0;
#pragma unroll
    for (int c0 = IT_PTS_EARLY / GS_WAVE; c0 < IT_PTS_CAP / GS_WAVE; c0 += IT_BLOCK / GS_WAVE) {
      const int c = c0 + wv, i = c * GS_WAVE + ln;
      if (c < IT_PTS_CAP / GS_WAVE && i < npts) it_load_lds16(gp4 + i, pts_s + c * GS_WAVE);
    }
  }
  // (the slab's cell table is fetched on demand, by the blocks that scan: see pass 2)
  float dprev = __builtin_inff();   // tiles without a slab: one more dependent load, the price of the exception
  if (!local && live && bounded) dprev = d2prev[s];

  IT_STAMP(1);
  // ---- prologue: finish the previous half-iteration (identical in every block)
  if (FULL) {
    double e1 = 0.0;
    if (it > 0) e1 = it_col27_finish(partials_in, nrows_in, col0, reinterpret_cast<double*>(red));
    else __syncthreads();
    if (threadIdx.x == 0) {
      if (it > 0)
        icp_update_math((float)e1, sm, prm, (tile == 0 && it - 1 < GS_ICP_MAX_ITERS) ? q.trace + 12 * (it - 1) : nullptr);
      L.scan_n = 0; L.unres_n = 0; L.hard_n = 0;
    }
  } else {
    it_rows_finish(partials_in, nrows_in, rows0, S, sub);
    if (threadIdx.x < GS_WAVE) {
      gs_solve_spd6_wave(S, sm.damp, sm.xi);   // 6x6 solve across the lanes of wave 0 (LDS accesses of a wave are ordered)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (threadIdx.x == 0) {
        icp_solve_finish(S, sm);
        L.scan_n = 0; L.unres_n = 0; L.hard_n = 0;
      }
    }
  }
  __syncthreads();
  if (tile == 0 && threadIdx.x < (int)(sizeof(IcpSmall) / 4))
    reinterpret_cast<float*>(q.st_out)[threadIdx.x] = reinterpret_cast<const float*>(&sm)[threadIdx.x];

  IT_STAMP(2);
  // ---- search, pass 1: one source point per IT_G-lane group, pending transform applied to the loaded point.
  // Tiles with a slab try the candidate list first; what it cannot prove goes to the scan pass.
  const ItBox box = {hdr.bx0, hdr.by0, hdr.bz0, hdr.nbx, hdr.nby, hdr.nbz};
  float qx = p0, qy = p0, qz = p0;
  const bool skip = !live || p0 != p0;   // beyond the lattice, or an empty slot (NaN stays NaN, contributes no row)
  if (!skip) {
    const float* T = FULL ? sm.T_step : sm.Tr;
    gs_rigid_fma(T, p0, p1, p2, qx, qy, qz);
    if (local) {
      bool proven = false;
      int win = -1;
      unsigned long long key = ~0ull;
      const float oa = __shfl_xor(ca, 1, IT_G), ob = __shfl_xor(cb, 1, IT_G);   // (never loaded for tiles without a slab)
      const float4 c0R = lane == 0 ? make_float4(ca, cb, oa, ob) : make_float4(oa, ob, ca, cb);
      // R > 0: list of slab slots; R < 0: list of global slots (a neighbour found by the cubes on the global grid)
      if (c0R.w > 0.0f) key = it_list_search(lw, c0R, pts_s, qx, qy, qz, &proven, &win);
      else if (c0R.w < 0.0f) key = it_list_search_global(lw, c0R, sorted, qx, qy, qz, &proven, &win);
      if (hb.force_scan && slot == 0) proven = false;
      if (proven) {
        if (win >= 0) bslot_s[slot] = c0R.w > 0.0f ? win : it_global_code(win);
        if (lane == 0) keys_s[slot] = key;
      } else if (lane == 0) {
        scan_q[atomicAdd(&L.scan_n, 1)] = slot;   // pass 2: the 2x2x2 scan, and a new list
      }
    } else {
      // no slab: the global grid, scan bounded by what the previous search found (the previous neighbour is still a
      // target; the previous query was Tr * p in the look-ahead half resp. p itself in the first half)
      float rball;
      {
        float ox = p0, oy = p1, oz = p2;
        if (FULL) gs_rigid_fma(sm.Tr, p0, p1, p2, ox, oy, oz);
        const float ex = qx - ox, ey = qy - oy, ez = qz - oz;
        rball = sqrtf(dprev) + sqrtf(ex * ex + ey * ey + ez * ez);
      }
      bool done, served;
      int win;
      const unsigned long long key = it_stage0<IT_G, false, false>(g, box, cell_start, sorted, qx, qy, qz, lane, rball,
                                                                   &done, &served, &win);
      if (win >= 0) bslot_s[slot] = it_global_code(win);
      if (lane == 0) {
        if (key == ~0ull) bslot_s[slot] = -1;
        keys_s[slot] = key;
        if (!done) hard_q[atomicAdd(&L.hard_n, 1)] = slot;
      }
    }
  }
  if (lane == 0 && live) {
    qs[slot][0] = qx; qs[slot][1] = qy; qs[slot][2] = qz;
    if (skip) keys_s[slot] = ~0ull;
  }
  __syncthreads();
  IT_STAMP(3);
  // ---- pass 2 (tiles with a slab): the 2x2x2 scan for the queries without a proof, which also writes their lists.
  // The slab's cell table is read from its LDS copy when there is one (tab_lds), else from global memory; the two
  // call sites keep the address space of the pointer known to the compiler.
  // (look-ahead launches only: the first halves scan every query once per solve, with all waves busy, and hardly ever
  // afterwards -- and their kernel has no scalar registers left for the copy)
  const bool tab_lds = !FULL && local && hdr.ncell + 1 <= IT_LDS_CELLS;
  const int ns = L.scan_n;   // block-uniform
  if (ns && tab_lds) {
    // the cell table is needed from here on, by the few launches that scan at all: copied into the LDS space of the
    // Gauss-Newton rows (written later); the scans and cube searches walk it row by row, from global memory every
    // step would be a round trip to cold lines and pages (nothing else touches this table)
    if ((int)threadIdx.x < (hdr.ncell + 1 + 7) / 8)   // (at most IT_LDS_CELLS / 8 = IT_BLOCK 16-byte words)
      it_load_lds16(reinterpret_cast<const uint4*>(slab + IT_OFF_CELLS) + threadIdx.x,
                    reinterpret_cast<uint4*>(&qa_s[0][0]) + wv * GS_WAVE);
    __syncthreads();
  }
  auto scan_pass = [&](const uint16_t* tab) __attribute__((always_inline)) {
    for (int i = threadIdx.x / IT_G; i < ns; i += IT_NQ) {
      const int hs = scan_q[i];
      const float hx = qs[hs][0], hy = qs[hs][1], hz = qs[hs][2];
      bool done, served;
      int win;
      ItList lst;
      // first search of a solve: the roomier per-axis block (every wave is busy, the extra cells cost throughput only);
      // later (a list lost its proof, few queries per tile, latency counts): the plain 2x2x2 block
      const unsigned long long key = !bounded
          ? it_scan_adaptive(g, box, tab, pts_s, hx, hy, hz, lane, &done, &served, &win, &lst)
          : it_stage0<IT_G, true, true>(g, box, tab, pts_s, hx, hy, hz, lane, __builtin_inff(), &done, &served, &win, &lst);
      if (win >= 0) bslot_s[hs] = win;
      const int sh = (ty0 + hs / IT_TW) * hb.Wl + (tx0 + hs % IT_TW);
      uint32_t* cw = q.cand + 4 * sh + 2 * lane;
      cw[0] = lst.w[0]; cw[1] = lst.w[1];
      if (lane == 0) {
        q.cq[sh] = make_float4(hx, hy, hz, lst.R);
        if (key == ~0ull) bslot_s[hs] = -1;
        keys_s[hs] = key;
        // bit 31: not served by the slab (the leftover pass starts with the global 2x2x2 stage)
        if (!done) hard_q[atomicAdd(&L.hard_n, 1)] = hs | (served ? 0 : (int)0x80000000);
      }
    }
  };
  if (ns) {
    if (tab_lds) scan_pass(reinterpret_cast<const uint16_t*>(&qa_s[0][0]));
    else scan_pass(reinterpret_cast<const uint16_t*>(slab + IT_OFF_CELLS));
    __syncthreads();
  }
  IT_STAMP(4);
  // ---- leftovers (no proof from the list, neighbour farther than the 2x2x2 stage can prove, or outside the slab):
  // groups of IT_HG lanes, cubes on the slab first, then on the global grid
  const int nh = L.hard_n;  // block-uniform
  auto hard_pass = [&](const uint16_t* tab) __attribute__((always_inline)) {
    for (int i = threadIdx.x / IT_HG; i < nh; i += IT_BLOCK / IT_HG) {
      const int e = hard_q[i], hs = e & 0x7fffffff, l16 = threadIdx.x & (IT_HG - 1);
      const float hx = qs[hs][0], hy = qs[hs][1], hz = qs[hs][2];
      unsigned long long key = keys_s[hs];
      bool done = false, served;
      int win;
      if (e < 0) {
        key = it_stage0<IT_HG, false, false>(g, box, cell_start, sorted, hx, hy, hz, l16, __builtin_inff(), &done,
                                             &served, &win);
        if (win >= 0) bslot_s[hs] = it_global_code(win);
      } else if (local) {
        // cubes on the slab; a query they serve gets a candidate list, so that it costs a search only once
        bool inbox;
        int kdone;
        key = it_rings_local<IT_HG>(g, box, tab, pts_s, hx, hy, hz, l16, key, IT_LOCAL_RINGS, &done, &inbox, &win, &kdone);
        if (win >= 0) bslot_s[hs] = win;
        if (done) {
          // staging: 32 bytes per group in the row-sum scratch (free between the prologue and the epilogue)
          char* stg = reinterpret_cast<char*>(&sub[0][0]) + 32 * (threadIdx.x / IT_HG);
          uint16_t* stage = reinterpret_cast<uint16_t*>(stg);
          int* stage_n = reinterpret_cast<int*>(stg + 16);
          const float R = it_emit_cube<IT_HG>(g, box, tab, pts_s, hx, hy, hz, l16,
                                              sqrtf(__uint_as_float((uint32_t)(key >> 32))), kdone, stage, stage_n);
          if (l16 == 0) {
            const int sh = (ty0 + hs / IT_TW) * hb.Wl + (tx0 + hs % IT_TW);
            *reinterpret_cast<uint4*>(q.cand + 4 * sh) = *reinterpret_cast<const uint4*>(stage);
            q.cq[sh] = make_float4(hx, hy, hz, R);
          }
        }
      }
      if (!done) {
        int kdone;
        key = grid_search_rings<IT_HG>(g, cell_start, sorted, hx, hy, hz, l16, key, &done, &win, FS_HARD_RINGS, &kdone);
        if (win >= 0) bslot_s[hs] = it_global_code(win);
        if (done && local) {   // a list of global slots: the next searches of this query cost two gathers
          char* stg = reinterpret_cast<char*>(&sub[0][0]) + 32 * (threadIdx.x / IT_HG);
          uint32_t* stage = reinterpret_cast<uint32_t*>(stg);
          int* stage_n = reinterpret_cast<int*>(stg + 16);
          const float R = it_emit_cube_global<IT_HG>(g, cell_start, sorted, hx, hy, hz, l16,
                                                     sqrtf(__uint_as_float((uint32_t)(key >> 32))), kdone, stage, stage_n);
          if (l16 == 0) {
            const int sh = (ty0 + hs / IT_TW) * hb.Wl + (tx0 + hs % IT_TW);
            *reinterpret_cast<uint4*>(q.cand + 4 * sh) = *reinterpret_cast<const uint4*>(stage);
            q.cq[sh] = make_float4(hx, hy, hz, -R);
          }
        }
      }
      if (l16 == 0) {
        keys_s[hs] = key;
        if (!done) unres_q[atomicAdd(&L.unres_n, 1)] = hs;
      }
    }
  };
  if (nh) {
    if (tab_lds) hard_pass(reinterpret_cast<const uint16_t*>(&qa_s[0][0]));
    else hard_pass(reinterpret_cast<const uint16_t*>(slab + IT_OFF_CELLS));
    __syncthreads();
  }
  const int nun = L.unres_n;  // block-uniform
  IT_NOTE(7, (unsigned long long)nun | ((unsigned long long)hdr.mode << 10) | ((unsigned long long)ns << 12) |
                 ((unsigned long long)nh << 22) | ((unsigned long long)hdr.npts << 32) | ((unsigned long long)hdr.ncell << 44));
  for (int u = 0; u < nun; u += FS_BQ)
    block_brute_min_sorted_multi<IT_BLOCK, FS_BQ>(qs, unres_q + u, nun - u < FS_BQ ? nun - u : FS_BQ, sorted,
                                                  cell_start[g.ncell], keys_s, bslot_s, true);

  IT_STAMP(5);
  // ---- Gauss-Newton row of every query: its group's first lane reads the match and leaves [a, res] in LDS
  if (lane == 0) {
    float a[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}, res = 0.0f;
    if (!skip) {
      const unsigned long long bb = keys_s[slot];
      if (!local) d2prev[s] = __uint_as_float((uint32_t)(bb >> 32));  // NaN bits when nothing was found
      const float d2 = __uint_as_float((uint32_t)(bb >> 32));
      const bool keep = (dist_thresh < 0.0f) || (d2 < dist_thresh);
      const int bsl = bslot_s[slot];
      if (bb != ~0ull && bsl >= 0) {
        float4 nn = nspec_s[slot];
        if (!bounded || __float_as_int(nn.w) != bsl) {   // the match changed (or the first search of the solve)
          nn = reinterpret_cast<const float4*>(slab + IT_OFF_NRM)[bsl];
          nn.w = __int_as_float(bsl);
          q.cn[s] = nn;
        }
        gn_row_pn(qs[slot][0], qs[slot][1], qs[slot][2], pts_s[bsl], nn, a, res);
      } else if (bb != ~0ull && bsl <= -2) {
        gn_row_pn(qs[slot][0], qs[slot][1], qs[slot][2], sorted[-2 - bsl], sorted_n[-2 - bsl], a, res);
        if (!bounded && local) q.cn[s] = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(-1));  // (nothing of an earlier frame)
      } else {  // every distance was NaN / no target at all: row 0 of the target array, as the brute-force engine
        gn_row(qs[slot][0], qs[slot][1], qs[slot][2], q.tgt, q.tn, 0, a, res);
        if (!bounded && local) q.cn[s] = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(-1));
      }
      if (!keep) {
#pragma unroll
        for (int i = 0; i < 6; ++i) a[i] = 0.0f;
        res = 0.0f;
      }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) qa_s[slot][i] = a[i];
    qa_s[slot][6] = res;
    // the transformed cloud of this iteration (stored here, behind the last wait for global memory: a store in
    // flight at a barrier is waited for)
    if (FULL && live) { src_out[3 * s] = qs[slot][0]; src_out[3 * s + 1] = qs[slot][1]; src_out[3 * s + 2] = qs[slot][2]; }
  }
  __syncthreads();
  double* prow = partials_out + (int64_t)tile * LIN_NV;
  if (!FULL) {  // residual only: wave sums of the IT_NQ squares, added in wave order
    double* red2 = reinterpret_cast<double*>(red);
    const int wave = threadIdx.x / GS_WAVE;
    if (threadIdx.x < IT_NQ) {
      const float res = qa_s[threadIdx.x][6];
      const double sum = gs_wave_sum_f64((double)res * (double)res);
      if ((threadIdx.x & (GS_WAVE - 1)) == 0) red2[wave] = sum;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = red2[0];
#pragma unroll
      for (int w = 1; w < IT_NQ / GS_WAVE; ++w) t += red2[w];
      prow[27] = t;
    }
    IT_STAMP(6);
    return;
  }
  // IT_RG groups of 28 threads add the products of IT_RPG queries each, then 28 threads add the sub-sums in order
  if (threadIdx.x < IT_RG * LIN_NV) {
    const int i = threadIdx.x % LIN_NV, part = threadIdx.x / LIN_NV;
    const int ia = fs_pa(i), ib = fs_pb(i);
    const float* r0 = qa_s[IT_RPG * part];
    double t = (double)r0[ia] * (double)r0[ib];
#pragma unroll
    for (int u = 1; u < IT_RPG; ++u) t += (double)r0[8 * u + ia] * (double)r0[8 * u + ib];
    sub_s[part][i] = t;
  }
  __syncthreads();
  if (threadIdx.x < LIN_NV) {
    double t = sub_s[0][threadIdx.x];
#pragma unroll
    for (int k = 1; k < IT_RG; ++k) t += sub_s[k][threadIdx.x];
    prow[threadIdx.x] = t;
  }
  IT_STAMP(6);
}

// rows of a large solve added up once per half-iteration (more than FS_REDUCE_ROWS tiles)
struct ItRowsBatch {
  int B, nrows;
  const double* in[GS_MAX_BATCH];
  double* out[GS_MAX_BATCH];
};
__global__ void __launch_bounds__(IT_BLOCK) gs_icp_tile_reduce_rows_kernel(const ItRowsBatch rb) {
  __shared__ double S[32];
  __shared__ double sub[IT_BLOCK / 32][32];
  icp_sum_rows<IT_BLOCK, 4>(rb.in[blockIdx.x], rb.nrows, S, sub);
  if (threadIdx.x < LIN_NV) rb.out[blockIdx.x][threadIdx.x] = S[threadIdx.x];
}
