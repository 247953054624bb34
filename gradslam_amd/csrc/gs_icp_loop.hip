// gs_icp_loop.hip — the device-resident LM loop of point_to_plane_ICP / point_to_plane_gradICP
// (odometry/icputils.py:235-367, :370-545): 2 exact 1-NN searches, one Gauss-Newton linearisation,
// one 6x6 solve, two SE(3) exponentials per iteration, 20 iterations, no host sync.
//
// Normal equations are accumulated in float64 from float32 products (order-independent to ~1e-16,
// so HIP and oracle agree after the single rounding to float32) with fixed-order reductions.
//
// Grid path (large target sets; the SLAM loop): ONE kernel per half-iteration.  Its prologue lets
// every block redundantly finish the PREVIOUS half-iteration (add up the partial rows the previous
// kernel left, then the scalar stage: solve + se3_exp, or LM / gradLM update) so that no separate
// single-block kernels sit on the critical path; block 0 records the state.  Then GQ_G (8) lanes per
// source point run the grid search (gs_knn.h), the block finishes unresolved queries by brute
// force, builds the Gauss-Newton rows and emits one partial row.  State and partial rows are
// double-buffered between consecutive kernels.  2 x numiters + 1 launches per solve.
//
// Brute-force path (small problems, GRADSLAM_HIP_KNN=brute): search / linearise / solve / update
// as separate kernels.
#include <stdlib.h>
#include <string.h>

#include <memory>
#include <type_traits>

#include "gs_icp_math.h"
#include "gs_knn.h"
#include "gs_knn_bbox.h"

constexpr int GS_ICP_MAX_ITERS = 1024;  // rows of the per-iteration trace kept in the scratch
struct GsIcpState {
  IcpSmall s[2];                        // double-buffered by the grid path; the brute-force path uses s[0]
  float trace[GS_ICP_MAX_ITERS * 12];   // [err, new_err, damp, sigmoid, xi(6), 0, 0] per iteration
};

// Optional forward tape for gs_icp_backward_f32 (layout: gs_icp_math.h:GsIcpTape).
struct TapePtrs {
  float* src;    // [K][n_src][3]
  int32_t* idx;  // [K][2][n_src]
  float* sys;    // [K][28]
};
GS_DEV void tape_write_sys(float* __restrict__ sys, int it, const double* S, float damp) {
  float* t = sys + 28 * it;
  for (int i = 0; i < 27; ++i) t[i] = (float)S[i];
  t[27] = damp;
}

// ---------------------------------------------------------------- fixed-order sums ------
// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a release fence that also drains the wave's
// outstanding GLOBAL loads (s_waitcnt vmcnt(0)); the list variants of the half-iteration kernel keep the gathers of
// their candidate lists in flight across the whole prologue, which is the point of issuing them there.
GS_DEV void gs_barrier_lds() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
template <bool LDS_ONLY>
GS_DEV void gs_bar() {
  if (LDS_ONLY) gs_barrier_lds();
  else __syncthreads();
}

// Adds up partial rows (nrows x LIN_NV doubles).  Thread t works on value t%32 and the contiguous
// chunk of CH rows number t/32 (then every STEP*CH rows further): its CH loads are 224 B apart, i.e.
// one base address with immediate offsets, all in flight together; the chunk sums are then added in
// chunk order.  Every block that runs this on the same rows gets bit-identical sums.
template <int BLOCK, int CH = 18>
GS_DEV void icp_sum_rows(const double* __restrict__ partials, int nrows, double* S, double (*sub)[32]) {
  constexpr int STEP = BLOCK / 32;
  // CH = rows per thread per round of independent loads: a 640x480 solve of the row-unit engine (200 rows of 96
  // queries, 9 per thread at 768 threads) is ONE round of memory latency with 18; more rows in flight would push
  // the kernel past 80 VGPRs (2 resident blocks per CU).
  const int i = threadIdx.x & 31, j = threadIdx.x >> 5;
  double s = 0.0;
  if (i < LIN_NV) {
    for (int b = j * CH; b < nrows; b += CH * STEP) {
      const double* base = partials + (int64_t)b * LIN_NV + i;
      const int left = nrows - b;
      double a[CH];
#pragma unroll
      for (int u = 0; u < CH; ++u) a[u] = (u < left) ? base[u * LIN_NV] : 0.0;
#pragma unroll
      for (int u = 0; u < CH; ++u) s += a[u];
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

// hook(): called by EVERY thread once the first round of loads has been issued and before it is used (the list variants
// of the half-iteration kernel issue their dependent gathers there: behind the row loads in the queue, in front of the
// wait for them).
// The same sums with HALF the loads in flight per thread (9 instead of 18: 18 VGPRs less at the point where the list
// variants of the half-iteration kernel also hold their gathered candidates), bit for bit: a wave owns chunk j = its
// index (and chunk j + BLOCK / 64 in a second pass, which only solves of more than 9 * BLOCK / 32 rows need); lanes
// 0-31 add the first nine rows of the chunk in order, hand the running sum to lanes 32-63, which add the other nine --
// the order of icp_sum_rows<BLOCK, 18>, whose thread j adds its 18 rows one after the other.
template <int BLOCK, class Hook>
GS_DEV void icp_sum_rows_split(const double* __restrict__ partials, int nrows, double* S, double (*sub)[32], Hook hook) {
  constexpr int CH = 18, HC = CH / 2, STEP = BLOCK / 32, NW = BLOCK / GS_WAVE;
  static_assert(2 * NW == STEP, "two chunks per wave");
  // (an opaque copy of the thread index: merged with the uses at the far end of the kernel, the wave index computed
  // here would be the one value the register allocator spills -- and a spill is a store in the memory queue of the
  // prologue, behind which every wait for a load drains the queue)
  unsigned tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int ln = tid & (GS_WAVE - 1), i = ln & 31, half = ln >> 5, w = tid / GS_WAVE;
  const bool act = i < LIN_NV;
  double tot[2];
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int j = w + pass * NW;   // wave-uniform
    double t = 0.0;
    if (pass == 0 || j * CH < nrows) {
      for (int b = j * CH, r = 0; r == 0 || b < nrows; b += CH * STEP, ++r) {   // (r > 0: solves of more than 432 rows)
        // UNCONDITIONAL loads, masked afterwards: a load under a branch may not have been issued, and the compiler then
        // has to wait for ALL outstanding loads wherever it needs an older one -- here that would drain the gathers the
        // hook puts in flight.  Rows beyond nrows (at most 24 x 18 of them from the start of the buffer, 97 KB; the
        // single-column variant below: FS_BLOCK rows, 172 KB) are read from what follows the rows in the caller's
        // scratch (the second row buffer, the grid, the lists -- localize_chunk checks that they cover it before it
        // plans a solve with lists) and never looked at; one base address + immediate offsets, as in icp_sum_rows.
        const int b0 = b + half * HC;
        const int left = act ? nrows - b0 : 0;
        const double* base = partials + (int64_t)b0 * LIN_NV + i;
        double a[HC];
#pragma unroll
        for (int u = 0; u < HC; ++u) a[u] = base[u * LIN_NV];
        if (pass == 0 && r == 0) hook();
#pragma unroll
        for (int u = 0; u < HC; ++u) a[u] = (u < left) ? a[u] : 0.0;
        const double carry = __shfl(t, i + 32, GS_WAVE);   // the running sum of the chunk lives in the upper half-wave
        if (half == 0) {
          t = carry;
#pragma unroll
          for (int u = 0; u < HC; ++u) t += a[u];
        }
        const double lo = __shfl(t, i, GS_WAVE);
        if (half == 1) {
          t = lo;
#pragma unroll
          for (int u = 0; u < HC; ++u) t += a[u];
        }
      }
    }
    tot[pass] = t;
  }
  if (half == 1) {
    sub[w][i] = tot[0];
    sub[w + NW][i] = tot[1];
  }
  gs_barrier_lds();
  if (threadIdx.x < LIN_NV) {
    double t = 0.0;
    for (int k = 0; k < STEP; ++k) t += sub[k][threadIdx.x];
    S[threadIdx.x] = t;
  }
  gs_barrier_lds();
}

// Same for the single residual column (value 27): all threads share the rows.
template <int BLOCK>
GS_DEV double icp_sum_col27(const double* __restrict__ partials, int nrows, double* red) {
  double s = 0.0;
  for (int b = threadIdx.x; b < nrows; b += BLOCK) s += partials[(int64_t)b * LIN_NV + 27];
  s = gs_wave_sum_f64(s);
  if ((threadIdx.x & (GS_WAVE - 1)) == 0) red[threadIdx.x / GS_WAVE] = s;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < BLOCK / GS_WAVE; ++w) t += red[w];
  __syncthreads();
  return t;
}
// ... with the hook of icp_sum_rows_split and barriers that order LDS only (the same additions in the same order)
template <int BLOCK, class Hook>
GS_DEV double icp_sum_col27_hook(const double* __restrict__ partials, int nrows, double* red, Hook hook) {
  double s = 0.0;
  {
    const bool has = (int)threadIdx.x < nrows;   // (unconditional load, masked afterwards: see icp_sum_rows_split)
    double a0 = partials[(int64_t)threadIdx.x * LIN_NV + 27];
    hook();
    a0 = has ? a0 : 0.0;
    s += a0;
  }
  for (int b = threadIdx.x + BLOCK; b < nrows; b += BLOCK) s += partials[(int64_t)b * LIN_NV + 27];
  s = gs_wave_sum_f64(s);
  if ((threadIdx.x & (GS_WAVE - 1)) == 0) red[threadIdx.x / GS_WAVE] = s;
  gs_barrier_lds();
  double t = 0.0;
  for (int w = 0; w < BLOCK / GS_WAVE; ++w) t += red[w];
  gs_barrier_lds();
  return t;
}

// ---------------------------------------------------------------- grid path -------------
constexpr int FS_BLOCK = 768;            // 12 waves; 2 blocks per CU keep all 200 blocks of a 640x480 solve resident
                                         // (measured alternatives at 8 lanes per query: 512 threads 11.8 us,
                                         // 1024 threads 11.2 us, 768 threads 10.7 us per kernel)
constexpr int FS_HARD_RINGS = 5;         // cube radius (cells) the 16-lane groups go to before the brute-force fallback
constexpr int FS_BQ = 4;                 // queries per pass of the block-wide brute-force fallback
constexpr int FS_QPB = FS_BLOCK / GQ_G;  // 96 queries per block, their rows are built by the first two waves
constexpr int FS_RPG = (FS_QPB / 4) * LIN_NV <= FS_BLOCK ? 4 : 8;  // rows per group in the block reduction
constexpr int FS_RG = FS_QPB / FS_RPG;                            // row groups
constexpr int FS_FAR_PASSES = 8;     // list-building launches per solve at most: behind the first halves of iterations
                                     // 0, 1 (before / after the large first step of a solve) and of every 4th after that
// pass index of iteration `it`, or -1 (host and device)
__host__ __device__ inline int fs_far_pass(int it) {
  const int p = it < 2 ? it : ((it & 3) == 0 ? 1 + it / 4 : -1);
  return p < FS_FAR_PASSES ? p : -1;
}
constexpr int FS_FAR_MIN_RING = 2;   // queries served by a cube of at least this radius get a candidate list (gs_knn.h)
constexpr int FS_HG = 16;  // lanes per query of the shell search for queries the 2x2x2 stage leaves open
constexpr int FS_LISTS_FROM = 1;   // iteration behind whose look-ahead search the candidate lists of ordinary queries are built
static_assert(FS_QPB <= 2 * GS_WAVE && FS_QPB % FS_RPG == 0 && FS_RG * LIN_NV <= FS_BLOCK, "block shape");

// FULL = true : first half of iteration `it`  (prologue: LM update of iteration it-1, then search
//               with T_step applied, full normal equations)
// FULL = false: look-ahead half              (prologue: solve + se3_exp, then search with Tr
//               applied, residual only)
// Workgroups are dealt round-robin to the 8 XCDs (block i runs on XCD i % 8), each with its own L2.
// Logical block ids are therefore assigned so that every XCD works on ONE contiguous eighth of the
// query rows: consecutive queries are neighbouring pixels of the frame, so the slice of the binned
// target set an XCD touches is an eighth of the whole (+ halo) and stays resident in its L2 across
// the 2 x numiters kernels of a solve (the full set of a 1296x968 frame is larger than one L2).
constexpr unsigned GS_XCDS = 8;
// b-th of nb blocks that are dealt round-robin to X XCDs -> logical block such that every XCD owns one contiguous
// range of logical blocks
GS_DEV unsigned gs_xcd_block(unsigned b, unsigned nb, unsigned X = GS_XCDS) {
  const unsigned q = nb / X, r = nb % X, x = b % X;
  return x * q + (x < r ? x : r) + b / X;
}

// What one sequence contributes to a half-iteration launch (the batched launch carries up to GS_MAX_BATCH of them).
struct IcpHalfSeq {
  const float* src_in;
  float* src_out;
  const float* tgt;
  const float* tn;
  GsCount n_tgt;
  const GsGrid* gp;
  const int* cell_start;
  const float4* sorted;
  const float4* sorted_n;   // normals binned with the targets (NULL: gather from tn)
  float* d2prev;            // [n_src] squared distance of every source point's neighbour in the previous half-iteration
                            // (written by every launch, read by the next; NULL: no search bound)
  const double* partials_in;
  double* partials_out;
  const IcpSmall* st_in;
  IcpSmall* st_out;
  float* trace;
  int64_t* out_idx;
  int32_t* tape_idx;
  float* tape_sys;
  // candidate lists of far queries (gs_knn.h; NULL: none kept): far_cq[s] = (position the list of source point s was
  // built at, exactness radius), far_c[GS_FAR_SLOTS * s ..] = its slots of `sorted`, written by gs_icp_far_build_kernel
  // after the first search of a solve.  "s has a list" travels in the sign bit of d2prev[s], which every search loads
  // anyway (and which the first search of a solve rewrites: nothing of an earlier frame is ever read).
  float4* far_cq;
  uint32_t* far_c;
  int* far_idx;   // [FS_FAR_PASSES][n_src] source points a first half with a list-building pass behind it (fs_far_pass)
  int* far_n;     // found far from every target, far_n[pass] of them
};
// WIDE lists of hard queries (round 5, gs_knn.h: far_write_from_top / block_brute_min_list_multi; cq == NULL: none kept;
// the variants without the far-list machinery above only): cq[s] = (position the list of source point s was made at,
// exactness radius; 0: no list), c[GS_FAR_SLOTS * s ..] = its slots.  Written by whatever serves a hard query (cube
// scan, block-wide pass) in any launch of the solve, checked first thing by the 16-lane group that serves it in every
// later launch; the prep launch of a frame clears every point's radius (nothing of an earlier frame is ever read).
struct IcpHalfWide {
  float4* cq;
  uint32_t* c;
};
// candidate lists of ordinary queries (gs_knn.h: gl_*; lq == NULL: none): lq[s] = (position the list of source point s
// was built at, exactness radius), ls[GL_SLOTS * s ..] = its slots of `sorted`, lstat = failure counters per launch
struct IcpHalfLists {
  float4* lq;
  uint32_t* ls;
  int* lstat;
};

// flags of an entry of the block's list of left-over queries (hard_q)
constexpr int FS_HQ_FAR = (int)0x80000000;   // the query has a far-candidate list (FAR variants)
constexpr int FS_HQ_SLOT = 0x3fffffff;

// index pairs (into [a0..a5, res]) of the 28 accumulated products: 21 upper-triangular a_i a_k, 6 a_i res, res res
// (three bits per entry in two 64-bit literals for the 21 matrix products; entries 21..27 are (i - 21, 6).  A table in
// constant memory would pin two 64-bit addresses in VGPRs across the whole kernel, which sits at the 80-VGPR limit.)
constexpr unsigned long long FS_PA_BITS = 0x591b692449240000ull;  // 0 0 0 0 0 0 1 1 1 1 1 2 2 2 2 3 3 3 4 4 5
constexpr unsigned long long FS_PB_BITS = 0x5b2c76356346c688ull;  // 0 1 2 3 4 5 1 2 3 4 5 2 3 4 5 3 4 5 4 5 5
GS_DEV int fs_pa(int i) { return i < 21 ? (int)((FS_PA_BITS >> (3 * i)) & 7ull) : i - 21; }
GS_DEV int fs_pb(int i) { return i < 21 ? (int)((FS_PB_BITS >> (3 * i)) & 7ull) : 6; }

// One half-iteration for one sequence.  G lanes serve a query, so a block holds NQ = FS_BLOCK / G query slots =
// NU row units of FS_QPB (96) queries.  Every unit produces ONE partial row, always with the same fixed-order sums,
// so the normal equations do not depend on G, on the number of blocks or on which block works on which unit: a
// block owns the contiguous units [lb * upb, (lb + 1) * upb) and walks them NU at a time (grids smaller than the
// unit count are how a GPU shared by 8 sequences keeps every block resident and pays the prologue once per block).
// LISTS (round 4): candidate lists of ordinary queries (gs_knn.h: gl_*), lmode = what this launch does with them:
//   0  nothing (the first search of a solve: the large step of iteration 0 follows)
//   1  every group builds the list of its point behind the search (the look-ahead of iteration 0)
//   2  the lists are tried first: their slots, the list centre and the listed points are fetched BEFORE the prologue
//      waits for the partial rows of the previous launch (the gathers ride behind the row loads and stay in flight
//      across the scalar stage: its barriers order LDS only), so a launch whose lists all prove has no search on its
//      critical path.  A list that gives no proof sends its point to the left-over pass, where a 16-lane group
//      scans the 2x2x2 block and writes a new list.
// Results do not depend on any of it: a proof is exact, everything else is the search that ran before.
template <bool FULL, int G, bool FAR, int LMODE>
GS_DEV void icp_half_body(const IcpHalfSeq& q, const IcpHalfLists& ql, const IcpHalfWide& qw, const GsCount n_src_c, const float dist_thresh,
                          const gs_icp_params& prm, const int it, const int rows_in_reduced, const unsigned lb,
                          const int upb, unsigned long long* __restrict__ tl_arg = nullptr, const float weak_room = 0.0f) {
  constexpr bool LISTS = LMODE != 0;
#ifdef GS_ICP_TIMELINE
  unsigned long long* __restrict__ tl = tl_arg;
#else
  constexpr unsigned long long* tl = nullptr;  // per-block timeline stamps: debugging builds only (-DGS_ICP_TIMELINE)
#endif
  static_assert(!(FAR && LISTS), "the two kinds of candidate lists are not combined");
  constexpr int NQ = FS_BLOCK / G, NU = NQ / FS_QPB;
  // (LMODE 3 = LMODE 1 with four entries per lane: the 8-entry lists the persistent solve reads, gs_icp_persist.h; 2 lanes only)
  constexpr int LK = LMODE == 3 ? 4 : gl_k<G>(), LM = G * LK;   // list entries per lane / per source point
  static_assert(NQ % FS_QPB == 0 && 2 * NU <= FS_BLOCK / GS_WAVE && LM <= GL_SLOTS, "block shape");
  const float* __restrict__ src_in = q.src_in;
  float* __restrict__ src_out = q.src_out;
  const float* __restrict__ tgt = q.tgt;
  const float* __restrict__ tn = q.tn;
  const int* __restrict__ cell_start = q.cell_start;
  const float4* __restrict__ sorted = q.sorted;
  const float4* __restrict__ sorted_n = q.sorted_n;
  float* __restrict__ d2prev = q.d2prev;
  const double* __restrict__ partials_in = q.partials_in;
  double* __restrict__ partials_out = q.partials_out;
  int64_t* __restrict__ out_idx = q.out_idx;
  int32_t* __restrict__ tape_idx = q.tape_idx;

  __shared__ alignas(16) IcpSmall sm;
  __shared__ double S[32];
  __shared__ double sub[FS_BLOCK / 32][32];
  __shared__ unsigned long long keys_s[NQ];
  __shared__ int bslot_s[NQ];     // slot of the winning candidate in the binned arrays
  __shared__ float qs[NQ][3];
  __shared__ float qa_s[NQ][8];   // a0..a5, residual of every query of the block (zero when filtered out)
  __shared__ double sub_s[NU][FS_RG][LIN_NV];
  __shared__ int unres_q[NQ], hard_q[NQ];   // (hard_q: slot | FS_HQ_* flags)
  __shared__ int unres_n, hard_n;
  // (LMODE 2) points whose list gave no proof: re-searched after the check by 16-lane groups (the 2x2x2 block as one flat
  // candidate list, 4 candidates per lane at 65 per block in a mature map: two round trips instead of the eight a 2-lane
  // group needs inline -- a launch with failing lists is as slow as its slowest group)
  __shared__ int fail_q[LMODE == 2 ? NQ : 1];
  __shared__ int fail_n;
  __shared__ unsigned long long red[FS_BLOCK / GS_WAVE];
  __shared__ uint8_t far_s[NQ];   // the query's candidate list proved this search (it keeps its flag)
  // (LMODE 2) source point and (list centre, radius) of every group, parked across the scalar stage of the prologue: its
  // one-lane float64 code needs the registers, the lanes that wait for it do not
  __shared__ float4 park_s[LMODE == 2 ? 2 * NQ : 1];
  __shared__ int lfail_s[3];   // (LMODE 2) lists of this block that gave no proof / were empty / points without a list
  __shared__ uint8_t rowdone_s[LMODE == 2 ? NQ : 1];   // (LMODE 2) the row of this slot was built by the lane that won its list check
  float4* __restrict__ far_cq = q.far_cq;
  uint32_t* __restrict__ far_c = q.far_c;
  // (FAR is a template parameter: the kernels sit at their register limit, and the code of the lists costs the
  // variant without them 0.5 us per launch when it is merely present)
  const bool far_on = FAR && far_cq != nullptr && d2prev != nullptr;
  const int far_pass = (FULL && FAR) ? fs_far_pass(it) : -1;
  float4* __restrict__ lq = LISTS ? ql.lq : nullptr;
  uint32_t* __restrict__ ls = LISTS ? ql.ls : nullptr;

  const int64_t n_src = gs_count(n_src_c), n_tgt = gs_count(q.n_tgt);
  const int nunits = (int)((n_src + FS_QPB - 1) / FS_QPB);
  // rows the previous kernel produced (already added up to one row by gs_icp_reduce_rows_kernel for large solves)
  const int nrows_in = rows_in_reduced ? 1 : nunits;
  const int u_first = (int)lb * upb, u_last = (u_first + upb < nunits) ? u_first + upb : nunits;
  if (u_first >= nunits && lb != 0) return;  // beyond the actual count (bound-sized grid)
  // lists are read by blocks that serve ONE group of units (the host plans it so whenever it asks for them)
  // (LMODE is a template parameter: one kernel per mode keeps each of them within the register budget)
  constexpr bool verify = LMODE == 2;
  constexpr bool build_all = LMODE == 1 || LMODE == 3;
  constexpr bool NPREF = LMODE == 2;   // the list check also fetches the normal of the likely match
  // the source point of the first slot does not depend on the prologue: issue its load first so that
  // the global-memory latency hides behind the scalar stage
  const int lane = threadIdx.x & (G - 1), slot = threadIdx.x / G;
  float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f;
  // the first search of a solve has no predecessor; a list check needs none
  const bool bounded = d2prev != nullptr && !(FULL && it == 0) && !LISTS;   // (the list variants scan whole regions)
  float dprev = __builtin_inff();
  float4 lqv = make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // (verify) list centre and exactness radius of this group's point
  uint32_t sl[LK];                                     // (verify) this lane's slots of the list
  float4 cv[LK];                                       // (verify) the points in those slots
  float4 cn0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // (verify) the normal of the first of them (the list's nearest
                                                       // when it was built: almost always the match again)
  typedef float gs_v4f __attribute__((ext_vector_type(4)));
  gs_v4f stv = {0.0f, 0.0f, 0.0f, 0.0f};             // (verify) this lane's 16 bytes of the state of the previous half
  int st_idx = 0;
  static_assert(sizeof(IcpSmall) % 16 == 0, "state copy");
  const bool st_lane = threadIdx.x >= GS_WAVE && threadIdx.x < GS_WAVE + (int)(sizeof(IcpSmall) / 16);
  if (LMODE == 2) {
    // Everything this variant loads before its prologue is UNCONDITIONAL (clamped addresses, masked afterwards): a load
    // under a branch may not have been issued, so wherever the compiler needs an OLDER load it has to wait for every
    // outstanding one -- and a conditional load next to a zero-initialised register is waited for on the spot.  The
    // copy of the state (the one conditional transfer) therefore goes first.
    // (hand-issued: a DMA into LDS makes the compiler drain the memory queue at every later wait -- it treats the
    // transfer as a flat access that may complete out of order -- and a plain load is sunk to its use, behind the
    // gathers.  The hook below waits for it by count and stores it.)
    // (an opaque copy of the thread index for the state's addresses: 16 x index merged with its uses at the far end of
    // the kernel is what the register allocator would spill -- a store in the memory queue of the prologue)
    st_idx = (int)threadIdx.x - GS_WAVE;
    asm volatile("" : "+v"(st_idx));
    if (st_lane)
      asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(stv) : "v"(reinterpret_cast<const float4*>(q.st_in) + st_idx) : "memory");
    const int64_t s = (int64_t)u_first * FS_QPB + slot;
    const bool vlive = slot / FS_QPB + u_first < u_last && s < n_src;
    const int64_t sc = vlive ? s : 0;   // (source point 0 stands in; masked below)
    p0 = src_in[3 * sc];
    p1 = src_in[3 * sc + 1];
    p2 = src_in[3 * sc + 2];
    lqv = lq[sc];
    const uint32_t* __restrict__ w = ls + GL_SLOTS * sc + LK * lane;
#pragma unroll
    for (int j = 0; j < LK; ++j) sl[j] = w[j];
    // (lanes without a source point carry point 0's values until the hook drops its list; they are not `live` below)
  } else {
#pragma unroll
    for (int j = 0; j < LK; ++j) { sl[j] = ~0u; cv[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); }
    const int64_t s = (int64_t)u_first * FS_QPB + slot;
    if (slot / FS_QPB + u_first < u_last && s < n_src) {
      p0 = src_in[3 * s];
      p1 = src_in[3 * s + 1];
      p2 = src_in[3 * s + 2];
      if (bounded) dprev = d2prev[s];
    }
  }

  // the state of the previous half and the grid header are fetched while the partial rows are summed
  // (their latency is off the critical path; the sums' barriers publish sm)
  // (the variant that tries lists first needs the grid for its left-overs only and fetches the header there: twelve
  // registers less across its prologue)
  GsGrid g = {};
  if (LMODE != 2) g = *q.gp;
  // (global -> LDS directly, 16 bytes per lane of the second wave: a copy through registers made that wave wait for
  // its loads before it had even requested the partial rows, and the block waits for its slowest wave)
  if (LMODE != 2 && st_lane) it_load_lds16(reinterpret_cast<const float4*>(q.st_in) + (threadIdx.x - GS_WAVE), &sm);
  // behind the row loads of the prologue: the listed points (the slots have arrived with the source point: same round
  // trip).  A slot is dereferenced only below the capacity of `sorted` (a list is always written by an earlier launch of
  // the SAME solve; the clamp costs one compare and keeps a stale word from faulting).
  const uint32_t nsl = (uint32_t)(q.n_tgt.host < 0x7fffffffll ? q.n_tgt.host : 0x7fffffffll);
  auto hook = [&]() {
    if (LISTS) {
      if (LMODE == 2 && verify) {
        const bool try_list = (slot / FS_QPB + u_first < u_last && (int64_t)u_first * FS_QPB + slot < n_src) &&
                              p0 == p0 && lqv.w > 0.0f;
#pragma unroll
        for (int j = 0; j < LK; ++j) {
          if (!try_list || !(sl[j] < nsl)) sl[j] = ~0u;
          cv[j] = sorted[sl[j] != ~0u ? sl[j] : 0u];   // (unconditional: slot 0 stands in for an empty entry, never looked at)
        }
        // (look-ahead half only: in the first half the four registers do not fit next to its scalar stage)
        if (NPREF && sorted_n) cn0 = sorted_n[sl[0] != ~0u ? sl[0] : 0u];   // (block-uniform: every lane issues it or none)
        if (lane == 0) {
          int slot_p = slot;   // (opaque: merged with the address of qa_s[slot] at the far end of the kernel it would be spilled)
          asm volatile("" : "+v"(slot_p));
          park_s[2 * slot_p] = make_float4(p0, p1, p2, 0.0f);
          park_s[2 * slot_p + 1] = lqv;
          rowdone_s[slot_p] = 0;
        }
      }
      // The barriers of this variant do not drain the memory queue, so the copy of the state (hand-issued before
      // everything else, or a DMA into LDS in the building variant) is waited for by hand: loads complete in the order
      // of issue, and exactly LK (verify) vector-memory instructions -- the unconditional gathers right above -- have
      // been issued since everything else.  The row sums are consumed next anyway, so this wait costs nothing.
      if (LMODE == 2 && verify) {
        if (NPREF && sorted_n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LK + 1) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LK) : "memory");
        if (st_lane) *reinterpret_cast<gs_v4f*>(reinterpret_cast<float4*>(&sm) + st_idx) = stv;
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
  };

  // ---- prologue: finish the previous half-iteration (identical in every block)
  if (FULL) {
    double e1 = 0.0;
    if (LISTS) {
      // (unconditionally: under `it > 0` the compiler hoists the head of the hook -- the wait for the source point and
      // the list -- in front of the branch, i.e. in front of the row load.  The sum of a first iteration is not used:
      // its rows are whatever the buffer holds.)
      e1 = icp_sum_col27_hook<FS_BLOCK>(partials_in, nrows_in, reinterpret_cast<double*>(red), hook);
    } else if (it > 0) {
      e1 = icp_sum_col27<FS_BLOCK>(partials_in, nrows_in, reinterpret_cast<double*>(red));
    } else {
      __syncthreads();
    }
    if (tl && threadIdx.x == 0) { tl[8] = wall_clock64(); tl[9] = tl[8]; }   // sums done (no solve in this half)
    if (threadIdx.x < GS_WAVE) {  // scalar stage by wave 0, in place on the LDS copy of the state
      if (it > 0)
        icp_update_math_wave((float)e1, sm, prm,
                             (lb == 0 && it - 1 < GS_ICP_MAX_ITERS) ? q.trace + 12 * (it - 1) : nullptr, (int)threadIdx.x);
      if (threadIdx.x == 0) {
        unres_n = 0;
        hard_n = 0;
        fail_n = 0;
        lfail_s[0] = lfail_s[1] = lfail_s[2] = 0;
      }
    }
  } else {
    if (LISTS) icp_sum_rows_split<FS_BLOCK>(partials_in, nrows_in, S, sub, hook);
    else icp_sum_rows<FS_BLOCK>(partials_in, nrows_in, S, sub);
    if (tl && threadIdx.x == 0) tl[8] = wall_clock64();   // sums done
    if (threadIdx.x < GS_WAVE) {   // 6x6 solve and the exponential of the step across the lanes of wave 0
      gs_solve_spd6_wave(S, sm.damp, sm.xi);
      if (tl && threadIdx.x == 0) tl[9] = wall_clock64();   // solve done
      icp_solve_finish_wave(S, sm, (int)threadIdx.x);
      if (threadIdx.x == 0) {
        if (q.tape_sys && lb == 0) tape_write_sys(q.tape_sys, it, S, sm.damp);
        unres_n = 0;
        hard_n = 0;
        fail_n = 0;
        lfail_s[0] = lfail_s[1] = lfail_s[2] = 0;
      }
    }
  }
  gs_bar<LISTS>();
  if (lb == 0 && threadIdx.x < (int)(sizeof(IcpSmall) / 4))
    reinterpret_cast<float*>(q.st_out)[threadIdx.x] = reinterpret_cast<const float*>(&sm)[threadIdx.x];
  if (tl && threadIdx.x == 0) { tl[4] = wall_clock64(); tl[7] = 0; tl[2] = 0; }
  if (LMODE == 2) g = *q.gp;   // (in flight while the lists are checked; re-scans and the left-over pass need it)

  // (the list variants serve ONE group of units per block -- the host plans them only then -- and say so to the
  // compiler: nothing is carried around a loop)
  int u0 = u_first;
  if (u0 < u_last) do {
    const int64_t s = (int64_t)u0 * FS_QPB + slot;
    const bool live = (u0 + slot / FS_QPB < u_last) && s < n_src;  // this slot holds a source point
    if (!LISTS && u0 != u_first) {
      p0 = p1 = p2 = 0.0f;
      dprev = __builtin_inff();
      if (live) {
        p0 = src_in[3 * s];
        p1 = src_in[3 * s + 1];
        p2 = src_in[3 * s + 2];
        if (bounded) dprev = d2prev[s];
      }
    }
    // ---- search: one source point per G-lane group, pending transform applied to the loaded point.
    // A NaN source point (empty slot of an un-compacted lattice, gs_lattice_source_f32) is skipped: it stays
    // NaN through every transform, is never searched and contributes no row.
    if (LMODE == 2 && live) {   // (parked by this group's first lane: same wave, LDS accesses of a wave are ordered)
      int slot_p = slot;   // (opaque, as where it was parked)
      asm volatile("" : "+v"(slot_p));
      const float4 pp = park_s[2 * slot_p];
      p0 = pp.x; p1 = pp.y; p2 = pp.z;
      lqv = park_s[2 * slot_p + 1];
    }
    if (live && p0 != p0) {
      if (lane == 0) {
        if (FULL) { src_out[3 * s] = p0; src_out[3 * s + 1] = p0; src_out[3 * s + 2] = p0; }
        qs[slot][0] = p0; qs[slot][1] = p0; qs[slot][2] = p0;
        keys_s[slot] = ~0ull;
        if (FAR) far_s[slot] = 0;
      }
    } else if (live) {
      const bool has_far = far_on && bounded && (__float_as_uint(dprev) >> 31) != 0u;
      if (far_on) dprev = __builtin_fabsf(dprev);
      const float* T = FULL ? sm.T_step : sm.Tr;
      float qx, qy, qz;
      gs_rigid_fma(T, p0, p1, p2, qx, qy, qz);
      bool done;
      bool refail = false;   // (LMODE 2) the list gave no proof: 16-lane re-search of the 2x2x2 block behind the check
      bool weak = false;     // (list-building launch) the list leaves the neighbour too little room: the cube scans re-make it
      int win;
      unsigned long long key;
      if (LMODE == 2 && verify) {
        // the list: every listed point against the query, the same key order as every other engine
        key = ~0ull;
        int wsl = -1;
        bool first = false;   // this lane's best is its first entry (whose normal it holds)
#pragma unroll
        for (int j = 0; j < LK; ++j) {
          const unsigned long long k2 = sl[j] != ~0u ? grid_key(qx, qy, qz, cv[j]) : ~0ull;
          if (k2 < key) { key = k2; wsl = (int)sl[j]; first = j == 0; }
        }
        const unsigned long long kmin = grid_group_min<G>(key);
        win = (key == kmin && wsl >= 0) ? wsl : -1;
        key = kmin;
        const float bd = __uint_as_float((uint32_t)(key >> 32));   // NaN: empty list
        const float ex = qx - lqv.x, ey = qy - lqv.y, ez = qz - lqv.z;
        const float delta = sqrtf(ex * ex + ey * ey + ez * ez);
        done = lqv.w > 0.0f && sqrtf(bd) + delta < lqv.w * 0.9999f;   // false for NaN
        // the lane that holds the match and its normal builds the Gauss-Newton row right here (the same operations on the
        // same values as the row pass below, which then has nothing to gather for this slot)
        if (NPREF && done && win >= 0 && first && sorted_n) {
          float a[6], res;
          gn_row_pn(qx, qy, qz, cv[0], cn0, a, res);
          const bool keep = (dist_thresh < 0.0f) || (bd < dist_thresh);
#pragma unroll
          for (int i = 0; i < 6; ++i) qa_s[slot][i] = keep ? a[i] : 0.0f;
          qa_s[slot][6] = keep ? res : 0.0f;
          rowdone_s[slot] = 1;
        }
        if (!done) {
          key = ~0ull;
          win = -1;
          // (failure counters: per block in LDS, three global atomics per block below -- thousands of failing groups
          // adding to ONE global word serialise at ~12 ns each, which made a launch with many failures three times as long)
          if (lane == 0) atomicAdd(&lfail_s[lqv.w < 0.0f ? 2 : (lqv.w == 0.0f ? 1 : 0)], 1);
          // No proof from the list: the point is re-searched behind the check by a 16-lane group, which scans the 2x2x2
          // block as one flat candidate list and leaves a new list for where the point is now (round 5; round 4 had the
          // point's own G lanes do it right here: in a mature map the block holds ~65 candidates, eight dependent round
          // trips for two lanes with four gathers in flight, and every block of a launch with failing lists waited for
          // that; the cube scans of the left-over pass, which round 4 tried instead, are slower still).
          // (R < 0: the 2x2x2 stage could not prove this point when it was last tried -- straight to the cube scans)
          refail = lqv.w >= 0.0f;
        }
      } else {
        if (build_all) {
          // the whole 2x2x2 block, every lane remembering its nearest candidates: the list is a by-product of the search
          GlTop<LK> top;
          gl_top_reset<LK>(top);
          float rc2;
          key = grid_search_stage0_top<G, LK>(g, cell_start, sorted, qx, qy, qz, lane, &done, &win, top, &rc2);
          if (done) {
            gl_write_lanes<G, LK>(top, rc2, qx, qy, qz, lane, ls + GL_SLOTS * s, lq + s);
            // A WEAK list: the bound of the 2x2x2 block (half a cell to a cell: 15 - 18 mm) caps the radius, and a source point
            // whose nearest target is 12 - 15 mm away -- its lattice neighbour in the map is a zeroed pixel -- is left with
            // millimetres of room: it loses its proof in every look-ahead of a solve that still moves (5 % of the points of
            // seed 0 / frame 8, each time a re-search pass for its whole block).  Such a point goes through the cube scans
            // below, ONCE: their bound is a whole cell (29 mm), and they leave it a wide list as well.
            if (weak_room > 0.0f) {
              const float out2 = gl_group_minf<G>(top.d[LK]);
              const float R2 = out2 < rc2 ? out2 : rc2;
              const float bdw = __uint_as_float((uint32_t)(key >> 32));
              weak = !(sqrtf(bdw) + weak_room * g.c < sqrtf(R2));
            }
          } else if (lane == 0) lq[s] = make_float4(qx, qy, qz, -1.0f);   // (the cubes below may still give it a list)
        } else {
          // search bound: the previous neighbour of this source point is still a target; the previous query was Tr * p
          // in the look-ahead half (this half: T_step * p) resp. p itself in the first half (this half: Tr * p)
          float rball;
          {
            float ox = p0, oy = p1, oz = p2;
            if (FULL) gs_rigid_fma(sm.Tr, p0, p1, p2, ox, oy, oz);
            const float ex = qx - ox, ey = qy - oy, ez = qz - oz;
            rball = sqrtf(dprev) + sqrtf(ex * ex + ey * ey + ez * ez);  // inf without a predecessor, NaN after a NaN match
          }
          key = grid_search_stage0<G>(g, cell_start, sorted, qx, qy, qz, lane, &done, &win, rball);
        }
      }
      if (win >= 0 || (lane == 0 && key == ~0ull)) bslot_s[slot] = win;  // one writer: the winning lane
      if (lane == 0) {
        if (FULL) {  // the transformed cloud of this iteration
          src_out[3 * s] = qx;
          src_out[3 * s + 1] = qy;
          src_out[3 * s + 2] = qz;
        }
        qs[slot][0] = qx; qs[slot][1] = qy; qs[slot][2] = qz;
        keys_s[slot] = key;
        if (FAR) far_s[slot] = 0;
        if (!done || weak) {
          if (LMODE == 2 && refail) fail_q[atomicAdd(&fail_n, 1)] = slot;
          else hard_q[atomicAdd(&hard_n, 1)] = slot | (has_far ? FS_HQ_FAR : 0);
        }
      }
    }
    __syncthreads();
    if (tl && threadIdx.x == 0 && u0 == u_first) tl[10] = wall_clock64();   // list check / search done
    if (LMODE == 2) {
      const int nf = fail_n;   // block-uniform
      if (tl && threadIdx.x == 0 && u0 == u_first) tl[12] = (unsigned long long)nf;
      // lanes per re-searched point by how many there are (block-uniform): 32 lanes see the ~65 candidates of a mature map
      // in two or three round trips (up to 24 points in one round of groups); beyond that 8 lanes with two gathers in
      // flight each (four or five round trips, 96 points per round: the look-aheads of a solve that wanders lose 50 - 200
      // lists per block, profiles/r05_c_failing_lookahead_timeline.txt; 16 lanes served 48 per round in as many trips)
      auto research = [&](auto fg_tag) {
        constexpr int FG = decltype(fg_tag)::value;
        for (int i = threadIdx.x / FG; i < nf; i += FS_BLOCK / FG) {
          const int hs = fail_q[i], lf = threadIdx.x & (FG - 1);
          const float hx = qs[hs][0], hy = qs[hs][1], hz = qs[hs][2];
          const int64_t sq = (int64_t)u0 * FS_QPB + hs;
          constexpr int KF = 2;   // candidates a lane remembers for the new list (the M nearest of 2 FG)
          GlTop<KF> top;
          gl_top_reset<KF>(top);
          float rc2;
          bool fdone;
          int fwin;
          const unsigned long long fkey = grid_search_stage0_top<FG, KF, (FG >= 16 ? 1 : 2)>(g, cell_start, sorted, hx, hy, hz, lf, &fdone, &fwin, top, &rc2);
          if (fdone) gl_select_write<FG, KF>(top, rc2, hx, hy, hz, lf, LM, ls + GL_SLOTS * sq, lq + sq);
          else if (lf == 0) lq[sq] = make_float4(hx, hy, hz, -1.0f);   // (the cubes below may still give it a list)
          if (fwin >= 0) bslot_s[hs] = fwin;   // one writer: the winning lane
          if (lf == 0) {
            keys_s[hs] = fkey;
            if (!fdone) hard_q[atomicAdd(&hard_n, 1)] = hs;
          }
        }
      };
      if (nf > FS_BLOCK / 32) research(std::integral_constant<int, 8>{});
      else if (nf) research(std::integral_constant<int, 32>{});
      if (nf) {
        __syncthreads();
        if (threadIdx.x == 0) fail_n = 0;
        __syncthreads();
      }
    }
    if (tl && threadIdx.x == 0 && u0 == u_first) tl[11] = wall_clock64();   // failed lists re-searched
    if (LMODE == 2 && threadIdx.x < 3 && ql.lstat) {   // failure counters of this launch (diagnostics)
      const int hl = 2 * it + (FULL ? 0 : 1);         // launch index within the solve
      const int c = lfail_s[threadIdx.x];
      if (c && hl < GL_STAT_LAUNCHES) atomicAdd(ql.lstat + threadIdx.x * GL_STAT_LAUNCHES + hl, c);
    }
    // ---- the few queries the 2x2x2 stage did not resolve (neighbour farther than ~half a cell): Chebyshev shells
    // by groups of FS_HG lanes, so that they do not hold up the waves of the common case
    const int nh = hard_n;  // block-uniform
    // wide lists of hard queries (gs_knn.h; IcpHalfWide): kept by the variants with ordinary lists only -- the plain variants
    // (the first three launches of a solve) are at their scalar-register limit, where two more pointers spill scalars into
    // vector registers and those to memory; the pointers are read here, where the rare path starts, not at the top of the
    // kernel
    constexpr bool WL = !FAR && LISTS;
    float4* __restrict__ wl_cq = WL ? qw.cq : nullptr;
    uint32_t* __restrict__ wl_c = WL ? qw.c : nullptr;
    const bool wl_on = WL && wl_cq != nullptr;
    for (int i = threadIdx.x / FS_HG; i < nh; i += FS_BLOCK / FS_HG) {
      const int e = hard_q[i], hs = e & FS_HQ_SLOT, l16 = threadIdx.x & (FS_HG - 1);
      const float hx = qs[hs][0], hy = qs[hs][1], hz = qs[hs][2];
      const int64_t sq = (int64_t)u0 * FS_QPB + hs;   // the query's source point
      bool done = false, listed = false;
      int win = -1;
      unsigned long long key = keys_s[hs];
      if (FAR && e < 0) {   // (only with far_on) the list of an earlier search of this solve: exact while the query stays close
        const float4 c0R = far_cq[sq];
        const unsigned long long kl = far_list_search<FS_HG>(c0R, far_c + GS_FAR_SLOTS * sq, sorted, hx, hy, hz, l16, &done, &win);
        if (done) { key = kl; listed = true; }
        else win = -1;
      }
      if (WL && wl_on) {   // the wide list an earlier launch of this solve left for the point, if any
        const float4 c0R = wl_cq[sq];
        if (c0R.w > 0.0f) {   // (group-uniform)
          const unsigned long long kl = wide_list_search<FS_HG>(c0R, wl_c + GS_FAR_SLOTS * sq, sorted, hx, hy, hz, l16, &done, &win);
          if (done) { key = kl; listed = true; }
          else win = -1;
        }
      }
      constexpr int KH = 2;   // candidates a lane of the 16-lane group remembers for the new list
      GlTop<KH> top;
      float rc2 = 0.0f;
      if (!done) {
        int kdone = 0;
        if (LISTS) {
          key = grid_search_rings_top<FS_HG, KH>(g, cell_start, sorted, hx, hy, hz, l16, key, &done, &win, FS_HARD_RINGS, top, &rc2);
        }
        else key = grid_search_rings<FS_HG>(g, cell_start, sorted, hx, hy, hz, l16, key, &done, &win, FS_HARD_RINGS, &kdone);
        // first halves with a list-building pass behind them (fs_far_pass): queries that needed a cube of radius >=
        // FS_FAR_MIN_RING are handed to gs_icp_far_build_kernel together with what the search found (squared distance,
        // radius of the last cube); the ones left to the brute-force pass follow below
        if (FULL && FAR && far_pass >= 0 && far_on && l16 == 0 && done && kdone >= FS_FAR_MIN_RING) {
          far_cq[sq] = make_float4(__uint_as_float((uint32_t)(key >> 32)), (float)kdone, 0.0f, 0.0f);
          q.far_idx[(int64_t)far_pass * n_src + atomicAdd(q.far_n + far_pass, 1)] = (int)sq;
        }
      }
      // (list variants) the scan that served the point leaves its list; what only the brute-force pass can serve keeps none
      if (LISTS && !(WL && listed)) {   // (a point its wide list served keeps what it has: no ordinary list, R < 0)
        if (done) {
          // (whatever the lanes remember of the proving cube is the point's wide list from now on)
          if (WL && wl_on) far_write_from_top<FS_HG, KH>(top, rc2, hx, hy, hz, l16, wl_c + GS_FAR_SLOTS * sq, wl_cq + sq);
          gl_select_write<FS_HG, KH>(top, rc2, hx, hy, hz, l16, LM, ls + GL_SLOTS * sq, lq + sq);
        } else if (l16 == 0) lq[sq] = make_float4(hx, hy, hz, -1.0f);
      }
      if (win >= 0) bslot_s[hs] = win;  // a candidate of the list / the cubes beat the 2x2x2 stage
      if (l16 == 0) {
        keys_s[hs] = key;
        if (FAR && listed) far_s[hs] = 1;
        if (!done) unres_q[atomicAdd(&unres_n, 1)] = hs;
      }
    }
    if (nh) {
      __syncthreads();
      if (threadIdx.x == 0) hard_n = 0;
      __syncthreads();
    }
    if (tl && threadIdx.x == 0 && u0 == u_first) { tl[5] = wall_clock64(); tl[2] += (unsigned long long)nh; }
    const int nun = unres_n;  // block-uniform
    if (tl && threadIdx.x == 0) { tl[7] += (unsigned long long)nun; if (u0 == u_first) tl[6] = wall_clock64(); }
    // FS_BQ queries per pass over the binned targets; with wide lists the pass also leaves the lists of its queries
    constexpr int BQL = 2;   // (two queries per list-leaving pass: four cost the 4-lane look-ahead variant a spill; such a pass
                             // now happens once per solve and far point, not once per launch)
    if (WL && wl_on) {   // (block-uniform)
      for (int u = 0; u < nun; u += BQL)
        block_brute_min_list_multi<FS_BLOCK, BQL>(qs, unres_q + u, nun - u < BQL ? nun - u : BQL, sorted, cell_start[g.ncell],
                                                  keys_s, bslot_s, (int64_t)u0 * FS_QPB, wl_cq, wl_c);
    } else {
      for (int u = 0; u < nun; u += FS_BQ)
        block_brute_min_sorted_multi<FS_BLOCK, FS_BQ>(qs, unres_q + u, nun - u < FS_BQ ? nun - u : FS_BQ, sorted,
                                                      cell_start[g.ncell], keys_s, bslot_s);
    }
    if (nun) {
      __syncthreads();
      if (FULL && FAR && far_pass >= 0 && far_on && (int)threadIdx.x < nun) {   // (see the cube searches above; -1: no cube)
        const int hs = unres_q[threadIdx.x];
        const int64_t sq = (int64_t)u0 * FS_QPB + hs;
        far_cq[sq] = make_float4(__uint_as_float((uint32_t)(keys_s[hs] >> 32)), -1.0f, 0.0f, 0.0f);
        q.far_idx[(int64_t)far_pass * n_src + atomicAdd(q.far_n + far_pass, 1)] = (int)sq;
      }
      __syncthreads();
      if (threadIdx.x == 0) unres_n = 0;
      __syncthreads();
    }

    // ---- Gauss-Newton row of every query: its own group's first lane gathers the match and leaves [a, res] in LDS
    if (lane == 0) {
      // (list variants: the index of the source point is formed again here; kept from the top of the kernel it is the
      // one value the register allocator spills -- the opaque copy keeps the compiler from merging the two)
      int u0r = u0;
      if (LISTS) asm volatile("" : "+v"(u0r));   // (a vector register: the unit index is a quotient, i.e. vector arithmetic)
      const int64_t s = LISTS ? (int64_t)u0r * FS_QPB + slot : (int64_t)u0 * FS_QPB + slot;
      float a[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}, res = 0.0f;
      bool rowdone = false;
      if (live && qs[slot][0] != qs[slot][0]) {  // skipped (NaN) source point
        if (FULL && out_idx) out_idx[s] = -1;
        if (tape_idx) tape_idx[s] = -1;
      } else if (live) {
        const unsigned long long bb = keys_s[slot];
        // NaN bits when nothing was found; with candidate lists the sign bit says "source point s has one"
        if (d2prev) {
          uint32_t db = (uint32_t)(bb >> 32);
          if (far_on) db = (db & 0x7fffffffu) | (far_s[slot] ? 0x80000000u : 0u);
          d2prev[s] = __uint_as_float(db);
        }
        int64_t j = (int64_t)(bb & 0xffffffffull);
        if (j >= n_tgt) j = 0;  // only when every distance was NaN
        const float d2 = __uint_as_float((uint32_t)(bb >> 32));
        const bool keep = (dist_thresh < 0.0f) || (d2 < dist_thresh);
        const int bsl = bslot_s[slot];
        if (NPREF && rowdone_s[slot]) rowdone = true;   // (built by the lane that won the list check)
        else if (sorted_n && bsl >= 0 && bb != ~0ull)   // matched point and normal from the binned copies (same bits)
          gn_row_pn(qs[slot][0], qs[slot][1], qs[slot][2], sorted[bsl], sorted_n[bsl], a, res);
        else
          gn_row(qs[slot][0], qs[slot][1], qs[slot][2], tgt, tn, j, a, res);
        if (FULL && out_idx) out_idx[s] = j;
        if (tape_idx) tape_idx[s] = keep ? (int32_t)j : -1;
        if (!keep) {
#pragma unroll
          for (int i = 0; i < 6; ++i) a[i] = 0.0f;
          res = 0.0f;
        }
      }
      if (!rowdone) {
#pragma unroll
        for (int i = 0; i < 6; ++i) qa_s[slot][i] = a[i];
        qa_s[slot][6] = res;
      }
    }
    __syncthreads();
    if (!FULL) {  // residual only: per unit the wave-sum tree over its 96 values (64 + 32), then the two wave sums
      double* red2 = reinterpret_cast<double*>(red);
      const int wave = threadIdx.x / GS_WAVE, wl = threadIdx.x & (GS_WAVE - 1);
      if (wave < 2 * NU) {
        const int r = (wave & 1) * GS_WAVE + wl;   // row within the unit
        double rr = 0.0;
        if (r < FS_QPB) {
          const float res = qa_s[(wave >> 1) * FS_QPB + r][6];
          rr = (double)res * (double)res;
        }
        const double sum = gs_wave_sum_f64(rr);
        if (wl == 0) red2[wave] = sum;
      }
      __syncthreads();
      if (threadIdx.x < NU && u0 + (int)threadIdx.x < u_last)
        partials_out[(int64_t)(u0 + threadIdx.x) * LIN_NV + 27] = red2[2 * threadIdx.x] + red2[2 * threadIdx.x + 1];
      __syncthreads();
      continue;
    }
    // FS_QPB rows x 28 values per unit: FS_RG groups of 28 threads add the products of 4 rows each, then 28
    // threads add the sub-sums, always in index order.
    for (int w = threadIdx.x; w < NU * FS_RG * LIN_NV; w += FS_BLOCK) {
      const int i = w % LIN_NV, part = (w / LIN_NV) % FS_RG, un = w / (LIN_NV * FS_RG);
      const int ia = fs_pa(i), ib = fs_pb(i);
      const float* r0 = qa_s[un * FS_QPB + FS_RPG * part];
      double t = (double)r0[ia] * (double)r0[ib];
#pragma unroll
      for (int u = 1; u < FS_RPG; ++u) t += (double)r0[8 * u + ia] * (double)r0[8 * u + ib];
      sub_s[un][part][i] = t;
    }
    __syncthreads();
    if (threadIdx.x < NU * LIN_NV) {
      const int i = threadIdx.x % LIN_NV, un = threadIdx.x / LIN_NV;
      if (u0 + un < u_last) {
        double t = sub_s[un][0][i];
#pragma unroll
        for (int k = 1; k < FS_RG; ++k) t += sub_s[un][k][i];
        partials_out[(int64_t)(u0 + un) * LIN_NV + i] = t;
      }
    }
    __syncthreads();
  } while (!LISTS && (u0 += NU) < u_last);
}

// Batched: block b works for sequence b % B (with B = 8 a sequence lives on one XCD: its binned targets, cell table
// and partial rows stay in that XCD's L2) as that sequence's block b / B; when B divides 8 the 8 / B XCDs of a
// sequence each own a contiguous range of its query rows.
struct IcpHalfBatch {
  int B;
  int upb;  // row units per block
  float weak_room;   // (list-building launch) cells of room between a point's neighbour and the radius of its list below which
                     // the list is re-made by the cube scans (0: never)
  unsigned long long* timeline;  // debugging aid (GRADSLAM_HIP_ICP_TIMELINE): per block [start, end, hw id, xcc id]
  IcpHalfSeq s[GS_MAX_BATCH];
  IcpHalfLists l[GS_MAX_BATCH];   // (read by the list variants only)
  IcpHalfWide w[GS_MAX_BATCH];    // (read in the left-over pass of the variants without far lists only)
};
template <bool FULL, int G, bool FAR, int LMODE>
__global__ void __launch_bounds__(FS_BLOCK, 6) gs_icp_half_batch_kernel(const IcpHalfBatch hb, GsCount n_src_c,
                                                                        float dist_thresh, gs_icp_params prm, int it,
                                                                        int rows_in_reduced) {
  const unsigned B = (unsigned)hb.B, blk = blockIdx.x / B, nblk = gridDim.x / B;
  const unsigned X = (GS_XCDS % B == 0) ? GS_XCDS / B : 1u;
#ifdef GS_ICP_TIMELINE
  unsigned long long t0 = 0;
  if (hb.timeline && threadIdx.x == 0) t0 = wall_clock64();
#endif
  icp_half_body<FULL, G, FAR, LMODE>(hb.s[blockIdx.x % B], hb.l[blockIdx.x % B], hb.w[blockIdx.x % B], n_src_c, dist_thresh, prm, it, rows_in_reduced,
                                     gs_xcd_block(blk, nblk, X), hb.upb,
                                     hb.timeline ? hb.timeline + 72 * (size_t)blockIdx.x : nullptr, hb.weak_room);
#ifdef GS_ICP_TIMELINE
  if (hb.timeline && threadIdx.x == 0) {
    unsigned long long* r = hb.timeline + 72 * (size_t)blockIdx.x;
    r[0] = t0;
    r[1] = wall_clock64();
  }
#endif
}

#include "gs_icp_persist.h"

static size_t icp_rows(int64_t n_src) { return (size_t)gs_ceil_div(n_src, FS_QPB); }  // >= ceil(n_src / LIN_BLOCK)

// Launch geometry of a half-iteration: lanes per query G, blocks per sequence nb and row units per block upb.
// Every sequence of the batch gets an equal share of the blocks; the largest G whose blocks all fit that share wins
// (more lanes per query = shorter searches, but only while every query of the sequence is in flight at once); if not
// even G = 2 fits, blocks walk several unit groups.  The share is ONE block per CU when some G fits it (round 6: every
// block pays the prologue -- row sums, float64 scalar stage -- and two of them on a CU pay it in each other's way:
// 4 lanes instead of 8 at two sequences per GPU is +11 %, 2 instead of 4 at four sequences +6 %, same bits), two per CU
// otherwise (8 sequences of 640x480: 400 blocks at 2 lanes per point).
// GRADSLAM_HIP_ICP_LANES = 2 | 4 | 8 forces G, GRADSLAM_HIP_ICP_BLOCKS_PER_CU the share (A/B runs: the results do not
// depend on either).
struct IcpHalfPlan {
  int G, nb, upb;
};
static IcpHalfPlan icp_half_plan(int64_t n_src, int B, int g_max = 8) {
  static int cus = 0, forced = -1, per_cu = 0;   // (per_cu 0: one block per CU if some G fits that, else two)
  if (cus == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
      cus = v;
    else
      cus = 256;
  }
  if (forced < 0) {
    const char* e = getenv("GRADSLAM_HIP_ICP_LANES");
    forced = e ? atoi(e) : 0;
    const char* f = getenv("GRADSLAM_HIP_ICP_BLOCKS_PER_CU");  // resident-block budget per CU (experiments)
    if (f && atoi(f) > 0) per_cu = atoi(f);
  }
  const int nunits = (int)icp_rows(n_src);
  int share = per_cu;
  if (share == 0) {
    const int NU2 = FS_BLOCK / 2 / FS_QPB;   // (the fewest blocks a single group of units per block needs: G = 2)
    share = (nunits + NU2 - 1) / NU2 <= cus / B ? 1 : 2;
  }
  const int budget = (share * cus) / B > 0 ? (share * cus) / B : 1;
  IcpHalfPlan pl{2, 1, 1};
  for (int G = g_max; G >= 2; G >>= 1) {
    const int NU = FS_BLOCK / G / FS_QPB, need = (nunits + NU - 1) / NU;
    if ((forced == 0 && need <= budget) || forced == G || (forced == 0 && G == 2) || (forced > g_max && G == g_max)) {
      pl.G = G;
      pl.nb = need <= budget ? need : budget;
      pl.upb = NU * ((nunits + NU * pl.nb - 1) / (NU * pl.nb));
      pl.nb = (nunits + pl.upb - 1) / pl.upb;
      break;
    }
  }
  return pl;
}
static unsigned long long* g_icp_tl_buf = nullptr;
// lmode: what the launch does with the candidate lists of ordinary queries (icp_half_body; 0 = the variant without them)
template <bool FULL>
static void icp_half_launch(const IcpHalfPlan& pl, IcpHalfBatch& hb, GsCount n_src_c, const gs_icp_params* prm, int it,
                            int rows_in_reduced, hipStream_t st, int lmode = 0) {
  hb.upb = pl.upb;
  // debugging aid (library built with -DGS_ICP_TIMELINE): per-block time stamps of the LAST iteration's two launches,
  // the first half into <path>, the look-ahead that follows it back to back into <path>.next (their first block starts
  // are one launch period apart); both are written after the second launch
  static const char* tl_path = getenv("GRADSLAM_HIP_ICP_TIMELINE");
  unsigned long long*& tl_buf = g_icp_tl_buf;   // (one buffer for both instantiations of this template)
  constexpr size_t TL_HALF = 72 * 7000;
  hb.timeline = nullptr;
  const size_t tl_n = 72 * (size_t)hb.B * pl.nb;
  static const char* tl_it_env = getenv("GRADSLAM_HIP_ICP_TIMELINE_IT");   // the iteration recorded (default: the last)
  const int tl_it = tl_it_env ? atoi(tl_it_env) : prm->numiters - 1;
  const bool tl = tl_path && it == tl_it;
  if (tl) {
    if (!tl_buf && hipMalloc(&tl_buf, 8 * 2 * TL_HALF) != hipSuccess) tl_buf = nullptr;
    if (tl_buf && tl_n <= TL_HALF) {
      hb.timeline = tl_buf + (FULL ? 0 : TL_HALF);
      if (FULL) (void)hipMemsetAsync(tl_buf, 0, 8 * 2 * TL_HALF, st);
    }
  }
  const dim3 grid((unsigned)(hb.B * pl.nb)), block(FS_BLOCK);
  const bool far = hb.s[0].far_cq != nullptr;   // candidate lists of far queries: the same for all sequences of a batch
#define GS_HALF_LAUNCH(G_, FAR_, LMODE_)                                                                               \
  hipLaunchKernelGGL((gs_icp_half_batch_kernel<FULL, G_, FAR_, LMODE_>), grid, block, 0, st, hb, n_src_c,              \
                     prm->dist_thresh, *prm, it, rows_in_reduced)
#define GS_HALF_LAUNCH_L(G_)                                                 \
  do {                                                                       \
    if (lmode == 2) GS_HALF_LAUNCH(G_, false, 2);                            \
    else if (lmode == 1) GS_HALF_LAUNCH(G_, false, 1);                       \
    else if (lmode == 3 && G_ == 2) GS_HALF_LAUNCH(2, false, 3);             \
    else GS_HALF_LAUNCH(G_, false, 0);                                       \
  } while (0)
  // (no 8-lane variant with far-candidate lists: its first half does not fit the register budget; localize_chunk plans
  // solves with such lists for at most 4 lanes per query)
  const bool lists = lmode != 0 && !far && hb.l[0].lq != nullptr;
  if (!lists) lmode = 0;
  if (pl.G == 8) GS_HALF_LAUNCH_L(8);
  else if (pl.G == 4) { if (far) GS_HALF_LAUNCH(4, true, 0); else GS_HALF_LAUNCH_L(4); }
  else { if (far) GS_HALF_LAUNCH(2, true, 0); else GS_HALF_LAUNCH_L(2); }
#undef GS_HALF_LAUNCH_L
#undef GS_HALF_LAUNCH
  if (hb.timeline && !FULL) {  // debugging aid: synchronous dump of the two launches' block records
    std::unique_ptr<unsigned long long[]> h(new unsigned long long[2 * TL_HALF]);
    if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(h.get(), tl_buf, 8 * 2 * TL_HALF, hipMemcpyDeviceToHost) == hipSuccess) {
      for (int part = 0; part < 2; ++part) {
        char path[1024];
        snprintf(path, sizeof(path), "%s%s", tl_path, part ? ".next" : "");
        FILE* f = fopen(path, "w");
        if (!f) continue;
        fprintf(f, "# B=%d G=%d nb=%d upb=%d lmode=%d: block start end(100MHz ticks) hw_id xcc_id after_prologue after_search after_unres n_unres\n", hb.B, pl.G, pl.nb, pl.upb, lmode);
        for (size_t i = 0; i < tl_n / 72; ++i) {
          fprintf(f, "%zu", i);
          for (int k = 0; k < 72; ++k) fprintf(f, " %llu", h[part * TL_HALF + 72 * i + k]);
          fprintf(f, "\n");
        }
        fclose(f);
      }
    }
  }
  hb.timeline = nullptr;
}

// Large solves (more rows than FS_REDUCE_ROWS): every block of the next kernel adding up all rows is
// O(rows^2) L2 traffic.  One extra single-block launch adds them up once, in the same order, into a one-row
// buffer.  The threshold sits above the 815 rows of a 1296x968 frame at dsratio 4: there the 20 extra launches per
// solve cost more than the 182 KB of rows per block (measured 2 % of the frame).
constexpr int FS_REDUCE_ROWS = 1024;
__global__ void __launch_bounds__(FS_BLOCK) gs_icp_reduce_rows_kernel(const double* __restrict__ partials_in,
                                                                      GsCount n_src_c, double* __restrict__ row_out) {
  __shared__ double S[32];
  __shared__ double sub[FS_BLOCK / 32][32];
  const int nrows = (int)((gs_count(n_src_c) + FS_QPB - 1) / FS_QPB);
  icp_sum_rows<FS_BLOCK>(partials_in, nrows, S, sub);
  if (threadIdx.x < LIN_NV) row_out[threadIdx.x] = S[threadIdx.x];
}

// After the last look-ahead: final LM / gradLM update and the (composed) result.
__global__ void __launch_bounds__(FS_BLOCK) gs_icp_finish_kernel(const double* __restrict__ partials_in, GsCount n_src_c,
                                                                 GsIcpState* __restrict__ st, int buf,
                                                                 gs_icp_params prm, const float* __restrict__ compose16,
                                                                 float* __restrict__ out_T16) {
  __shared__ double red[FS_BLOCK / GS_WAVE];
  const int nrows_in = (int)((gs_count(n_src_c) + FS_QPB - 1) / FS_QPB);
  const double e1 = icp_sum_col27<FS_BLOCK>(partials_in, nrows_in, red);
  if (threadIdx.x != 0) return;
  IcpSmall sm = st->s[buf];
  const int it = prm.numiters - 1;
  icp_update_math((float)e1, sm, prm, it < GS_ICP_MAX_ITERS ? st->trace + 12 * it : nullptr);
  st->s[buf ^ 1] = sm;
  icp_write_result(sm, compose16, out_T16);
}

// ---------------------------------------------------------------- brute-force path ------
constexpr int LIN_BLOCK = 256;
constexpr int SUM_BLOCK = 1024;

// Reads the KNN result of every source point (and re-arms best[] for the next search), builds
// its row and reduces the normal equations.  FULL = false: residual only (look-ahead).
template <bool FULL>
__global__ void __launch_bounds__(LIN_BLOCK) gs_icp_linearize_kernel(
    const float* __restrict__ src, const float* __restrict__ Tapply, int64_t n_src,
    const float* __restrict__ tgt, const float* __restrict__ tn, int64_t n_tgt,
    unsigned long long* __restrict__ best, float dist_thresh, double* __restrict__ partials,
    int64_t* __restrict__ out_idx, int32_t* __restrict__ tape_idx) {
  __shared__ double red[LIN_BLOCK / GS_WAVE][LIN_NV];
  const int64_t s = (int64_t)blockIdx.x * LIN_BLOCK + threadIdx.x;
  double v[LIN_NV];
#pragma unroll
  for (int i = 0; i < LIN_NV; ++i) v[i] = 0.0;
  if (s < n_src) {
    const unsigned long long bb = best[s];
    best[s] = ~0ull;
    int64_t j = (int64_t)(bb & 0xffffffffull);
    if (j >= n_tgt) j = 0;  // only when every distance was NaN
    const float d2 = __uint_as_float((uint32_t)(bb >> 32));
    const bool keep = (dist_thresh < 0.0f) || (d2 < dist_thresh);
    float p0 = src[3 * s], p1 = src[3 * s + 1], p2 = src[3 * s + 2];
    if (Tapply) {
      float T[12];
#pragma unroll
      for (int i = 0; i < 12; ++i) T[i] = Tapply[i];
      float q0, q1, q2;
      gs_rigid_fma(T, p0, p1, p2, q0, q1, q2);
      p0 = q0; p1 = q1; p2 = q2;
    }
    float a[6], r;
    gn_row(p0, p1, p2, tgt, tn, j, a, r);
    if (FULL && out_idx) out_idx[s] = j;
    if (tape_idx) tape_idx[s] = keep ? (int32_t)j : -1;
    if (keep) {
      if (FULL) {
        int q = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
          for (int k = i; k < 6; ++k) v[q++] = (double)a[i] * (double)a[k];
#pragma unroll
        for (int i = 0; i < 6; ++i) v[21 + i] = (double)a[i] * (double)r;
      }
      v[27] = (double)r * (double)r;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = FULL ? 0 : 27; i < LIN_NV; ++i) {
    const double sum = gs_wave_sum_f64(v[i]);
    if (lane == 0) red[wave][i] = sum;
  }
  __syncthreads();
  if (threadIdx.x < LIN_NV && (FULL || threadIdx.x == 27)) {
    const int i = threadIdx.x;
    partials[(int64_t)blockIdx.x * LIN_NV + i] = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
  }
}

__global__ void __launch_bounds__(SUM_BLOCK) gs_icp_solve_kernel(const double* __restrict__ partials, int nrows,
                                                                 GsIcpState* __restrict__ st, int it,
                                                                 float* __restrict__ tape_sys) {
  __shared__ double S[32];
  __shared__ double sub[SUM_BLOCK / 32][32];
  __shared__ float xi_s[8];
  icp_sum_rows<SUM_BLOCK>(partials, nrows, S, sub);
  if (threadIdx.x >= GS_WAVE) return;
  gs_solve_spd6_wave(S, st->s[0].damp, xi_s);  // same wave: LDS accesses of a wave are ordered
  __builtin_amdgcn_wave_barrier();
  if (threadIdx.x != 0) return;
  IcpSmall sm = st->s[0];
  if (tape_sys) tape_write_sys(tape_sys, it, S, sm.damp);
  for (int k = 0; k < 6; ++k) sm.xi[k] = xi_s[k];
  icp_solve_finish(S, sm);
  st->s[0] = sm;
}

__global__ void __launch_bounds__(SUM_BLOCK) gs_icp_update_kernel(const double* __restrict__ partials, int nrows,
                                                                  GsIcpState* __restrict__ st, gs_icp_params prm,
                                                                  int it, const float* __restrict__ compose16,
                                                                  float* __restrict__ out_T16) {
  __shared__ double red[SUM_BLOCK / GS_WAVE];
  const double e1 = icp_sum_col27<SUM_BLOCK>(partials, nrows, red);
  if (threadIdx.x != 0) return;
  IcpSmall sm = st->s[0];
  icp_update_math((float)e1, sm, prm, it < GS_ICP_MAX_ITERS ? st->trace + 12 * it : nullptr);
  st->s[0] = sm;
  if (it == prm.numiters - 1) icp_write_result(sm, compose16, out_T16);
}

// ---------------------------------------------------------------- host side ------------
__global__ void gs_icp_init_kernel(GsIcpState* __restrict__ st, const float* __restrict__ init16, float damp,
                                   int numiters, const float* __restrict__ compose16, float* __restrict__ out_T16) {
  if (threadIdx.x != 0) return;
  IcpSmall sm;
  for (int i = 0; i < 16; ++i) {
    sm.T_total[i] = init16[i];
    sm.T_step[i] = init16[i];  // the first search applies the initial transform
    sm.Tr[i] = (i % 5 == 0) ? 1.0f : 0.0f;
  }
  for (int i = 0; i < 8; ++i) sm.xi[i] = 0.0f;
  sm.damp = damp;
  sm.err = 0.0f;
  sm.pad[0] = sm.pad[1] = 0.0f;
  st->s[0] = sm;
  st->s[1] = sm;
  if (numiters == 0) icp_write_result(sm, compose16, out_T16);  // degenerate: the (composed) initial transform
}

struct IcpScratch {
  GsIcpState* state;
  double* rowred;  // one partial row (large solves: gs_icp_reduce_rows_kernel)
  unsigned long long* best;
  float* srcA;
  float* srcB;
  double* partials[2];
  void* grid;
};
static IcpScratch icp_carve(void* scratch, int64_t n_src) {
  char* p = reinterpret_cast<char*>(scratch);
  IcpScratch s;
  s.state = reinterpret_cast<GsIcpState*>(p); p += gs_align(sizeof(GsIcpState));
  s.rowred = reinterpret_cast<double*>(p); p += gs_align(sizeof(double) * 32);
  s.best = reinterpret_cast<unsigned long long*>(p); p += gs_align(8 * (size_t)n_src);
  s.srcA = reinterpret_cast<float*>(p); p += gs_align(12 * (size_t)n_src);
  s.srcB = reinterpret_cast<float*>(p); p += gs_align(12 * (size_t)n_src);
  for (int k = 0; k < 2; ++k) {
    s.partials[k] = reinterpret_cast<double*>(p);
    p += gs_align(sizeof(double) * LIN_NV * icp_rows(n_src));
  }
  s.grid = p;
  return s;
}

// GRADSLAM_HIP_KNN=brute forces the brute-force engine (A/B runs; results are identical).
static bool icp_grid_enabled() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("GRADSLAM_HIP_KNN");
    v = (e && strcmp(e, "brute") == 0) ? 0 : 1;
  }
  return v == 1;
}

extern "C" int64_t gs_icp_scratch_bytes(int64_t n_src, int64_t n_tgt) {
  if (n_src < 1) n_src = 1;
  return (int64_t)(gs_align(sizeof(GsIcpState)) + gs_align(sizeof(double) * 32) + gs_align(8 * (size_t)n_src) + 2 * gs_align(12 * (size_t)n_src) +
                   2 * gs_align(sizeof(double) * LIN_NV * icp_rows(n_src)) + gs_knn_grid_scratch_bytes(n_src, n_tgt) +
                   4096);
}

// the per-iteration trace [err, new_err, damp_after, sigmoid, xi(6), 0, 0] completes the tape
static int icp_tape_finish(void* tape, const GsIcpState* state, int64_t n_src, int numiters, hipStream_t st) {
  GsIcpTape t = gs_icp_tape_carve(tape, n_src, numiters);
  if (numiters > 0)
    GS_HIP(hipMemcpyAsync(t.trace, state->trace, sizeof(float) * 12 * (size_t)numiters, hipMemcpyDeviceToDevice, st));
  return GS_OK;
}

// Algorithmic (compulsory) bytes of the 2 x numiters half-iteration kernels of one solve: per slot of
// the source array 12 B read; per searched query 12 B written (first half: the transformed cloud) + 24 B matched target
// point and normal; one partial row (28 / 1 doubles) per FS_QPB queries; one 16 B pass over the binned targets
// per half-iteration.  Cell-bound look-ups and candidate gathers beyond that are traffic, not compulsory bytes.
static double icp_alg_bytes(int numiters, int64_t n_slots, int64_t n_queries, int64_t n_binned) {
  const double rows = (double)gs_ceil_div(n_slots, FS_QPB);
  const double full = 12.0 * n_slots + 36.0 * n_queries + 8.0 * LIN_NV * rows + 16.0 * n_binned;
  const double look = 12.0 * n_slots + 24.0 * n_queries + 8.0 * rows + 16.0 * n_binned;
  return (double)numiters * (full + look);
}

static int icp_run(const float* src, int64_t n_src, const float* tgt, const float* tgt_normals,
                   int64_t n_tgt, const float* init16, const float* compose16,
                   const gs_icp_params* prm, float* out_T16, int64_t* out_idx, void* icp_scratch,
                   void* tape, void* stream, const int64_t* n_src_dev = nullptr,
                   const int64_t* n_tgt_dev = nullptr, GsTargetFilter flt = GsTargetFilter{nullptr, 1, 1}) {
  GS_REQUIRE(prm, "params_host must not be NULL");
  GS_REQUIRE(n_src > 0 && n_tgt > 0, "empty point set");
  GS_REQUIRE(n_tgt < 0x7fffffffll && n_src < 0x7fffffffll, "too many points");
  GS_REQUIRE(src && tgt && tgt_normals && init16 && out_T16 && icp_scratch, "NULL pointer");
  GS_REQUIRE(prm->numiters >= 0 && prm->numiters <= GS_ICP_MAX_ITERS, "numiters must be in [0, 1024]");
  GS_REQUIRE(prm->mode == 0 || prm->mode == 1, "mode must be 0 (ICP) or 1 (gradICP)");
  hipStream_t st = gs_stream(stream);
  IcpScratch sc = icp_carve(icp_scratch, n_src);
  hipLaunchKernelGGL(gs_icp_init_kernel, dim3(1), dim3(64), 0, st, sc.state, init16, prm->damp, prm->numiters,
                     compose16, out_T16);
  // device-side counts (n_src / n_tgt are then upper bounds) always take the grid path
  const bool dev_counts = n_src_dev || n_tgt_dev || flt.pix;  // a target filter exists only in the grid path
  const bool use_grid = prm->numiters > 0 && (dev_counts || (icp_grid_enabled() && gs_knn_use_grid(n_src, n_tgt)));
  const GsCount n_src_c{n_src, n_src_dev}, n_tgt_c{n_tgt, n_tgt_dev};
  float* bufs[2] = {sc.srcA, sc.srcB};
  TapePtrs tp = {nullptr, nullptr, nullptr};
  if (tape) {
    GsIcpTape t = gs_icp_tape_carve(tape, n_src, prm->numiters);
    tp.src = t.src; tp.idx = t.idx; tp.sys = t.sys;
  }
  // iteration it works on the cloud bufs[it & 1], or on its tape slot when a tape is recorded
  auto cloud = [&](int it) { return tp.src ? tp.src + (size_t)it * 3 * (size_t)n_src : bufs[it & 1]; };
  auto tidx = [&](int it, int which) { return tp.idx ? tp.idx + ((size_t)it * 2 + which) * (size_t)n_src : nullptr; };

  if (use_grid) {
    // the target set is fixed for all 2*numiters searches of this solve: bin it once
    int rc = gs_knn_grid_build(tgt, n_tgt_c, n_src, sc.grid, st, flt, tgt_normals);
    if (rc != GS_OK) return rc;
    GridMem gm = grid_carve(sc.grid, n_src, n_tgt);
    const int nfs = (int)icp_rows(n_src);
    const bool reduce_rows = nfs > FS_REDUCE_ROWS;
    const float* cur_in = src;  // cloud before the pending transform of the half-iteration
    int h = 0;                  // half-iteration index: kernel h reads s[h&1] / partials[(h+1)&1], writes the others
    // one event pair around the 2 x numiters half-iteration kernels; work = algorithmic bytes (icp_alg_bytes).
    // With a target filter n_tgt is the whole map: the binned count is read back (profile passes only).
    double prof_bytes = 0.0;
    if (g_gs_prof_on) {
      int64_t n_binned = n_tgt;
      if (flt.pix) {
        unsigned hits = 0;
        GS_HIP(hipMemcpyAsync(&hits, gm.bbox + 6, 4, hipMemcpyDeviceToHost, st));
        GS_HIP(hipStreamSynchronize(st));
        n_binned = hits;
      }
      prof_bytes = icp_alg_bytes(prm->numiters, n_src, n_src, n_binned);
    }
    std::unique_ptr<GsProf> prof_loop(new GsProf(GS_PROF_ICP_FUSED, prof_bytes, st, 2 * prm->numiters));
    const IcpHalfPlan plan = icp_half_plan(n_src, 1);
    IcpHalfBatch hb;
    hb.B = 1;
    hb.weak_room = 0.0f;
    hb.l[0] = IcpHalfLists{nullptr, nullptr, nullptr};
    hb.w[0] = IcpHalfWide{nullptr, nullptr};
    for (int it = 0; it < prm->numiters; ++it) {
      float* cur = cloud(it);
      hb.s[0] = IcpHalfSeq{cur_in, cur, tgt, tgt_normals, n_tgt_c, gm.g, gm.cell_start, gm.sorted, gm.sorted_n,
                           reinterpret_cast<float*>(sc.best), sc.partials[(h + 1) & 1], sc.partials[h & 1], &sc.state->s[h & 1], &sc.state->s[(h + 1) & 1],
                           sc.state->trace, out_idx, tidx(it, 0), nullptr};
      icp_half_launch<true>(plan, hb, n_src_c, prm, it, 0, st);
      ++h;
      if (reduce_rows) {
        GsProf prof(GS_PROF_SOLVE, 1.0, st);
        hipLaunchKernelGGL(gs_icp_reduce_rows_kernel, dim3(1), dim3(FS_BLOCK), 0, st, sc.partials[(h + 1) & 1], n_src_c,
                           sc.rowred);
      }
      hb.s[0] = IcpHalfSeq{cur, nullptr, tgt, tgt_normals, n_tgt_c, gm.g, gm.cell_start, gm.sorted, gm.sorted_n,
                           reinterpret_cast<float*>(sc.best), reduce_rows ? sc.rowred : sc.partials[(h + 1) & 1], sc.partials[h & 1], &sc.state->s[h & 1],
                           &sc.state->s[(h + 1) & 1], sc.state->trace, nullptr, tidx(it, 1), tp.sys};
      icp_half_launch<false>(plan, hb, n_src_c, prm, it, reduce_rows ? 1 : 0, st);
      ++h;
      cur_in = cur;
    }
    prof_loop.reset();  // closing event right behind the last half-iteration kernel
    if (prm->numiters > 0) {
      GsProf prof(GS_PROF_SOLVE, 1.0, st);
      hipLaunchKernelGGL(gs_icp_finish_kernel, dim3(1), dim3(FS_BLOCK), 0, st, sc.partials[(h + 1) & 1], n_src_c,
                         sc.state, h & 1, *prm, compose16, out_T16);
    }
    GS_LAUNCH_CHECK();
    if (tape) return icp_tape_finish(tape, sc.state, n_src, prm->numiters, st);
    return GS_OK;
  }

  // ---- brute-force path
  GS_HIP(hipMemsetAsync(sc.best, 0xff, 8 * (size_t)n_src, st));
  const int nblk = (int)gs_ceil_div(n_src, LIN_BLOCK);
  double* partials = sc.partials[0];
  const float* cur_in = src;
  for (int it = 0; it < prm->numiters; ++it) {
    float* cur = cloud(it);
    // apply the pending transform (initial transform or last T_step) while searching
    gs_knn_brute_launch(cur_in, sc.state->s[0].T_step, cur, n_src, tgt, n_tgt, sc.best, st);
    {
      GsProf prof(GS_PROF_LINEARIZE, 44.0 * (double)n_src, st);  // 8 B best + 12 B src + 24 B gather
      hipLaunchKernelGGL((gs_icp_linearize_kernel<true>), dim3(nblk), dim3(LIN_BLOCK), 0, st, cur, nullptr, n_src,
                         tgt, tgt_normals, n_tgt, sc.best, prm->dist_thresh, partials, out_idx, tidx(it, 0));
    }
    {
      GsProf prof(GS_PROF_SOLVE, 1.0, st);
      hipLaunchKernelGGL(gs_icp_solve_kernel, dim3(1), dim3(SUM_BLOCK), 0, st, partials, nblk, sc.state, it, tp.sys);
    }
    // look-ahead: one_step = Tr * cur, searched and reduced without materialising it
    gs_knn_brute_launch(cur, sc.state->s[0].Tr, nullptr, n_src, tgt, n_tgt, sc.best, st);
    {
      GsProf prof(GS_PROF_LINEARIZE, 44.0 * (double)n_src, st);
      hipLaunchKernelGGL((gs_icp_linearize_kernel<false>), dim3(nblk), dim3(LIN_BLOCK), 0, st, cur,
                         sc.state->s[0].Tr, n_src, tgt, tgt_normals, n_tgt, sc.best, prm->dist_thresh, partials,
                         nullptr, tidx(it, 1));
    }
    {
      GsProf prof(GS_PROF_SOLVE, 1.0, st);
      hipLaunchKernelGGL(gs_icp_update_kernel, dim3(1), dim3(SUM_BLOCK), 0, st, partials, nblk, sc.state, *prm, it,
                         compose16, out_T16);
    }
    cur_in = cur;
  }
  GS_LAUNCH_CHECK();
  if (tape) return icp_tape_finish(tape, sc.state, n_src, prm->numiters, st);
  return GS_OK;
}

extern "C" int gs_icp_f32(const float* src, int64_t n_src, const float* tgt, const float* tgt_normals,
                          int64_t n_tgt, const float* init16, const float* compose16,
                          const gs_icp_params* prm, float* out_T16, int64_t* out_idx, void* icp_scratch,
                          void* stream) {
  return icp_run(src, n_src, tgt, tgt_normals, n_tgt, init16, compose16, prm, out_T16, out_idx, icp_scratch, nullptr,
                 stream);
}

extern "C" int gs_icp_dc_f32(const float* src, int64_t n_src_bound, const int64_t* n_src_dev, const float* tgt,
                             const float* tgt_normals, int64_t n_tgt_bound, const int64_t* n_tgt_dev,
                             const float* init16, const float* compose16, const gs_icp_params* prm, float* out_T16,
                             int64_t* out_idx, void* icp_scratch, void* stream) {
  return icp_run(src, n_src_bound, tgt, tgt_normals, n_tgt_bound, init16, compose16, prm, out_T16, out_idx,
                 icp_scratch, nullptr, stream, n_src_dev, n_tgt_dev);
}

extern "C" int gs_icp_map_dc_f32(const float* src, int64_t n_src_bound, const int64_t* n_src_dev,
                                 const float* map_points, const float* map_normals, const int32_t* pix,
                                 int64_t n_map_bound, const int64_t* n_map_dev, int W, int ds, const float* init16,
                                 const float* compose16, const gs_icp_params* prm, float* out_T16,
                                 void* icp_scratch, void* stream) {
  GS_REQUIRE(pix && W > 0 && ds > 0, "bad target filter");
  return icp_run(src, n_src_bound, map_points, map_normals, n_map_bound, init16, compose16, prm, out_T16, nullptr,
                 icp_scratch, nullptr, stream, n_src_dev, n_map_dev, GsTargetFilter{pix, W, ds});
}


// ---------------------------------------------------------------- batched localisation -----
// ICPSLAM._localize (slam/icpslam.py:238-247) for B independent sequences in ONE chain of launches: every kernel
// below runs for all sequences at once (block b -> sequence b % B), so the dependent-launch floor of the 2 x numiters
// half-iteration kernels, which bounds a single 640x480 sequence, is paid once per B frames and the chip
// is filled by B x the blocks.  Per sequence the arithmetic is that of gs_lattice_source_f32 + gs_project_map_dc_f32
// + gs_icp_map_dc_f32, bit for bit (same device functions).
struct LocSeq {
  const float* vertex;
  const float* depth;
  const float* pose16;   // previous pose: lattice transform, projection and the composed result
  float* lattice;        // [n_lat][3] ICP source (NaN = no depth)
  GsIcpState* state;
  float* out_pose16;
  char* clear_ptr;       // grid scratch bytes that must be zero before the build
  int64_t* n_valid;      // number of lattice slots with depth (profiling / roofline accounting)
  int* far_n;            // [FS_FAR_PASSES] counters of the far-query lists of the solve, zeroed here
  int* lstat;            // [3 * GL_STAT_LAUNCHES] failure counters of the ordinary candidate lists, zeroed here
  float4* wl_cq;         // [n_lat] wide lists of hard queries: every radius cleared here (NULL: none kept)
  unsigned* sync;        // [PS_WORDS] ticket / arrival / error words of the persistent solve (gs_icp_persist.h), zeroed here
};
struct LocBatch {
  int B, W, ds, Wl;
  int64_t n_lat;
  size_t clear_bytes;
  float damp;
  int numiters;
  int count_valid;       // profile passes: count the lattice slots with depth
  LocSeq s[GS_MAX_BATCH];
};
constexpr int LP_CLEAR_ITEMS = 8;  // 16-byte stores per thread of a clearing block (32 KB per block)

// lattice source (gs_lattice_source_kernel) + solver state (gs_icp_init_kernel) + zeroing of the grid scratch
GS_DEV void loc_prep_block(const LocBatch& lb, const unsigned bid, const unsigned nb_lat) {
  const LocSeq& q = lb.s[bid % lb.B];
  const unsigned blk = bid / lb.B;
  if (blk < nb_lat) {
    const int64_t e = (int64_t)blk * 256 + threadIdx.x;
    int valid = 0;
    if (e < lb.n_lat) {
      const int64_t p = (e / lb.Wl) * lb.ds * (int64_t)lb.W + (e % lb.Wl) * lb.ds;
      float g0 = __builtin_nanf(""), g1 = g0, g2 = g0;
      if (q.depth[p] > 0.0f) {
        float T[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) T[i] = q.pose16[i];
        gs_rigid_fma(T, q.vertex[3 * p], q.vertex[3 * p + 1], q.vertex[3 * p + 2], g0, g1, g2);
        valid = 1;
      }
      q.lattice[3 * e] = g0; q.lattice[3 * e + 1] = g1; q.lattice[3 * e + 2] = g2;
      if (q.wl_cq) q.wl_cq[e] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // (no wide list of an earlier frame is ever read)
    }
    const unsigned long long m = lb.count_valid ? __ballot(valid) : 0ull;
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(reinterpret_cast<unsigned long long*>(q.n_valid), (unsigned long long)__popcll(m));
    if (blk == 0 && threadIdx.x == 0) {  // gs_icp_init_kernel with init = identity
      IcpSmall sm;
      for (int i = 0; i < 16; ++i) {
        const float id = (i % 5 == 0) ? 1.0f : 0.0f;
        sm.T_total[i] = id; sm.T_step[i] = id; sm.Tr[i] = id;
      }
      for (int i = 0; i < 8; ++i) sm.xi[i] = 0.0f;
      sm.damp = lb.damp;
      sm.err = 0.0f;
      sm.pad[0] = sm.pad[1] = 0.0f;
      q.state->s[0] = sm;
      q.state->s[1] = sm;
      if (q.far_n)
        for (int i = 0; i < FS_FAR_PASSES; ++i) q.far_n[i] = 0;
      if (q.lstat)
        for (int i = 0; i < 3 * GL_STAT_LAUNCHES; ++i) q.lstat[i] = 0;
      if (q.sync)
        for (int i = 0; i < PS_WORDS; ++i) q.sync[i] = 0u;
      if (lb.numiters == 0) icp_write_result(sm, q.pose16, q.out_pose16);
    }
    return;
  }
  float4* dst = reinterpret_cast<float4*>(q.clear_ptr);
  const size_t n16 = lb.clear_bytes / 16;
  const size_t base = (size_t)(blk - nb_lat) * 256 * LP_CLEAR_ITEMS + threadIdx.x;
#pragma unroll
  for (int u = 0; u < LP_CLEAR_ITEMS; ++u) {
    const size_t i = base + (size_t)u * 256;
    if (i < n16) dst[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  }
}
__global__ void __launch_bounds__(256) gs_loc_prep_kernel(const LocBatch lb, unsigned nb_lat) {
  loc_prep_block(lb, blockIdx.x, nb_lat);
}
// The one-call step clears the grid scratch in its frame-map launch (gs_frame_maps_batch_clear): the lattice blocks
// then have nothing to wait for and share a launch with the first pass of the grid build (projection of the map under
// the previous pose + bounding box + target list), which does not read what they write.
static_assert(GB_BLOCK == 256, "lattice and bbox blocks share a launch");
__global__ void __launch_bounds__(256) gs_loc_prep_bbox_kernel(const LocBatch lb, const GsGridBatch gb, const unsigned nb_lat,
                                                               const float u_hi, const float v_hi) {
  const unsigned n_prep = (unsigned)lb.B * nb_lat;
  if (blockIdx.x < n_prep) loc_prep_block(lb, blockIdx.x, nb_lat);
  else gridb_bbox_block(gb, blockIdx.x - n_prep, u_hi, v_hi);
}

struct IcpRowsBatch {
  int B;
  const double* in[GS_MAX_BATCH];
  double* out[GS_MAX_BATCH];
};
__global__ void __launch_bounds__(FS_BLOCK) gs_icp_reduce_rows_batch_kernel(const IcpRowsBatch rb, GsCount n_src_c) {
  __shared__ double S[32];
  __shared__ double sub[FS_BLOCK / 32][32];
  const int nrows = (int)((gs_count(n_src_c) + FS_QPB - 1) / FS_QPB);
  icp_sum_rows<FS_BLOCK>(rb.in[blockIdx.x], nrows, S, sub);
  if (threadIdx.x < LIN_NV) rb.out[blockIdx.x][threadIdx.x] = S[threadIdx.x];
}

struct IcpFinishBatch {
  int B;
  const double* partials_in[GS_MAX_BATCH];
  GsIcpState* st[GS_MAX_BATCH];
  const float* compose16[GS_MAX_BATCH];
  float* out_T16[GS_MAX_BATCH];
  const unsigned* sync[GS_MAX_BATCH];   // sync record of a persistent solve (NULL: none): a raised error word = NaN pose
};
__global__ void __launch_bounds__(FS_BLOCK) gs_icp_finish_batch_kernel(const IcpFinishBatch fb, GsCount n_src_c, int buf,
                                                                       gs_icp_params prm) {
  __shared__ double red[FS_BLOCK / GS_WAVE];
  const int b = blockIdx.x;
  const int nrows_in = (int)((gs_count(n_src_c) + FS_QPB - 1) / FS_QPB);
  const double e1 = icp_sum_col27<FS_BLOCK>(fb.partials_in[b], nrows_in, red);
  if (threadIdx.x != 0) return;
  GsIcpState* st = fb.st[b];
  IcpSmall sm = st->s[buf];
  const int it = prm.numiters - 1;
  icp_update_math((float)e1, sm, prm, it < GS_ICP_MAX_ITERS ? st->trace + 12 * it : nullptr);
  st->s[buf ^ 1] = sm;
  icp_write_result(sm, fb.compose16[b], fb.out_T16[b]);
  if (fb.sync[b] && fb.sync[b][PS_ERROR] != 0u)   // a block of the persistent solve gave up waiting (gs_icp_persist.h): fail loudly
    for (int i = 0; i < 16; ++i) fb.out_T16[b][i] = __builtin_nanf("");
}

static int64_t loc_lattice(int H, int W, int ds) { return (int64_t)((H + ds - 1) / ds) * ((W + ds - 1) / ds); }

// candidate lists of ordinary queries (gs_knn.h: gl_*), behind the ICP scratch of a sequence
struct ListMem {
  float4* lq;      // [n_lat] (position the list was built at, exactness radius; 0: no list, < 0: no list and the 2x2x2
                   // stage cannot prove this point -- cube scans)
  uint32_t* ls;    // [n_lat][GL_SLOTS] slots of `sorted` (~0: empty)
  int* stat;       // [3 * GL_STAT_LAUNCHES] per launch of the solve: lists that failed their proof, points whose list
                   // is empty (nothing fitted), points the 2x2x2 stage cannot prove (no list)
};
static size_t list_mem_bytes(int64_t n_lat) {
  return gs_align(16 * (size_t)n_lat) + gs_align(4 * GL_SLOTS * (size_t)n_lat) + gs_align(4 * 3 * GL_STAT_LAUNCHES);
}
static ListMem list_carve(void* base, int64_t n_lat) {
  char* p = reinterpret_cast<char*>(base);
  ListMem m;
  m.lq = reinterpret_cast<float4*>(p); p += gs_align(16 * (size_t)n_lat);
  m.ls = reinterpret_cast<uint32_t*>(p); p += gs_align(4 * GL_SLOTS * (size_t)n_lat);
  m.stat = reinterpret_cast<int*>(p);
  return m;
}

// candidate lists of far queries (gs_knn.h), behind the lists of the ordinary ones
struct FarMem {
  uint32_t* c;    // [n_lat][GS_FAR_SLOTS] slots of `sorted`
  float4* cq;     // [n_lat] (position the list was built at, exactness radius)
  int* idx;       // [FS_FAR_PASSES][n_lat] source points handed to the list builder
  int* n;         // [FS_FAR_PASSES] how many
};
static size_t far_mem_bytes(int64_t n_lat) {
  return gs_align(4 * GS_FAR_SLOTS * (size_t)n_lat) + gs_align(16 * (size_t)n_lat) +
         gs_align(4 * FS_FAR_PASSES * (size_t)n_lat) + 256;
}
static FarMem far_carve(void* base, int64_t n_lat) {
  char* p = reinterpret_cast<char*>(base);
  FarMem m;
  m.c = reinterpret_cast<uint32_t*>(p); p += gs_align(4 * GS_FAR_SLOTS * (size_t)n_lat);
  m.cq = reinterpret_cast<float4*>(p); p += gs_align(16 * (size_t)n_lat);
  m.idx = reinterpret_cast<int*>(p); p += gs_align(4 * FS_FAR_PASSES * (size_t)n_lat);
  m.n = reinterpret_cast<int*>(p);
  return m;
}

// Lists for the source points a first half found far from every target (IcpHalfSeq::far_idx): one launch behind the
// first halves of iterations 0, 1, 4, 8, ... (fs_far_pass; the step of iteration 0 is the large one of a solve: lists
// built before it often do not survive it; solves that keep wandering by a millimetre per iteration lose lists later
// too, and a point without a list pays a cube scan or a pass over all targets in EVERY launch until the next pass).
// The half-iteration hands over what its search found (far_cq[s] = squared distance, radius of the proving cube or -1),
// so a group of FS_HG lanes per point only collects everything within R of the point's position (the transformed cloud
// of the iteration) from that cube; points only the brute-force pass served are finished by the block (one collecting
// pass over the binned targets, FS_BQ points at a time).  The flag in the sign bit of d2prev[s] tells the following
// searches that s has a list.
struct FarBuildSeq {
  const float* src;         // transformed cloud of the iteration
  const int* idx;           // the pass's far source points, n[0] of them
  const int* n;
  const GsGrid* gp;
  const int* cell_start;
  const float4* sorted;
  float* d2prev;
  FarMem m;
};
struct FarBuildBatch {
  int B;
  FarBuildSeq s[GS_MAX_BATCH];
};
constexpr int FAR_BLOCK = 512;
constexpr int FAR_BLOCKS_PER_SEQ = 64;
__global__ void __launch_bounds__(FAR_BLOCK) gs_icp_far_build_kernel(const FarBuildBatch fb) {
  const FarBuildSeq& q = fb.s[blockIdx.x % fb.B];
  const int blk = (int)(blockIdx.x / fb.B);
  const int n = *q.n;
  if (blk >= n) return;   // (entry i is served by block i % FAR_BLOCKS_PER_SEQ)
  constexpr int NG = FAR_BLOCK / FS_HG;
  __shared__ float qs[NG][3];
  __shared__ float d1_s[NG];
  __shared__ uint8_t flag_s[NG];
  __shared__ int unres_q[NG], sq_s[NG];
  __shared__ int unres_n;
  __shared__ uint32_t stage_s[NG][GS_FAR_SLOTS + 4];
  const GsGrid g = *q.gp;
  const int grp = threadIdx.x / FS_HG, l16 = threadIdx.x & (FS_HG - 1);
  // entries of this block: blk, blk + FAR_BLOCKS_PER_SEQ, ...; NG of them per round
  for (int base = blk; base < n; base += FAR_BLOCKS_PER_SEQ * NG) {   // block-uniform
    if (threadIdx.x == 0) unres_n = 0;
    __syncthreads();
    const int e = base + grp * FAR_BLOCKS_PER_SEQ;
    if (e < n) {
      const int sq = q.idx[e];
      const float hx = q.src[3 * (int64_t)sq], hy = q.src[3 * (int64_t)sq + 1], hz = q.src[3 * (int64_t)sq + 2];
      // what the search of the half-iteration found for this point: squared distance of its neighbour, radius of the
      // cube that proved it (-1: the brute-force pass did)
      const float4 found = q.m.cq[sq];
      const float d1 = sqrtf(found.x);
      const int kdone = (int)found.y;
      if (kdone >= 0) {
        const float R = far_emit_cube<FS_HG>(g, q.cell_start, q.sorted, hx, hy, hz, l16, d1, kdone, stage_s[grp],
                                             reinterpret_cast<int*>(&stage_s[grp][GS_FAR_SLOTS]));
        for (int u = l16; u < GS_FAR_SLOTS; u += FS_HG) q.m.c[GS_FAR_SLOTS * (int64_t)sq + u] = stage_s[grp][u];
        if (l16 == 0) {
          q.m.cq[sq] = make_float4(hx, hy, hz, R);
          if (R > 0.0f) q.d2prev[sq] = __uint_as_float(__float_as_uint(q.d2prev[sq]) | 0x80000000u);
        }
      } else if (l16 == 0) {
        qs[grp][0] = hx; qs[grp][1] = hy; qs[grp][2] = hz;
        d1_s[grp] = d1;
        sq_s[grp] = sq;
        unres_q[atomicAdd(&unres_n, 1)] = grp;
      }
    }
    __syncthreads();
    const int nun = unres_n;   // block-uniform
    for (int u = 0; u < nun; u += FS_BQ) {
      const int nq = nun - u < FS_BQ ? nun - u : FS_BQ;
      block_brute_collect_multi<FAR_BLOCK, FS_BQ>(qs, unres_q + u, nq, q.sorted, q.cell_start[g.ncell], d1_s,
                                                  GS_FAR_RADD * g.c, sq_s, q.m.cq, q.m.c, flag_s);
      if ((int)threadIdx.x < nq) {
        const int gq = unres_q[u + threadIdx.x];
        const int s2 = sq_s[gq];
        if (flag_s[gq]) q.d2prev[s2] = __uint_as_float(__float_as_uint(q.d2prev[s2]) | 0x80000000u);
      }
    }
    __syncthreads();
  }
}

extern "C" int64_t gs_localize_scratch_bytes(int H, int W, int ds, int64_t n_map_bound) {
  if (H < 1 || W < 1 || ds < 1) return 0;
  const int64_t n_lat = loc_lattice(H, W, ds);
  return (int64_t)(gs_align(12 * (size_t)n_lat) + gs_align(4 * (size_t)(n_map_bound > 0 ? n_map_bound : 1)) + 256) +
         gs_icp_scratch_bytes(n_lat, n_map_bound) + (int64_t)list_mem_bytes(n_lat) +
         (int64_t)far_mem_bytes(n_lat);
}

static int64_t loc_rows(const gs_map_view& m) { return m.capacity > m.n_bound ? m.capacity : m.n_bound; }

// scratch of one sequence: lattice | pix | n_valid | ICP scratch (which holds the grid)
struct LocCarve {
  float* lattice;
  int32_t* pix;
  int64_t* n_valid;
  IcpScratch sc;
  GridMem gm;
};
static LocCarve loc_carve(const gs_localize_seq& q, int64_t n_lat) {
  LocCarve c;
  char* p = reinterpret_cast<char*>(q.scratch);
  c.lattice = reinterpret_cast<float*>(p); p += gs_align(12 * (size_t)n_lat);
  // the layout follows the map's CAPACITY (what the scratch was sized for), not the count bound of this frame: every
  // pointer below then stays the same from frame to frame (the half-iteration launches are replayed as a graph)
  const int64_t rows = loc_rows(q.map);
  c.pix = reinterpret_cast<int32_t*>(p); p += gs_align(4 * (size_t)rows);
  c.n_valid = reinterpret_cast<int64_t*>(p); p += 256;
  c.sc = icp_carve(p, n_lat);
  c.gm = grid_carve(c.sc.grid, n_lat, rows);
  return c;
}

// grid_cleared: the caller has zeroed the first gs_knn_grid_clear_bytes() bytes of every sequence's grid scratch
// (loc_carve().gm.g) on the stream already
static int localize_chunk(const gs_localize_seq* seqs, int B, int H, int W, int ds, const gs_icp_params* prm,
                          hipStream_t st, bool grid_cleared) {
  const int64_t n_lat = loc_lattice(H, W, ds);
  const int Wl = (W + ds - 1) / ds;
  LocBatch lb;
  GsGridBatch gb;
  IcpScratch sc[GS_MAX_BATCH];
  GridMem gm[GS_MAX_BATCH];
  lb.B = gb.B = B;
  lb.W = gb.W = W; gb.H = H;
  lb.ds = gb.ds = ds;
  lb.Wl = Wl; lb.n_lat = n_lat; lb.damp = prm->damp; lb.numiters = prm->numiters;
  gb.cells_cap = gs_knn_grid_cells_cap(n_lat);
  for (int b = 0; b < B; ++b) {
    const gs_localize_seq& q = seqs[b];
    const LocCarve cv = loc_carve(q, n_lat);
    float* lattice = cv.lattice;
    int32_t* pix = cv.pix;
    int64_t* n_valid = cv.n_valid;
    sc[b] = cv.sc;
    gm[b] = cv.gm;
    lb.s[b] = LocSeq{q.vertex, q.depth, q.prev_pose16, lattice, sc[b].state, q.out_pose16,
                     reinterpret_cast<char*>(gm[b].g), n_valid, nullptr, nullptr, nullptr,
                     reinterpret_cast<unsigned*>(sc[b].rowred + LIN_NV)};   // (behind the one row of large solves)
    static int binned_normals = -1;  // GRADSLAM_HIP_ICP_BINNED_NORMALS=0: gather the matches' normals from the map (A/B)
    if (binned_normals < 0) {
      const char* e = getenv("GRADSLAM_HIP_ICP_BINNED_NORMALS");
      binned_normals = (e && atoi(e) == 0) ? 0 : 1;
    }
    if (!binned_normals) gm[b].sorted_n = nullptr;
    gb.s[b] = GsGridSeq{q.map.points, GsCount{q.map.n_bound, q.map.n_dev}, pix, q.prev_pose16, q.K16,
                        binned_normals ? q.map.normals : nullptr, gm[b]};
    if (g_gs_prof_on) GS_HIP(hipMemsetAsync(n_valid, 0, 8, st));
  }
  // candidate lists of far queries (gs_knn.h; the results do not depend on them), behind the ordinary lists in the scratch.
  // Policy: round 3 built them for 1296x968 (78k source points, clusters of far ones at the frame borders: +4 % frames/s
  // over 200 frames), where they REPLACED the ordinary lists (the two do not fit one kernel).  Round 5: the wide lists of
  // hard queries (below) give those points lists inside the variants with ordinary lists, and ordinary + wide lists beat the
  // far lists at 1296x968 by 18 % (780 vs 662 frames/s over 150 frames, same poses: profiles/r05_c5_far_vs_wide.txt) --
  // so the far lists are opt-in now: GRADSLAM_HIP_ICP_FAR=1.
  static int far_lists = -1;
  if (far_lists < 0) {
    const char* e = getenv("GRADSLAM_HIP_ICP_FAR");
    far_lists = (e && atoi(e) != 0) ? 1 : 0;
  }
  const bool far_on = far_lists == 1 && prm->numiters > 0;
  FarMem fm[GS_MAX_BATCH];
  for (int b = 0; b < B; ++b) {
    fm[b] = far_carve(reinterpret_cast<char*>(sc[b].state) + gs_icp_scratch_bytes(n_lat, loc_rows(seqs[b].map)) +
                          list_mem_bytes(n_lat), n_lat);
    lb.s[b].far_n = fm[b].n;   // (zeroed by the prep launch whether or not this solve keeps lists: gs_localize_far_stats_i64)
    if (!far_on) fm[b] = FarMem{nullptr, nullptr, nullptr, nullptr};
  }
  // candidate lists of ordinary queries (gs_knn.h: gl_*; the results do not depend on them): every solve that keeps no
  // far lists.  GRADSLAM_HIP_ICP_LISTS=0 switches them off (A/B runs).
  // GRADSLAM_HIP_ICP_LISTS_FROM=k: the lists are built behind the look-ahead search of iteration k and tried from
  // iteration k + 1 on.  Default FS_LISTS_FROM = 1: the step of iteration 0 is the large one of a solve (millimetres);
  // a list built before it would not survive it.  Later starts were measured too: 0 / 1 / 4 / 8 give
  // 7.40 / 7.42 / 7.42 / 7.31 k frames/s at 8 sequences per GPU.
  static int ord_lists = -1, lists_from = FS_LISTS_FROM;
  if (ord_lists < 0) {
    const char* e = getenv("GRADSLAM_HIP_ICP_LISTS");
    ord_lists = (e && atoi(e) == 0) ? 0 : 1;
    const char* f = getenv("GRADSLAM_HIP_ICP_LISTS_FROM");
    if (f && atoi(f) >= 0) lists_from = atoi(f);
  }
  ListMem lm[GS_MAX_BATCH];
  for (int b = 0; b < B; ++b) {
    lm[b] = list_carve(reinterpret_cast<char*>(sc[b].state) + gs_icp_scratch_bytes(n_lat, loc_rows(seqs[b].map)), n_lat);
    lb.s[b].lstat = lm[b].stat;   // (zeroed by the prep launch whether or not this solve keeps lists)
  }
  // wide lists of hard queries (gs_knn.h; the results do not depend on them): in the memory of the far lists, whenever
  // those are not in use.  GRADSLAM_HIP_ICP_WIDE=0 switches them off (A/B runs).
  static int wide_lists = -1;
  if (wide_lists < 0) {
    const char* e = getenv("GRADSLAM_HIP_ICP_WIDE");
    wide_lists = (e && atoi(e) == 0) ? 0 : 1;
  }
  const bool wide_on = wide_lists == 1 && !far_on && prm->numiters > 0;
  FarMem wm[GS_MAX_BATCH];
  for (int b = 0; b < B; ++b) {
    wm[b] = far_carve(reinterpret_cast<char*>(sc[b].state) + gs_icp_scratch_bytes(n_lat, loc_rows(seqs[b].map)) +
                          list_mem_bytes(n_lat), n_lat);
    lb.s[b].wl_cq = wide_on ? wm[b].cq : nullptr;
  }
  lb.count_valid = g_gs_prof_on ? 1 : 0;
  lb.clear_bytes = gs_knn_grid_clear_bytes(gm[0], gb.cells_cap);  // same layout offsets for every sequence
  const unsigned nb_lat = (unsigned)gs_ceil_div(n_lat, 256);
  const unsigned nb_clear = (unsigned)gs_ceil_div((int64_t)(lb.clear_bytes / 16), 256 * LP_CLEAR_ITEMS);
  {
    double bytes = 0.0;  // lattice 28 B per slot; projection 16 B + two filter passes 4 B per map row; cell table
    for (int b = 0; b < B; ++b) bytes += 28.0 * n_lat + 24.0 * seqs[b].map.n_bound + 16.0 * gb.cells_cap;
    GsProf prof(GS_PROF_COMPACT, bytes, st, grid_cleared ? 4 : 5);
    int rc;
    if (grid_cleared) {
      const float u_hi = (float)((double)W - 0.999), v_hi = (float)((double)H - 0.999);   // (as gs_knn_grid_build_batch)
      hipLaunchKernelGGL(gs_loc_prep_bbox_kernel, dim3((unsigned)B * nb_lat + gs_knn_gridb_bbox_blocks(gb)), dim3(256), 0, st,
                         lb, gb, nb_lat, u_hi, v_hi);
      rc = gs_knn_grid_build_batch(gb, st, true);
    } else {
      hipLaunchKernelGGL(gs_loc_prep_kernel, dim3((unsigned)B * (nb_lat + nb_clear)), dim3(256), 0, st, lb, nb_lat);
      rc = gs_knn_grid_build_batch(gb, st);
    }
    if (rc != GS_OK) return rc;
  }
  if (prm->numiters == 0) { GS_LAUNCH_CHECK(); return GS_OK; }

  const int nfs = (int)icp_rows(n_lat);
  const bool reduce_rows = nfs > FS_REDUCE_ROWS;
  const GsCount n_src_c{n_lat, nullptr};
  double prof_bytes = 0.0;
  if (g_gs_prof_on) {  // exact counts for the roofline line (profile passes only: one sync)
    for (int b = 0; b < B; ++b) {
      unsigned hits = 0;
      int64_t nv = 0;
      GS_HIP(hipMemcpyAsync(&hits, gm[b].bbox + 6, 4, hipMemcpyDeviceToHost, st));
      GS_HIP(hipMemcpyAsync(&nv, lb.s[b].n_valid, 8, hipMemcpyDeviceToHost, st));
      GS_HIP(hipStreamSynchronize(st));
      prof_bytes += icp_alg_bytes(prm->numiters, n_lat, nv, hits);
    }
  }
  std::unique_ptr<GsProf> prof_loop(new GsProf(GS_PROF_ICP_FUSED, prof_bytes, st, 2 * prm->numiters));
  // The list-checking half-iterations as ONE persistent launch per XCD-resident sequence (gs_icp_persist.h) whenever a
  // sequence's source points fit the 32 CUs of one XCD at 672 per block (640x480 at dsratio 4: 29 blocks); the launches in
  // front of it then run at 2 lanes per point (the persistent kernel reads the 4-entry lists two lanes write).
  // OPT-IN (GRADSLAM_HIP_ICP_PERSIST=1; the results do not depend on it: tests/test_hip_batch.py::
  // test_persistent_xcd_solve_leaves_results_identical).  Measured, round 6 (profiles/r06_xcd_persistent_*): its steady-state
  // half-iteration takes 6.5 - 7.6 us against 10.5 - 12.5 us per launch, but a half-iteration in which any list of a block
  // fails costs ~27 us (re-search 9 + cubes 5 + new lists 3 on top), solves that still move lose lists in every look-ahead,
  // and the launches in front of it run at 2 lanes per point: +3.9 / +3.2 / +7.8 % at 8 / 4 / 2 sequences per GPU over the
  // 20-step window, -1.2 % over the 200-step window, -2 ... -6 % for a lone sequence (profiles/r06_xcd_persistent_ab_bench.json).
  // Its liveness also rests on the dispatcher placing at least 29 of the launch's blocks on every XCD (every wait is
  // bounded: a NaN pose, not a hang).  The launch-per-half-iteration path stays the default.
  static int persist_env = -1;
  if (persist_env < 0) {
    const char* e = getenv("GRADSLAM_HIP_ICP_PERSIST");
    persist_env = (e && atoi(e) == 1) ? 1 : 0;
  }
  // (as few blocks as the XCD's CUs allow, equally loaded: 200 row units -> 29 blocks of 7)
  const int ps_upb = (int)gs_ceil_div((int64_t)icp_rows(n_lat), PS_CUS_PER_XCD);
  const int ps_nb = (int)gs_ceil_div((int64_t)icp_rows(n_lat), ps_upb > 0 ? ps_upb : 1);
  bool persist_on = persist_env == 1 && ord_lists == 1 && !far_on && B <= (int)GS_XCDS && ps_upb <= PS_UPB &&
                    prm->numiters > lists_from + 1 && !getenv("GRADSLAM_HIP_ICP_TIMELINE") &&
                    !getenv("GRADSLAM_HIP_ICP_LANES");
  for (int b = 0; b < B && persist_on; ++b) persist_on = gm[b].sorted_n != nullptr;
  const IcpHalfPlan plan = icp_half_plan(n_lat, B, far_on ? 4 : (persist_on ? 2 : 8));
  // (a block reads the lists of ONE group of row units: solves whose blocks walk several groups keep none)
  bool lists_on = ord_lists == 1 && !far_on && plan.upb * plan.G * FS_QPB <= FS_BLOCK;
  if (plan.G != 2) persist_on = false;
  // The list variants sum the partial rows with UNCONDITIONAL loads, masked afterwards (icp_sum_col27_hook: row
  // threadIdx.x of every thread; icp_sum_rows_split: 24 x 18 rows): up to FS_BLOCK rows of LIN_NV doubles from the start
  // of a row buffer whatever the row count.  What follows the buffers in the scratch (the grid, the lists) must cover
  // that, or the solve keeps no lists (nothing else enforces the layout).
  for (int b = 0; b < B && lists_on; ++b) {
    const char* end = reinterpret_cast<const char*>(seqs[b].scratch) + gs_localize_scratch_bytes(H, W, ds, loc_rows(seqs[b].map));
    const char* first = reinterpret_cast<const char*>(reduce_rows ? sc[b].rowred : sc[b].partials[1]);
    if (reinterpret_cast<const char*>(sc[b].partials[0]) > first) first = reinterpret_cast<const char*>(sc[b].partials[0]);
    if (reinterpret_cast<const char*>(sc[b].partials[1]) > first) first = reinterpret_cast<const char*>(sc[b].partials[1]);
    if (end - first < (ptrdiff_t)(sizeof(double) * LIN_NV * FS_BLOCK)) lists_on = false;
  }
  IcpHalfBatch hb;
  hb.B = B;
  // GRADSLAM_HIP_ICP_WEAK_ROOM=<cells> (0: off): see the list-building branch of icp_half_body
  static float weak_room = -1.0f;
  if (weak_room < 0.0f) {
    const char* e = getenv("GRADSLAM_HIP_ICP_WEAK_ROOM");
    weak_room = e ? (float)atof(e) : 0.0f;
    if (!(weak_room >= 0.0f && weak_room < 1.0f)) weak_room = 0.0f;
  }
  hb.weak_room = weak_room;
  int h = 0;
  if (!lists_on) persist_on = false;
  // The persistent launch takes over behind the list-building look-ahead, which then leaves 8-entry lists (LMODE 3): the
  // persistent kernel keeps the four nearest of a list in registers and falls back on the other four, lane by lane, before
  // anything is re-searched (gs_icp_persist.h).  (Measured with 4-entry lists and a later start, GRADSLAM_HIP_ICP_PERSIST_FROM
  // = 2 / 5 / 7 / 9 / 12: 7 924 / 7 857 / 7 872 / 7 816 / 7 763 frames/s against 7 623 without -- the earliest start won.)
  int ps_it0 = prm->numiters;
  if (persist_on) {
    ps_it0 = lists_from + 1;
    if (ps_it0 >= prm->numiters) { persist_on = false; ps_it0 = prm->numiters; }
  }
  for (int it = 0; it < ps_it0; ++it) {
    for (int b = 0; b < B; ++b) {
      const gs_localize_seq& q = seqs[b];
      const float* cur_in = it == 0 ? lb.s[b].lattice : (((it - 1) & 1) ? sc[b].srcB : sc[b].srcA);
      float* cur = (it & 1) ? sc[b].srcB : sc[b].srcA;
      hb.s[b] = IcpHalfSeq{cur_in, cur, q.map.points, q.map.normals, GsCount{q.map.n_bound, q.map.n_dev}, gm[b].g,
                           gm[b].cell_start, gm[b].sorted, gm[b].sorted_n, reinterpret_cast<float*>(sc[b].best),
                           sc[b].partials[(h + 1) & 1], sc[b].partials[h & 1],
                           &sc[b].state->s[h & 1], &sc[b].state->s[(h + 1) & 1], sc[b].state->trace, nullptr, nullptr,
                           nullptr, fm[b].cq, fm[b].c, fm[b].idx, fm[b].n};
      hb.w[b] = wide_on ? IcpHalfWide{wm[b].cq, wm[b].c} : IcpHalfWide{nullptr, nullptr};
      hb.l[b] = lists_on ? IcpHalfLists{lm[b].lq, lm[b].ls, lm[b].stat} : IcpHalfLists{nullptr, nullptr, nullptr};
    }
    // lists: built behind the look-ahead search of iteration lists_from, tried from then on
    icp_half_launch<true>(plan, hb, n_src_c, prm, it, 0, st, lists_on && it > lists_from ? 2 : 0);
    if (far_on && fs_far_pass(it) >= 0) {   // lists for the far source points this search found
      const int fp = fs_far_pass(it);
      FarBuildBatch fbb;
      fbb.B = B;
      for (int b = 0; b < B; ++b)
        fbb.s[b] = FarBuildSeq{(it & 1) ? sc[b].srcB : sc[b].srcA, fm[b].idx + (int64_t)fp * n_lat, fm[b].n + fp, gm[b].g,
                               gm[b].cell_start, gm[b].sorted, reinterpret_cast<float*>(sc[b].best), fm[b]};
      hipLaunchKernelGGL(gs_icp_far_build_kernel, dim3((unsigned)B * FAR_BLOCKS_PER_SEQ), dim3(FAR_BLOCK), 0, st, fbb);
    }
    ++h;
    if (reduce_rows) {
      IcpRowsBatch rb;
      rb.B = B;
      for (int b = 0; b < B; ++b) { rb.in[b] = sc[b].partials[(h + 1) & 1]; rb.out[b] = sc[b].rowred; }
      hipLaunchKernelGGL(gs_icp_reduce_rows_batch_kernel, dim3((unsigned)B), dim3(FS_BLOCK), 0, st, rb, n_src_c);
    }
    for (int b = 0; b < B; ++b) {
      IcpHalfSeq& u = hb.s[b];
      u.src_in = (it & 1) ? sc[b].srcB : sc[b].srcA;
      u.src_out = nullptr;
      u.partials_in = reduce_rows ? sc[b].rowred : sc[b].partials[(h + 1) & 1];
      u.partials_out = sc[b].partials[h & 1];
      u.st_in = &sc[b].state->s[h & 1];
      u.st_out = &sc[b].state->s[(h + 1) & 1];
    }
    icp_half_launch<false>(plan, hb, n_src_c, prm, it, reduce_rows ? 1 : 0, st,
                           lists_on ? (it == lists_from ? (persist_on ? 3 : 1) : (it > lists_from ? 2 : 0)) : 0);
    ++h;
  }
  if (persist_on) {
    IcpPersistBatch pb;
    pb.B = B;
    pb.nb = ps_nb;
    pb.upb = ps_upb;
    pb.h0 = h;
    pb.tl_h = 0;
    pb.timeline = nullptr;
    for (int b = 0; b < B; ++b) {
      const gs_localize_seq& q = seqs[b];
      pb.s[b] = IcpPersistSeq{((ps_it0 - 1) & 1) ? sc[b].srcB : sc[b].srcA, GsCount{q.map.n_bound, q.map.n_dev}, gm[b].g, gm[b].cell_start,
                              gm[b].sorted, gm[b].sorted_n, {sc[b].partials[0], sc[b].partials[1]}, sc[b].state, lm[b].lq, lm[b].ls,
                              lm[b].stat, wide_on ? wm[b].cq : nullptr, wide_on ? wm[b].c : nullptr, lb.s[b].sync};
    }
#ifdef GS_ICP_TIMELINE
    // debugging aid (library built with -DGS_ICP_TIMELINE; GRADSLAM_HIP_ICP_PERSIST_TIMELINE=<path>): per block, the phase
    // stamps of the LAST iteration's two half-iterations (tools/icp_persist_timeline.py)
    static const char* ptl_path = getenv("GRADSLAM_HIP_ICP_PERSIST_TIMELINE");
    static unsigned long long* ptl_buf = nullptr;
    constexpr size_t PTL_WORDS = 72 * (size_t)GS_XCDS * PS_CUS_PER_XCD;
    if (ptl_path) {
      if (!ptl_buf && hipMalloc(&ptl_buf, 8 * PTL_WORDS) != hipSuccess) ptl_buf = nullptr;
      if (ptl_buf) {
        (void)hipMemsetAsync(ptl_buf, 0, 8 * PTL_WORDS, st);
        pb.timeline = ptl_buf;
        static const char* tl_it_env = getenv("GRADSLAM_HIP_ICP_TIMELINE_IT");
        pb.tl_h = 2 * (tl_it_env ? atoi(tl_it_env) : prm->numiters - 1);
      }
    }
#endif
    hipLaunchKernelGGL(gs_icp_persist_kernel, dim3(GS_XCDS * PS_CUS_PER_XCD), dim3(PS_BLOCK), 0, st, pb, n_lat, prm->dist_thresh,
                       *prm);
#ifdef GS_ICP_TIMELINE
    if (pb.timeline) {
      std::unique_ptr<unsigned long long[]> hbuf(new unsigned long long[PTL_WORDS]);
      if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(hbuf.get(), ptl_buf, 8 * PTL_WORDS, hipMemcpyDeviceToHost) == hipSuccess) {
        FILE* f = fopen(ptl_path, "w");
        if (f) {
          fprintf(f, "# B=%d nb=%d upb=%d h0=%d tl_h=%d: per block: xcc lb end | first half: wait_done sums scalar check rare arrived at_barrier research hard brute - - | look-ahead: the same | start releases... (100 MHz ticks)\n",
                  B, pb.nb, pb.upb, pb.h0, pb.tl_h);
          for (size_t i = 0; i < PTL_WORDS / 72; ++i) {
            const unsigned long long* r = hbuf.get() + 72 * i;
            if (!r[0]) continue;
            fprintf(f, "%llu %llu %llu |", r[1], r[2], r[3]);
            for (int k = 8; k < 32; ++k) fprintf(f, " %llu%s", r[k], k == 19 ? " |" : "");
            fprintf(f, " | %llu", r[4]);   // block start, then the release time of every half-iteration, then the end
            for (int k = 32; k < 72; ++k) fprintf(f, " %llu", r[k]);
            fprintf(f, "\n");
          }
          fclose(f);
        }
      }
    }
#endif
    h = 2 * prm->numiters;
  }
  prof_loop.reset();
  {
    GsProf prof(GS_PROF_SOLVE, 1.0, st);
    IcpFinishBatch fb;
    fb.B = B;
    for (int b = 0; b < B; ++b) {
      fb.partials_in[b] = sc[b].partials[(h + 1) & 1];
      fb.st[b] = sc[b].state;
      fb.compose16[b] = seqs[b].prev_pose16;
      fb.out_T16[b] = seqs[b].out_pose16;
      fb.sync[b] = persist_on ? lb.s[b].sync : nullptr;
    }
    hipLaunchKernelGGL(gs_icp_finish_batch_kernel, dim3((unsigned)B), dim3(FS_BLOCK), 0, st, fb, n_src_c, h & 1, *prm);
  }
  GS_LAUNCH_CHECK();
  return GS_OK;
}

static int localize_batch(const gs_localize_seq* seqs_host, int B, int H, int W, int ds, const gs_icp_params* prm,
                          void* stream, bool grid_cleared) {
  GS_REQUIRE(seqs_host && prm && B > 0 && H > 0 && W > 0 && ds > 0, "bad arguments");
  GS_REQUIRE(prm->numiters >= 0 && prm->numiters <= GS_ICP_MAX_ITERS, "numiters must be in [0, 1024]");
  GS_REQUIRE(prm->mode == 0 || prm->mode == 1, "mode must be 0 (ICP) or 1 (gradICP)");
  GS_REQUIRE((int64_t)H * W < (1ll << 31), "image too large for int32 pixel ids");
  for (int b = 0; b < B; ++b) {
    const gs_localize_seq& q = seqs_host[b];
    GS_REQUIRE(q.vertex && q.depth && q.K16 && q.prev_pose16 && q.out_pose16 && q.scratch, "NULL pointer");
    GS_REQUIRE(q.map.points && q.map.normals && q.map.n_bound > 0 && q.map.n_bound < 0x7fffffffll,
               "every sequence needs a non-empty map (points + normals)");
    GS_REQUIRE(q.map.capacity <= 0 || q.map.n_bound <= q.map.capacity, "map.n_bound exceeds map.capacity");
  }
  hipStream_t st = gs_stream(stream);
  for (int c0 = 0; c0 < B; c0 += GS_MAX_BATCH) {
    const int nb = B - c0 < GS_MAX_BATCH ? B - c0 : GS_MAX_BATCH;
    const int rc = localize_chunk(seqs_host + c0, nb, H, W, ds, prm, st, grid_cleared);
    if (rc != GS_OK) return rc;
  }
  return GS_OK;
}
extern "C" int gs_localize_batch_f32(const gs_localize_seq* seqs_host, int B, int H, int W, int ds,
                                     const gs_icp_params* prm, void* stream) {
  return localize_batch(seqs_host, B, H, W, ds, prm, stream, false);
}

__global__ void __launch_bounds__(256) gs_far_stats_kernel(const int* __restrict__ far_idx, const int* __restrict__ far_n,
                                                          const float* __restrict__ d2prev, const float4* __restrict__ far_cq,
                                                          int64_t n_lat, int* __restrict__ out4) {
  // (the first pass: what the first search of the solve found; clamped: a scratch no solve has used holds anything)
  const int n0 = far_n[0], n = n0 < 0 ? 0 : (n0 > (int)n_lat ? (int)n_lat : n0);
  if (blockIdx.x == 0 && threadIdx.x == 0) { out4[0] = n; out4[2] = far_n[1]; }
  int c = 0, built = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int s = far_idx[i];
    if (s < 0 || s >= (int)n_lat) continue;
    c += (__float_as_uint(d2prev[s]) >> 31) ? 1 : 0;
    built += far_cq[s].w > 0.0f ? 1 : 0;
  }
  if (c) atomicAdd(&out4[1], c);
  if (built) atomicAdd(&out4[3], built);
}
extern "C" int gs_localize_far_stats_i64(const void* scratch, int H, int W, int ds, int64_t map_rows, int64_t* out4_host,
                                         void* stream) {
  GS_REQUIRE(scratch && out4_host && H > 0 && W > 0 && ds > 0 && map_rows > 0, "bad arguments");
  const int64_t n_lat = loc_lattice(H, W, ds);
  gs_localize_seq q;
  memset(&q, 0, sizeof(q));
  q.scratch = const_cast<void*>(scratch);
  q.map.capacity = map_rows; q.map.n_bound = map_rows;
  const LocCarve cv = loc_carve(q, n_lat);
  const FarMem fm = far_carve(reinterpret_cast<char*>(cv.sc.state) + gs_icp_scratch_bytes(n_lat, map_rows) +
                                  list_mem_bytes(n_lat), n_lat);
  hipStream_t st = gs_stream(stream);
  // (the result words live behind the counters, in the 256 bytes reserved for them)
  int* out4 = fm.n + 8;
  GS_HIP(hipMemsetAsync(out4, 0, 16, st));
  hipLaunchKernelGGL(gs_far_stats_kernel, dim3(64), dim3(256), 0, st, fm.idx, fm.n, reinterpret_cast<const float*>(cv.sc.best),
                     fm.cq, n_lat, out4);
  int h[4] = {0, 0, 0, 0};
  GsGrid g;
  GS_HIP(hipMemcpyAsync(h, out4, 16, hipMemcpyDeviceToHost, st));
  GS_HIP(hipMemcpyAsync(&g, cv.gm.g, sizeof(g), hipMemcpyDeviceToHost, st));
  GS_HIP(hipStreamSynchronize(st));
  for (int i = 0; i < 4; ++i) out4_host[i] = h[i];
  if (getenv("GRADSLAM_HIP_DEBUG_GRID"))
    fprintf(stderr, "grid: box (%.3f %.3f %.3f)-(%.3f %.3f %.3f) c %.4f cells %d x %d x %d = %d\n", g.ox, g.oy, g.oz, g.mx,
            g.my, g.mz, g.c, g.nx, g.ny, g.nz, g.ncell);
  return GS_OK;
}

extern "C" int gs_localize_list_stats_i64(const void* scratch, int H, int W, int ds, int64_t map_rows, int64_t* out192_host,
                                          void* stream) {
  GS_REQUIRE(scratch && out192_host && H > 0 && W > 0 && ds > 0 && map_rows > 0, "bad arguments");
  const int64_t n_lat = loc_lattice(H, W, ds);
  gs_localize_seq q;
  memset(&q, 0, sizeof(q));
  q.scratch = const_cast<void*>(scratch);
  q.map.capacity = map_rows; q.map.n_bound = map_rows;
  const LocCarve cv = loc_carve(q, n_lat);
  const ListMem lm = list_carve(reinterpret_cast<char*>(cv.sc.state) + gs_icp_scratch_bytes(n_lat, map_rows), n_lat);
  int h[3 * GL_STAT_LAUNCHES];
  hipStream_t st = gs_stream(stream);
  GS_HIP(hipMemcpyAsync(h, lm.stat, sizeof(h), hipMemcpyDeviceToHost, st));
  GS_HIP(hipStreamSynchronize(st));
  for (int i = 0; i < 3 * GL_STAT_LAUNCHES; ++i) out192_host[i] = h[i];
  return GS_OK;
}

extern "C" int gs_pointfusion_step_batch_f32(const gs_step_seq* seqs_host, int B, int H, int W, int ds,
                                             const gs_icp_params* prm, float two_sigma_sq, float dist_th, float dot_th,
                                             int renorm_all, void* stream) {
  GS_REQUIRE(seqs_host && prm && B > 0 && H > 0 && W > 0 && ds > 0, "bad arguments");
  const int64_t P = (int64_t)H * W;
  const gs_step_seq& s0 = seqs_host[0];
  const int64_t dstride = B > 1 ? seqs_host[1].depth - s0.depth : P;
  for (int b = 0; b < B; ++b) {
    const gs_step_seq& q = seqs_host[b];
    GS_REQUIRE(q.depth && q.rgb && q.K16 && q.prev_pose16 && q.out_pose16 && q.vertex && q.normal && q.alpha &&
                   q.best_pix && q.new_count_out && q.loc_scratch && q.upd_scratch, "NULL pointer");
    GS_REQUIRE((q.gvertex != nullptr) == (q.gnormal != nullptr) && (q.gvertex != nullptr) == (s0.gvertex != nullptr),
               "gvertex / gnormal: both or none, the same choice for every sequence");
    GS_REQUIRE(q.out_pose16 != q.prev_pose16, "out_pose16 must not alias prev_pose16");
    GS_REQUIRE(q.vertex == s0.vertex + 3 * P * b && q.normal == s0.normal + 3 * P * b && q.alpha == s0.alpha + P * b,
               "vertex / normal / alpha of the batch must be dense");
    GS_REQUIRE(q.depth == s0.depth + dstride * b && q.K16 == s0.K16 + 16 * (int64_t)b, "depth / K16 must be equally strided");
  }
  GS_REQUIRE(dstride >= P || B == 1, "overlapping depth images");
  // the sequences are independent: chunk by chunk through the frame maps and the localisation; the map update is ONE
  // call for the whole batch (its merge looks at the correspondences of every sequence of the call)
  // Global maps not asked for (gvertex == NULL): they are never written -- the update computes the global vertex / normal
  // of a pixel where it uses them -- and the frame-map launch initialises the update's per-pixel tables, so that the
  // update has no per-pixel pass at all (it was 26 us of a 1.05 ms frame at 8 x 640x480).
  const bool implicit_global = s0.gvertex == nullptr;
  std::unique_ptr<gs_update_seq[]> us(new gs_update_seq[B]);
  for (int c0 = 0; c0 < B; c0 += GS_MAX_BATCH) {
    const int nb = B - c0 < GS_MAX_BATCH ? B - c0 : GS_MAX_BATCH;
    const gs_step_seq* sq = seqs_host + c0;
    gs_localize_seq ls[GS_MAX_BATCH];
    for (int b = 0; b < nb; ++b) {
      const gs_step_seq& q = sq[b];
      ls[b].vertex = q.vertex; ls[b].depth = q.depth; ls[b].K16 = q.K16; ls[b].prev_pose16 = q.prev_pose16;
      ls[b].map = q.map; ls[b].out_pose16 = q.out_pose16; ls[b].scratch = q.loc_scratch;
      gs_update_seq& u = us[c0 + b];
      u.map = q.map; u.vertex = q.vertex; u.normal = q.normal; u.depth = q.depth; u.rgb = q.rgb;
      u.alpha = q.alpha; u.pose16 = q.out_pose16; u.K16 = q.K16; u.gvertex = q.gvertex;
      u.gnormal = q.gnormal; u.best_pix = q.best_pix; u.new_count_out = q.new_count_out;
      u.scratch = q.upd_scratch;
    }
    // the frame-map launch also zeroes the grid scratch of this chunk's localisation (which then starts with ONE launch
    // for the source lattice and the map projection instead of a clearing launch followed by the projection)
    const int64_t n_lat = loc_lattice(H, W, ds);
    GsClearJob job;
    job.n = nb;
    for (int b = 0; b < nb; ++b) {
      GS_REQUIRE(ls[b].map.points && ls[b].map.n_bound > 0, "every sequence needs a non-empty map");
      const LocCarve cv = loc_carve(ls[b], n_lat);
      job.ptr[b] = reinterpret_cast<char*>(cv.gm.g);
      if (b == 0) job.bytes = gs_knn_grid_clear_bytes(cv.gm, gs_knn_grid_cells_cap(n_lat));
    }
    GsPixelTables tabs{};
    tabs.n = implicit_global ? nb : 0;
    tabs.call_flag = c0 == 0 ? gs_update_map_call_flag(seqs_host[0].upd_scratch) : nullptr;
    for (int b = 0; b < tabs.n; ++b) {
      gs_update_map_tables(sq[b].upd_scratch, &tabs.any_flag[b], &tabs.key_pix[b]);
      tabs.best_pix[b] = sq[b].best_pix;
    }
    int rc = gs_frame_maps_batch_clear(sq[0].depth, dstride, P, sq[0].K16, nb, 1, H, W, two_sigma_sq, sq[0].vertex,
                                       sq[0].normal, sq[0].alpha, &job, stream, &tabs);
    if (rc != GS_OK) return rc;
    rc = localize_batch(ls, nb, H, W, ds, prm, stream, true);
    if (rc != GS_OK) return rc;
  }
  return gs_update_map_fusion_batch_impl(us.get(), B, H, W, dist_th, dot_th, renorm_all, stream, implicit_global);
}

extern "C" int64_t gs_icp_tape_bytes(int64_t n_src, int numiters) { return (int64_t)gs_icp_tape_size(n_src, numiters); }

extern "C" int gs_icp_tape_f32(const float* src, int64_t n_src, const float* tgt, const float* tgt_normals,
                               int64_t n_tgt, const float* init16, const float* compose16,
                               const gs_icp_params* prm, float* out_T16, int64_t* out_idx, void* icp_scratch,
                               void* tape, void* stream) {
  GS_REQUIRE(tape, "tape must not be NULL");
  return icp_run(src, n_src, tgt, tgt_normals, n_tgt, init16, compose16, prm, out_T16, out_idx, icp_scratch, tape,
                 stream);
}

extern "C" int gs_icp_trace_f32(const void* icp_scratch, int numiters, float* trace_out, void* stream) {
  GS_REQUIRE(icp_scratch && trace_out && numiters >= 0 && numiters <= GS_ICP_MAX_ITERS, "bad arguments");
  const GsIcpState* st = reinterpret_cast<const GsIcpState*>(icp_scratch);
  GS_HIP(hipMemcpyAsync(trace_out, st->trace, sizeof(float) * 12 * (size_t)numiters, hipMemcpyDeviceToDevice,
                        gs_stream(stream)));
  return GS_OK;
}
