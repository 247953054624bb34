// gs_icp_persist.h — the list-checking half-iterations of a solve as ONE launch: a persistent solve per sequence,
// resident on ONE XCD (round 6).  Included by gs_icp_loop.hip behind icp_half_body, whose helpers it shares.
//
// Why.  The 36 list-checking launches of a 20-iteration solve (gs_icp_half_batch_kernel<*, *, false, 2>) are a chain of
// dependent kernels of 11 - 12.5 us each at 8 sequences per GPU, of which ~1.8 us is the kernel boundary and 3.4 - 4 us
// the cold start behind it: the per-XCD L2s are written back and invalidated at every boundary, so each launch pulls its
// working set (source points, lists, listed targets, partial rows: 17 - 33 MB for 8 sequences) from the Infinity Cache
// again (profiles/r05_a_icp_l2_vs_frame.txt).  A sequence's blocks already live on one XCD (block b serves sequence
// b % 8), and inside one XCD the L2 is the point of coherence: the 40 global reductions of a solve need neither a kernel
// boundary nor an L2 write-back, only an L1 bypass on the reading side.
//
// How.  ONE block of 1024 threads per CU (16 waves: 128 VGPRs per lane, the whole LDS), up to 7 row units = 672 source
// points per block, one LANE per source point (29 blocks for a 640x480 lattice, all on the 32 CUs of the sequence's
// XCD).  A block keeps ITS source
// points, their candidate lists and the listed targets / normals in REGISTERS across the half-iterations (the cloud of
// an iteration is T_step applied to the registers; a list is re-read only after something rewrote it), the solver state
// in LDS.  Per half-iteration: partial rows by plain stores (the lines stay in this XCD's L2) -> s_waitcnt vmcnt(0)
// -> one L2 atomic on the sequence's arrival counter -> every block polls the counter with L1-bypassing loads, sums
// the rows with L1-bypassing loads (fixed order: the sums of icp_sum_rows), runs the scalar stage redundantly, checks
// its lists.  No gather, no search and no global load of per-point data on the path of a half-iteration whose lists
// prove; what does not prove goes through exactly the passes of icp_half_body (re-search, cubes, wide lists, block
// pass), so the arithmetic -- and with it every bit of the result -- is that of the launch-per-half-iteration path.
//
// Placement is ESTABLISHED, not assumed: a block reads the XCD it runs on (HW_REG_XCC_ID) and takes a ticket from that
// XCD's sequence; blocks beyond the sequence's block count leave at once.  Correctness therefore never depends on the
// dispatcher (only on the L2 being shared by the CUs of one XCD); what depends on it is liveness -- an XCD that
// receives fewer than `nb` resident blocks would leave its sequence waiting -- so every wait is bounded: on timeout
// the block raises the sequence's error word, the finish launch turns it into a NaN pose, and nothing hangs.
// (Observed dispatch: block b -> XCD b % 8, all 256 blocks of the launch resident at once.)
// First attempt of the round, for the record: 2 blocks of 768 threads per CU at 2 lanes per point (the geometry of the
// launch-per-half kernels) -- bit-identical at once, and 2.5x SLOWER than the launches: 21 registers kept across the
// halves on top of an 80-VGPR budget meant ~300 scratch reloads per half-iteration in the dependent chain.
#pragma once

constexpr int PS_BLOCK = 1024;                 // threads per block, ONE block per CU
constexpr int PS_UPB = 7;                      // row units per block (at most)
constexpr int PS_NQ = PS_UPB * FS_QPB;         // 672 source points per block, one lane each
constexpr int PS_LM = 4;                       // list entries per source point (what 2 lanes x 2 write in the launches before)
constexpr int PS_CUS_PER_XCD = 32;             // the grid is 8 x this
constexpr unsigned PS_SPIN_LIMIT = 1u << 22;   // polls before a block gives up (~ seconds)
static_assert(PS_NQ <= PS_BLOCK - GS_WAVE && 2 * PS_UPB <= PS_BLOCK / GS_WAVE && PS_LM <= GL_SLOTS && PS_LM == 2 * gl_k<2>(), "block shape");

// words of a sequence's sync record (zeroed by the prep launch of the frame)
enum { PS_TICKET = 0, PS_ARRIVED = 1, PS_ERROR = 2, PS_WORDS = 8 };

struct IcpPersistSeq {
  const float* src_in;       // the cloud behind the first half of iteration it0 - 1 (read once)
  GsCount n_tgt;
  const GsGrid* gp;
  const int* cell_start;
  const float4* sorted;
  const float4* sorted_n;
  double* partials[2];       // half h reads [(h + 1) & 1], writes [h & 1]
  GsIcpState* state;         // s[h0 & 1]: state behind launch h0 - 1; the last half leaves s[(2 numiters) & 1]
  float4* lq;
  uint32_t* ls;
  int* lstat;
  float4* wl_cq;
  uint32_t* wl_c;
  unsigned* sync;            // [PS_WORDS]
};
struct IcpPersistBatch {
  int B;
  int nb;                    // blocks per sequence
  int upb;                   // row units per block
  int h0;                    // first half-iteration served (even: a first half)
  int tl_h;                  // (timeline builds) the first of the two half-iterations whose phases are stamped
  unsigned long long* timeline;
  IcpPersistSeq s[GS_MAX_BATCH];
};

// L1-bypassing accesses (served by the XCD's L2) and L2 atomics, spelled in ISA so that no compiler reasoning about
// scopes applies to them
GS_DEV unsigned ps_load_u32_sc1(const unsigned* p) {
  unsigned v;
  asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
GS_DEV unsigned ps_atomic_inc_ret(unsigned* p) {
  unsigned v, one = 1u;
  asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p), "v"(one) : "memory");
  return v;
}
GS_DEV void ps_atomic_inc(unsigned* p) {
  unsigned one = 1u;
  asm volatile("global_atomic_add %0, %1, off" : : "v"(p), "v"(one) : "memory");
}
GS_DEV double ps_load_f64_sc1(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
GS_DEV float4 ps_load_f4_sc1(const float4* p) {
  typedef float v4 __attribute__((ext_vector_type(4)));
  v4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return make_float4(v.x, v.y, v.z, v.w);
}
GS_DEV uint4 ps_load_u4_sc1(const uint32_t* p) {
  typedef unsigned v4 __attribute__((ext_vector_type(4)));
  v4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return make_uint4(v.x, v.y, v.z, v.w);
}

// icp_sum_rows<FS_BLOCK, 18> / icp_sum_col27<FS_BLOCK> by the first FS_BLOCK threads of the block with L1-bypassing loads:
// the same additions in the same order (every thread of the block calls them)
GS_DEV void ps_sum_rows(const double* __restrict__ partials, int nrows, double* S, double (*sub)[32]) {
  constexpr int CH = 18, STEP = FS_BLOCK / 32;
  const int i = threadIdx.x & 31, j = threadIdx.x >> 5;
  if (threadIdx.x < FS_BLOCK) {
    double s = 0.0;
    if (i < LIN_NV) {
      for (int b = j * CH; b < nrows; b += CH * STEP) {
        const double* base = partials + (int64_t)b * LIN_NV + i;
        const int left = nrows - b;
        double a[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) a[u] = (u < left) ? ps_load_f64_sc1(base + u * LIN_NV) : 0.0;
#pragma unroll
        for (int u = 0; u < CH; ++u) s += a[u];
      }
    }
    sub[j][i] = s;
  }
  gs_barrier_lds();
  if (threadIdx.x < LIN_NV) {
    double t = 0.0;
    for (int k = 0; k < STEP; ++k) t += sub[k][threadIdx.x];
    S[threadIdx.x] = t;
  }
  gs_barrier_lds();
}
GS_DEV double ps_sum_col27(const double* __restrict__ partials, int nrows, double* red) {
  double s = 0.0;
  if (threadIdx.x < FS_BLOCK)
    for (int b = threadIdx.x; b < nrows; b += FS_BLOCK) s += ps_load_f64_sc1(partials + (int64_t)b * LIN_NV + 27);
  s = gs_wave_sum_f64(s);
  if ((threadIdx.x & (GS_WAVE - 1)) == 0) red[threadIdx.x / GS_WAVE] = s;
  gs_barrier_lds();
  double t = 0.0;
  for (int w = 0; w < FS_BLOCK / GS_WAVE; ++w) t += red[w];
  gs_barrier_lds();
  return t;
}

// The two scalar stages OUT OF LINE.  Inlined into the loop over the half-iterations their float64 code hands the compiler
// ~100 loop-invariant constants (polynomial coefficients: 64-bit literals have to sit in register pairs) to hoist out of
// the loop, and everything that lives across the loop goes to scratch (measured: 194 scalar + 201 vector spills, ~300
// scratch reloads per half-iteration).  A call keeps the constants where they are used.  The state and the sums live
// in LDS: the pointers carry the address space, so the callee addresses LDS directly (no flat accesses).
typedef __attribute__((address_space(3))) IcpSmall PsLdsSmall;
typedef __attribute__((address_space(3))) double PsLdsF64;
__device__ __noinline__ void ps_stage_update(const float e1, PsLdsSmall* smp, const gs_icp_params prm, float* trace_row) {
  IcpSmall& sm = *(IcpSmall*)smp;
  icp_update_math_wave(e1, sm, prm, trace_row, (int)(threadIdx.x & (GS_WAVE - 1)));
}
__device__ __noinline__ void ps_stage_solve(PsLdsF64* Sp, PsLdsSmall* smp) {
  const double* S = (const double*)Sp;
  IcpSmall& sm = *(IcpSmall*)smp;
  gs_solve_spd6_wave(S, sm.damp, sm.xi);
  icp_solve_finish_wave(S, sm, (int)(threadIdx.x & (GS_WAVE - 1)));
}

struct IcpPersistShared {
  alignas(16) IcpSmall sm;
  double S[32];
  double sub[FS_BLOCK / 32][32];
  unsigned long long keys_s[PS_NQ];
  int bslot_s[PS_NQ];
  float qs[PS_NQ][3];
  float qa_s[PS_NQ][8];
  double sub_s[PS_UPB][FS_RG][LIN_NV];
  int unres_q[PS_NQ], hard_q[PS_NQ], fail_q[PS_NQ];
  int unres_n, hard_n, fail_n;
  unsigned long long red[PS_BLOCK / GS_WAVE];
  int lfail_s[3];
  int ctl[2];                  // [0] ticket, [1] abort
  IcpPersistSeq q;             // the sequence's record, for the out-of-line passes
  GsGrid g;                    // the grid header
  // the candidate lists of this block's source points: (position the list was made at, exactness radius) and the slots of
  // `sorted` -- the working copy; nothing is written back (lists are per solve).  The listed points themselves sit in the
  // owner lanes' registers.
  float4 lq[PS_NQ];
  uint32_t ls[PS_NQ][GL_SLOTS];
  uint8_t relist_s[PS_NQ];     // the list of this slot's point was rewritten by the out-of-line passes: its lane re-reads it
  int lstat_blk[3][GL_STAT_LAUNCHES];   // failure counters of this block per half-iteration (diagnostics)
};

// the registers a lane keeps across the half-iterations: its source point and the points of its candidate list (the list's
// header and slots are in LDS)
struct PsRegs {
  float p0, p1, p2;            // the cloud of the current iteration
  float4 cv[PS_LM];            // the listed points
  float4 cn0;                  // the normal of the first of them (the nearest when the list was made: almost always the match)
};

GS_DEV void ps_regs_empty(PsRegs& r) {
  r.p0 = r.p1 = r.p2 = __builtin_nanf("");
#pragma unroll
  for (int j = 0; j < PS_LM; ++j) r.cv[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  r.cn0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}
// the listed points of slot `slot` (its header and slots are in LDS already); a list that cannot be tried -- no source
// point, no radius, a slot beyond the binned targets -- is emptied
GS_DEV void ps_list_gather(const IcpPersistSeq& q, IcpPersistShared& L, const int slot, const bool live, const uint32_t nsl, PsRegs& r) {
  uint4 sv = make_uint4(~0u, ~0u, ~0u, ~0u);
  if (live) {
    const bool try_list = r.p0 == r.p0 && L.lq[slot].w > 0.0f;
    sv = *reinterpret_cast<const uint4*>(&L.ls[slot][0]);
    uint32_t sl[PS_LM] = {sv.x, sv.y, sv.z, sv.w};
    bool changed = false;
#pragma unroll
    for (int j = 0; j < PS_LM; ++j) {
      if (!try_list || !(sl[j] < nsl)) { changed = changed || sl[j] != ~0u; sl[j] = ~0u; }
    }
    sv = make_uint4(sl[0], sl[1], sl[2], sl[3]);
    if (changed) *reinterpret_cast<uint4*>(&L.ls[slot][0]) = sv;
  }
  r.cv[0] = q.sorted[sv.x != ~0u ? sv.x : 0u];
  r.cv[1] = q.sorted[sv.y != ~0u ? sv.y : 0u];
  r.cv[2] = q.sorted[sv.z != ~0u ? sv.z : 0u];
  r.cv[3] = q.sorted[sv.w != ~0u ? sv.w : 0u];
  r.cn0 = q.sorted_n[sv.x != ~0u ? sv.x : 0u];
}
static_assert(PS_LM == 4, "four listed points per lane");

#ifdef GS_ICP_TIMELINE
#define PS_STAMP(k) do { if (tl && threadIdx.x == 0) tl[(k)] = wall_clock64(); } while (0)
#else
#define PS_STAMP(k) do { } while (0)
#endif

// ---- two tiers (round 6).  A point's list in memory has EIGHT entries and the radius R8 within which it holds every target
// of the position q0 it was made at (global memory: q.lq / q.ls, written by the list-building launch, LMODE 3, and by the
// passes below).  The lane keeps the FOUR nearest in registers -- tier 1: (centre, R4) and its slots in LDS -- and checks
// them first; when they give no proof it fetches the eight itself (no pass of the block, no barrier: two round trips),
// proves on them, and re-centres its tier 1 on where the point is now: the four nearest of the eight, within
// R4' = min(0.9999 R8 - |q - q0|, distance of the fifth) -- every target that close to q is within R8 of q0, hence one of
// the eight, hence one of the four.  tools/icp_list_sim.py: in a solve whose steps grow back to millimetres (seed 0, frame
// 8) 4-entry lists lose 1 281 proofs, 8-entry lists 2; with 4-entry lists every one of those is a re-search pass of its
// whole block (~20 us: profiles/r06_xcd_persistent_timeline_b8.txt).
GS_DEV void ps_rank8(const unsigned long long (&k)[8], int (&rank)[8]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    int c = 0;
#pragma unroll
    for (int f = 0; f < 8; ++f) c += (f != e && k[f] < k[e]) ? 1 : 0;
    rank[e] = c;   // (keys carry the target index: unique unless ~0, which is never selected)
  }
}
// slots of rank 0 .. 3 and the squared distance of rank 4 (+inf: fewer than five candidates)
GS_DEV void ps_top4_of8(const unsigned long long (&k)[8], const uint32_t (&s8)[8], uint32_t (&t)[4], float& d5) {
  int rank[8];
  ps_rank8(k, rank);
  d5 = __builtin_inff();
#pragma unroll
  for (int j = 0; j < 4; ++j) t[j] = ~0u;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    if (k[e] != ~0ull) {
#pragma unroll
      for (int j = 0; j < 4; ++j) t[j] = rank[e] == j ? s8[e] : t[j];
      if (rank[e] == 4) d5 = __uint_as_float((uint32_t)(k[e] >> 32));
    }
  }
}
// keys of the eight listed targets seen from (x, y, z): two rounds of four gathers (the points are not kept)
GS_DEV void ps_keys8(const float4* __restrict__ sorted, const uint32_t (&s8)[8], const uint32_t nsl, float x, float y, float z,
                     unsigned long long (&k)[8]) {
#pragma unroll
  for (int h4 = 0; h4 < 2; ++h4) {
    float4 pt[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) pt[u] = sorted[s8[4 * h4 + u] < nsl ? s8[4 * h4 + u] : 0u];
#pragma unroll
    for (int u = 0; u < 4; ++u) k[4 * h4 + u] = s8[4 * h4 + u] < nsl ? grid_key(x, y, z, pt[u]) : ~0ull;
  }
}

// ---- points whose list gave no proof: re-searched IN LINE by 16-lane groups (64 points per round of groups), which leave
// the new list.  A solve that still moves by tenths of a millimetre per iteration loses a few lists per block in every
// look-ahead (each block holds 672 points: one is enough), so this pass is on the path of most half-iterations of such a
// solve and is built for latency: the 2x2x2 block of cells around the point as ONE flat candidate list, four gathers in
// flight per lane (the ~65 candidates of a mature map: two round trips), every lane remembering its two nearest
// candidates and the distance of its third; then the 4 nearest of the 32 remembered ones by RANK -- every lane reads the 32
// keys of its group from LDS and counts those below its own two (one LDS round, no dependent chain of cross-lane
// reductions: the group minimum and the M rounds of gl_select_write are ~50 dependent LDS-crossbar shuffles).  The list =
// the candidates of rank 0 .. 3, R = min(distance of rank 4, nearest candidate a lane dropped, bound of the block): the
// definition of gl_select_write, hence the same exactness argument.
struct PsGroupArea {
  unsigned long long key[32];   // the two nearest candidates of every lane (~0: none)
  uint32_t slot[32];
  float drop[16];               // squared distance of the nearest candidate a lane did NOT remember (+inf: none)
  float d5;                     // squared distance of the candidate of rank 4 (+inf: fewer than five remembered)
  float pad;
};
static_assert(sizeof(PsGroupArea) * (PS_BLOCK / 16) <= sizeof(double) * PS_UPB * FS_RG * LIN_NV, "the group areas live in sub_s");
GS_DEV void ps_wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
typedef __attribute__((address_space(3))) IcpPersistShared PsLdsShared;
__device__ __noinline__ void ps_research(PsLdsShared* Lp, const int u_first) {
  IcpPersistShared& L = *(IcpPersistShared*)Lp;
  const IcpPersistSeq& q = L.q;
  const int nf = L.fail_n;   // block-uniform
  const int* __restrict__ cell_start = q.cell_start;
  const float4* __restrict__ sorted = q.sorted;
  const GsGrid g = L.g;
  constexpr int FG = 16, NF = 4;
  PsGroupArea* areas = reinterpret_cast<PsGroupArea*>(&L.sub_s[0][0][0]);
  const int l = threadIdx.x & (FG - 1), grp = threadIdx.x / FG;
  PsGroupArea& A = areas[grp];
  for (int i = grp; i < nf; i += PS_BLOCK / FG) {
    const int hs = L.fail_q[i];
    const float qx = L.qs[hs][0], qy = L.qs[hs][1], qz = L.qs[hs][2];
    // the 2x2x2 block of cells whose centre is nearest to the (projected) query, as grid_search_stage0_top
    const GsQueryCell qc = grid_query_cell(g, qx, qy, qz);
    const float fx = (qc.px - g.ox) * g.inv_c - (float)qc.cx, fy = (qc.py - g.oy) * g.inv_c - (float)qc.cy,
                fz = (qc.pz - g.oz) * g.inv_c - (float)qc.cz;
    const int cx = qc.cx, cy = qc.cy, cz = qc.cz;
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
    unsigned long long k0 = ~0ull, k1 = ~0ull;   // this lane's two nearest candidates (k0 <= k1)
    uint32_t s0 = ~0u, s1 = ~0u;
    float drop = __builtin_inff();
    for (int t0 = l; t0 < total; t0 += NF * FG) {
      float4 pt[NF];
      int ix[NF];
#pragma unroll
      for (int u = 0; u < NF; ++u) {
        const int t = t0 + u * FG, tt = t < total ? t : 0;
        ix[u] = tt < e1 ? sb0 + tt : (tt < e2 ? sb1 + (tt - e1) : (tt < e3 ? sb2 + (tt - e2) : sb3 + (tt - e3)));
        pt[u] = sorted[ix[u]];
      }
#pragma unroll
      for (int u = 0; u < NF; ++u) {
        const unsigned long long k2 = t0 + u * FG < total ? grid_key(qx, qy, qz, pt[u]) : ~0ull;
        // (keys order by distance bits, then index; a NaN distance gives ~0 and is never remembered)
        const bool b0 = k2 < k0, b1 = k2 < k1;
        const unsigned long long out = b1 ? k1 : k2;   // the candidate that is not (or no longer) remembered
        const float od = __uint_as_float((uint32_t)(out >> 32));
        drop = (out != ~0ull && od < drop) ? od : drop;
        k1 = b0 ? k0 : (b1 ? k2 : k1);
        s1 = b0 ? s0 : (b1 ? (uint32_t)ix[u] : s1);
        k0 = b0 ? k2 : k0;
        s0 = b0 ? (uint32_t)ix[u] : s0;
      }
    }
    A.key[2 * l] = k0; A.key[2 * l + 1] = k1;
    A.slot[2 * l] = s0; A.slot[2 * l + 1] = s1;
    A.drop[l] = drop;
    const int64_t sq = (int64_t)u_first * FS_QPB + hs;   // the point's list in memory
    uint32_t* __restrict__ ls8 = q.ls + GL_SLOTS * sq;
    if (l == 0) { A.d5 = __builtin_inff(); A.pad = __builtin_inff(); }
    if (l < 4) L.ls[hs][l] = ~0u;
    if (l < GL_SLOTS) ls8[l] = ~0u;
    ps_wave_lds_sync();
    int r0 = 0, r1 = 0;   // ranks of this lane's two candidates among the 32 of the group (keys are unique: they carry the index)
#pragma unroll 8
    for (int e = 0; e < 32; ++e) {   // (eight keys in flight: all 32 at once cost the function its caller-saved registers)
      const unsigned long long ke = A.key[e];
      r0 += ke < k0 ? 1 : 0;
      r1 += ke < k1 ? 1 : 0;
    }
    // (the global slots were cleared by this wave a few instructions ago: memory operations of one wave to one address stay
    // in order)
    if (k0 != ~0ull) {
      if (r0 < PS_LM) L.ls[hs][r0] = s0;
      if (r0 < GL_SLOTS) ls8[r0] = s0;
      if (r0 == PS_LM) A.d5 = __uint_as_float((uint32_t)(k0 >> 32));
      if (r0 == GL_SLOTS) A.pad = __uint_as_float((uint32_t)(k0 >> 32));   // (d9)
      if (r0 == 0) { L.bslot_s[hs] = (int)s0; L.keys_s[hs] = k0; }
    }
    if (k1 != ~0ull) {
      if (r1 < PS_LM) L.ls[hs][r1] = s1;
      if (r1 < GL_SLOTS) ls8[r1] = s1;
      if (r1 == PS_LM) A.d5 = __uint_as_float((uint32_t)(k1 >> 32));
      if (r1 == GL_SLOTS) A.pad = __uint_as_float((uint32_t)(k1 >> 32));
    }
    ps_wave_lds_sync();
    if (l == 0) {
      float out2 = A.pad;   // the nearest candidate that is not on the 8-entry list: rank 8, or one a lane dropped
#pragma unroll 8
      for (int e = 0; e < FG; ++e) out2 = A.drop[e] < out2 ? A.drop[e] : out2;
      // proof of the nearest candidate and the bound of the scanned block (grid_search_stage0_top)
      unsigned long long kb = ~0ull;
#pragma unroll 8
      for (int e = 0; e < 32; e += 2) kb = A.key[e] < kb ? A.key[e] : kb;
      const float rb = (amin < 1.0e30f) ? (amin - 0.001f) * g.c : BIG;
      const float bd = __uint_as_float((uint32_t)(kb >> 32));   // NaN while nothing was found
      const bool fdone = rb > 0.0f && (rb >= 1.0e30f ? bd == bd : bd <= rb * rb);
      const float rcov2 = rb >= 1.0e30f ? __builtin_inff() : (rb > 0.0f ? rb * rb : 0.0f);
      const float R8sq = out2 < rcov2 ? out2 : rcov2;
      const float R4sq = A.d5 < R8sq ? A.d5 : R8sq;
      if (kb == ~0ull) { L.keys_s[hs] = ~0ull; L.bslot_s[hs] = -1; }
      // (no proof inside the block: the cubes of the out-of-line passes serve the point and may still give it a list)
      L.lq[hs] = make_float4(qx, qy, qz, fdone ? sqrtf(R4sq) : -1.0f);
      q.lq[sq] = make_float4(qx, qy, qz, fdone ? sqrtf(R8sq) : -1.0f);
      L.relist_s[hs] = 1;
      // (measured and dropped: re-making lists whose radius leaves the neighbour less than a quarter cell of room by the cube
      // scans -- their bound is a whole cell, 29 mm, against the 15 - 18 mm of the 2x2x2 block -- made early frames 10 %
      // faster and late frames 10 % slower: 7 655 against 7 862 frames/s over the 20-step window)
      if (!fdone) L.hard_q[atomicAdd(&L.hard_n, 1)] = hs;
    }
    ps_wave_lds_sync();   // (the group's area is reused by its next point)
  }
}

// The passes for points the 2x2x2 stage cannot prove, OUT OF LINE (wide list, cubes by 16-lane groups, block pass: those of
// icp_half_body<*, 2, false, 2>).  Only blocks with such a point call it (every thread of the block does).  They leave keys_s
// / bslot_s of the points they serve, the new lists in LDS and relist_s = 1; the owner lanes pick that up behind the call.
// Inline, these passes put the whole kernel over its register file.
__device__ __noinline__ void ps_hard_passes(PsLdsShared* Lp, const int u_first, unsigned long long* tl) {
  IcpPersistShared& L = *(IcpPersistShared*)Lp;
  const IcpPersistSeq& q = L.q;
  const int* __restrict__ cell_start = q.cell_start;
  const float4* __restrict__ sorted = q.sorted;
  const GsGrid g = L.g;
  const int nh = L.hard_n;   // block-uniform
  float4* __restrict__ wl_cq = q.wl_cq;
  uint32_t* __restrict__ wl_c = q.wl_c;
  const bool wl_on = wl_cq != nullptr;
  for (int i = threadIdx.x / FS_HG; i < nh; i += PS_BLOCK / FS_HG) {
    const int hs = L.hard_q[i], l16 = threadIdx.x & (FS_HG - 1);
    const float hx = L.qs[hs][0], hy = L.qs[hs][1], hz = L.qs[hs][2];
    const int64_t sq = (int64_t)u_first * FS_QPB + hs;
    bool done = false, listed = false;
    int win = -1;
    unsigned long long key = L.keys_s[hs];
    if (wl_on) {
      const float4 c0R = wl_cq[sq];
      if (c0R.w > 0.0f) {   // (group-uniform)
        const unsigned long long kl = wide_list_search<FS_HG>(c0R, wl_c + GS_FAR_SLOTS * sq, sorted, hx, hy, hz, l16, &done, &win);
        if (done) { key = kl; listed = true; }
        else win = -1;
      }
    }
    constexpr int KH = 2;
    GlTop<KH> top;
    float rc2 = 0.0f;
    if (!done) key = grid_search_rings_top<FS_HG, KH>(g, cell_start, sorted, hx, hy, hz, l16, key, &done, &win, FS_HARD_RINGS, top, &rc2);
    if (!listed) {   // (a point its wide list served keeps what it has: no ordinary list, R < 0)
      if (done) {
        if (wl_on) far_write_from_top<FS_HG, KH>(top, rc2, hx, hy, hz, l16, wl_c + GS_FAR_SLOTS * sq, wl_cq + sq);
        gl_select_write<FS_HG, KH>(top, rc2, hx, hy, hz, l16, GL_SLOTS, q.ls + GL_SLOTS * sq, q.lq + sq);
        if (l16 < PS_LM) L.ls[hs][l16] = ~0u;   // (tier 1 empty: the next check falls back on the eight and re-centres it)
        if (l16 == 0) L.lq[hs] = make_float4(hx, hy, hz, 0.0f);
      } else if (l16 == 0) {
        L.lq[hs] = make_float4(hx, hy, hz, -1.0f);
        q.lq[sq] = make_float4(hx, hy, hz, -1.0f);
      }
    }
    if (win >= 0) L.bslot_s[hs] = win;
    if (l16 == 0) {
      L.keys_s[hs] = key;
      L.relist_s[hs] = 1;
      if (!done) L.unres_q[atomicAdd(&L.unres_n, 1)] = hs;
    }
  }
  __syncthreads();
  PS_STAMP(8);
  const int nun = L.unres_n;   // block-uniform
  constexpr int BQL = 2;
  if (wl_on) {
    for (int u = 0; u < nun; u += BQL)
      block_brute_min_list_multi<PS_BLOCK, BQL>(L.qs, L.unres_q + u, nun - u < BQL ? nun - u : BQL, sorted, cell_start[g.ncell],
                                                L.keys_s, L.bslot_s, (int64_t)u_first * FS_QPB, wl_cq, wl_c);
  } else {
    for (int u = 0; u < nun; u += FS_BQ)
      block_brute_min_sorted_multi<PS_BLOCK, FS_BQ>(L.qs, L.unres_q + u, nun - u < FS_BQ ? nun - u : FS_BQ, sorted,
                                                    cell_start[g.ncell], L.keys_s, L.bslot_s);
  }
  if (nun) __syncthreads();
  PS_STAMP(9);
}

// One half-iteration of one block.
template <bool FULL>
GS_DEV void ps_half(const IcpPersistSeq& q, IcpPersistShared& L, PsRegs& r, const int64_t n_src, const float dist_thresh,
                    const gs_icp_params& prm, const int it, const int h, const int lb, const int u_first, const int u_last,
                    const int nunits, const uint32_t nsl, unsigned long long* tl) {
  const float4* __restrict__ sorted = q.sorted;
  const float4* __restrict__ sorted_n = q.sorted_n;
  const double* __restrict__ partials_in = q.partials[(h + 1) & 1];
  double* __restrict__ partials_out = q.partials[h & 1];
  const int slot = threadIdx.x;
  const int64_t s = (int64_t)u_first * FS_QPB + slot;
  const bool live = slot < PS_NQ && (u_first + slot / FS_QPB < u_last) && s < n_src;

  // ---- prologue: finish the previous half-iteration (identical in every block of the sequence)
  if (FULL) {
    const double e1 = ps_sum_col27(partials_in, nunits, reinterpret_cast<double*>(L.red));
    PS_STAMP(1);
    if (threadIdx.x >= PS_BLOCK - GS_WAVE) {   // the last wave: it holds no source points
      ps_stage_update((float)e1, (PsLdsSmall*)&L.sm, prm, (lb == 0 && it - 1 < GS_ICP_MAX_ITERS) ? q.state->trace + 12 * (it - 1) : nullptr);
      ps_regs_empty(r);   // (nothing of this wave's list registers lives across the call: no saves around it)
    }
  } else {
    ps_sum_rows(partials_in, nunits, L.S, L.sub);
    PS_STAMP(1);
    if (threadIdx.x >= PS_BLOCK - GS_WAVE) {
      ps_stage_solve((PsLdsF64*)L.S, (PsLdsSmall*)&L.sm);
      ps_regs_empty(r);
    }
  }
  if (threadIdx.x == 0) {
    L.unres_n = 0; L.hard_n = 0; L.fail_n = 0;
    L.lfail_s[0] = L.lfail_s[1] = L.lfail_s[2] = 0;
  }
  gs_barrier_lds();
  PS_STAMP(2);

  // ---- the list check, on registers; a lane whose list proves builds its Gauss-Newton row on the spot
  if (live) {
    float a[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}, res = 0.0f;
    unsigned long long key = ~0ull;
    int win = -1;
    float qx = r.p0, qy = r.p0, qz = r.p0;
    if (r.p0 == r.p0) {   // (an empty lattice slot stays NaN, is never searched and contributes no row)
      const float* T = FULL ? L.sm.T_step : L.sm.Tr;
      gs_rigid_fma(T, r.p0, r.p1, r.p2, qx, qy, qz);
      if (FULL) { r.p0 = qx; r.p1 = qy; r.p2 = qz; }   // the cloud of this iteration
      const float4 lqv = L.lq[slot];
      const uint4 sv = *reinterpret_cast<const uint4*>(&L.ls[slot][0]);
      const uint32_t sl[PS_LM] = {sv.x, sv.y, sv.z, sv.w};
      // (the winner's point and slot are tracked by value: an index into the register arrays would put them in scratch)
      float4 wp = r.cv[0];
      uint32_t ws = sl[0];
      bool first = true;
#pragma unroll
      for (int j = 0; j < PS_LM; ++j) {
        const unsigned long long k2 = sl[j] != ~0u ? grid_key(qx, qy, qz, r.cv[j]) : ~0ull;
        if (k2 < key) { key = k2; wp = r.cv[j]; ws = sl[j]; first = j == 0; }
      }
      const float bd = __uint_as_float((uint32_t)(key >> 32));   // NaN: empty list
      const float ex = qx - lqv.x, ey = qy - lqv.y, ez = qz - lqv.z;
      const float delta = sqrtf(ex * ex + ey * ey + ez * ez);
      const bool done = lqv.w > 0.0f && sqrtf(bd) + delta < lqv.w * 0.9999f;   // false for NaN
      if (done) {
        win = (int)ws;
        const float4 wn = first ? r.cn0 : sorted_n[ws];   // (the binned normal of the match: the same bits either way)
        gn_row_pn(qx, qy, qz, wp, wn, a, res);
        const bool keep = (dist_thresh < 0.0f) || (bd < dist_thresh);
        if (!keep) {
#pragma unroll
          for (int i = 0; i < 6; ++i) a[i] = 0.0f;
          res = 0.0f;
        }
      } else {
        bool served = false;
        if (lqv.w >= 0.0f) {   // tier 2: the eight entries of the list in memory, by this lane on its own
          const float4 lq2 = q.lq[s];
          if (lq2.w > 0.0f) {
            const uint4 sa = *reinterpret_cast<const uint4*>(q.ls + GL_SLOTS * s), sb = *reinterpret_cast<const uint4*>(q.ls + GL_SLOTS * s + 4);
            const uint32_t s8[8] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w};
            unsigned long long k8[8];
            ps_keys8(sorted, s8, nsl, qx, qy, qz, k8);
            unsigned long long kb = ~0ull;
#pragma unroll
            for (int e = 0; e < 8; ++e) kb = k8[e] < kb ? k8[e] : kb;
            const float bd8 = __uint_as_float((uint32_t)(kb >> 32));   // NaN: nothing listed
            const float fx = qx - lq2.x, fy = qy - lq2.y, fz = qz - lq2.z;
            const float delta0 = sqrtf(fx * fx + fy * fy + fz * fz);
            if (sqrtf(bd8) + delta0 < lq2.w * 0.9999f) {   // exact on the eight
              uint32_t t[4];
              float d5;
              ps_top4_of8(k8, s8, t, d5);
              // tier 1 re-centred on q: the four nearest of the eight and the radius within which they hold every target
              r.cv[0] = sorted[t[0]];   // (t[0]: the match -- it exists, bd8 is a number)
              r.cn0 = sorted_n[t[0]];
#pragma unroll
              for (int j = 1; j < PS_LM; ++j) r.cv[j] = sorted[t[j] != ~0u ? t[j] : 0u];
              float R4 = lq2.w * 0.9999f - delta0;
              const float r5 = sqrtf(d5);
              R4 = r5 < R4 ? r5 : R4;
              L.lq[slot] = make_float4(qx, qy, qz, R4 > 0.0f ? R4 : 0.0f);
              *reinterpret_cast<uint4*>(&L.ls[slot][0]) = make_uint4(t[0], t[1], t[2], t[3]);
              key = kb;
              win = (int)t[0];
              gn_row_pn(qx, qy, qz, r.cv[0], r.cn0, a, res);
              const bool keep = (dist_thresh < 0.0f) || (bd8 < dist_thresh);
              if (!keep) {
#pragma unroll
                for (int i = 0; i < 6; ++i) a[i] = 0.0f;
                res = 0.0f;
              }
              served = true;
            }
          }
        }
        if (!served) {
          key = ~0ull;
          atomicAdd(&L.lfail_s[lqv.w < 0.0f ? 2 : (lqv.w == 0.0f ? 1 : 0)], 1);
          // (R < 0: the 2x2x2 stage could not prove this point when it was last tried -- straight to the cube scans)
          if (lqv.w >= 0.0f) L.fail_q[atomicAdd(&L.fail_n, 1)] = slot;
          else L.hard_q[atomicAdd(&L.hard_n, 1)] = slot;
        }
      }
    }
    L.bslot_s[slot] = win;
    L.qs[slot][0] = qx; L.qs[slot][1] = qy; L.qs[slot][2] = qz;
    L.keys_s[slot] = key;
#pragma unroll
    for (int i = 0; i < 6; ++i) L.qa_s[slot][i] = a[i];
    L.qa_s[slot][6] = res;
  } else if (slot < PS_NQ) {
#pragma unroll
    for (int i = 0; i < 7; ++i) L.qa_s[slot][i] = 0.0f;
  }
  gs_barrier_lds();
  PS_STAMP(3);
  // ---- the rare passes: only blocks with a list that gave no proof (or a point without one) enter.  Behind them the lanes of
  // the points they served pick up the new list (its points: one round of gathers) and build the point's Gauss-Newton row
  // from it: the new list's nearest entry is the match the search found.
  if (L.fail_n | L.hard_n) {   // block-uniform
    if (L.fail_n) {
      ps_research((PsLdsShared*)&L, u_first);
      gs_barrier_lds();
    }
    PS_STAMP(7);
    // (failure counters of this half-iteration, diagnostics: kept in LDS and added to the sequence's counters when the block
    // ends -- a global atomic here sits in wave 0's memory queue in front of everything it does next)
    if (threadIdx.x < 3 && h < GL_STAT_LAUNCHES) L.lstat_blk[threadIdx.x][h] = L.lfail_s[threadIdx.x];
    if (L.hard_n) ps_hard_passes((PsLdsShared*)&L, u_first, tl);   // (block-uniform; it ends with a barrier)
    // (every lane re-reads the points of its list, rewritten or not: one round of gathers that hit the L2, and the list
    // registers are dead across the out-of-line passes -- nothing to save around the calls)
    ps_list_gather(q, L, slot < PS_NQ ? slot : 0, live, nsl, r);
    if (live && L.relist_s[slot]) {
      L.relist_s[slot] = 0;
      const unsigned long long bb = L.keys_s[slot];
      const int bsl = L.bslot_s[slot];
      float a[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}, res = 0.0f;
      // (nothing found at all -- every distance NaN -- leaves a zero row: icp_half_body takes target 0 there, which only
      // an empty target set can reach)
      if (bsl >= 0 && bb != ~0ull) {
        const float d2 = __uint_as_float((uint32_t)(bb >> 32));
        const bool keep = (dist_thresh < 0.0f) || (d2 < dist_thresh);
        if (keep) {
          const bool head = L.ls[slot][0] == (uint32_t)bsl;   // (the list's nearest entry: in the registers already)
          const float4 mp = head ? r.cv[0] : sorted[bsl];
          const float4 mn = head ? r.cn0 : sorted_n[bsl];
          gn_row_pn(L.qs[slot][0], L.qs[slot][1], L.qs[slot][2], mp, mn, a, res);
        }
      }
#pragma unroll
      for (int k = 0; k < 6; ++k) L.qa_s[slot][k] = a[k];
      L.qa_s[slot][6] = res;
    }
    gs_barrier_lds();
  }
  PS_STAMP(4);
  if (!FULL) {   // residual only: per unit the wave-sum tree over its 96 values (64 + 32), then the two wave sums
    double* red2 = reinterpret_cast<double*>(L.red);
    const int wave = threadIdx.x / GS_WAVE, wl = threadIdx.x & (GS_WAVE - 1);
    if (wave < 2 * PS_UPB) {
      const int rr_i = (wave & 1) * GS_WAVE + wl;
      double rr = 0.0;
      if (rr_i < FS_QPB) {
        const float res = L.qa_s[(wave >> 1) * FS_QPB + rr_i][6];
        rr = (double)res * (double)res;
      }
      const double sum = gs_wave_sum_f64(rr);
      if (wl == 0) red2[wave] = sum;
    }
    gs_barrier_lds();
    if (threadIdx.x < PS_UPB && u_first + (int)threadIdx.x < u_last)
      partials_out[(int64_t)(u_first + threadIdx.x) * LIN_NV + 27] = red2[2 * threadIdx.x] + red2[2 * threadIdx.x + 1];
  } else {
    for (int w = threadIdx.x; w < PS_UPB * FS_RG * LIN_NV; w += PS_BLOCK) {
      const int i = w % LIN_NV, part = (w / LIN_NV) % FS_RG, un = w / (LIN_NV * FS_RG);
      const int ia = fs_pa(i), ib = fs_pb(i);
      const float* r0 = L.qa_s[un * FS_QPB + FS_RPG * part];
      double t = (double)r0[ia] * (double)r0[ib];
#pragma unroll
      for (int u = 1; u < FS_RPG; ++u) t += (double)r0[8 * u + ia] * (double)r0[8 * u + ib];
      L.sub_s[un][part][i] = t;
    }
    gs_barrier_lds();
    if (threadIdx.x < PS_UPB * LIN_NV) {
      const int i = threadIdx.x % LIN_NV, un = threadIdx.x / LIN_NV;
      if (u_first + un < u_last) {
        double t = L.sub_s[un][0][i];
#pragma unroll
        for (int k = 1; k < FS_RG; ++k) t += L.sub_s[un][k][i];
        partials_out[(int64_t)(u_first + un) * LIN_NV + i] = t;
      }
    }
  }
  // ---- arrive: the rows of this block are in the L2 before the counter says so
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  gs_barrier_lds();
  if (threadIdx.x == 0) ps_atomic_inc(q.sync + PS_ARRIVED);
  PS_STAMP(5);
}

__global__ void __launch_bounds__(PS_BLOCK) gs_icp_persist_kernel(const IcpPersistBatch pb, const int64_t n_src, const float dist_thresh,
                                                                  const gs_icp_params prm) {
  __shared__ IcpPersistShared L;
#ifdef GS_ICP_TIMELINE
  const unsigned long long t_start = wall_clock64();
#endif
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u;   // HW_REG_XCC_ID[3:0]
  if ((int)xcc >= pb.B) return;              // sequence b lives on XCD b
  const IcpPersistSeq& q = pb.s[xcc];
  if (threadIdx.x == 0) {
    L.ctl[0] = (int)ps_atomic_inc_ret(q.sync + PS_TICKET);
    L.ctl[1] = 0;
    L.q = q;
    L.g = *q.gp;
  }
  __syncthreads();
  const int lb = L.ctl[0];
  if (lb >= pb.nb) return;                   // the sequence has its blocks
  const int nunits = (int)((n_src + FS_QPB - 1) / FS_QPB);
  const int u_first = lb * pb.upb, u_last = (u_first + pb.upb < nunits) ? u_first + pb.upb : nunits;
  const int slot = threadIdx.x;
  const int64_t s = (int64_t)u_first * FS_QPB + slot;
  const bool live = slot < PS_NQ && (u_first + slot / FS_QPB < u_last) && s < n_src;
  const uint32_t nsl = (uint32_t)(q.n_tgt.host < 0x7fffffffll ? q.n_tgt.host : 0x7fffffffll);
  const int h_end = 2 * prm.numiters;
  // the state behind launch h0 - 1, this block's source points and their lists (written by the launches before)
  if (threadIdx.x < (int)(sizeof(IcpSmall) / 4))
    reinterpret_cast<float*>(&L.sm)[threadIdx.x] = reinterpret_cast<const float*>(&q.state->s[pb.h0 & 1])[threadIdx.x];
  PsRegs r;
  r.p0 = r.p1 = r.p2 = __builtin_nanf("");
  if (slot < PS_NQ) {
    float4 lqv = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    uint4 sv = make_uint4(~0u, ~0u, ~0u, ~0u);
    if (live) {
      r.p0 = q.src_in[3 * s]; r.p1 = q.src_in[3 * s + 1]; r.p2 = q.src_in[3 * s + 2];
      // tier 1 from the 8-entry list the launch before left (centre q0, radius R8): the four nearest to q0, within
      // min(distance of the fifth, R8) of it
      const float4 lq2 = q.lq[s];
      lqv = make_float4(lq2.x, lq2.y, lq2.z, lq2.w > 0.0f ? 0.0f : lq2.w);
      if (r.p0 == r.p0 && lq2.w > 0.0f) {
        const uint4 sa = *reinterpret_cast<const uint4*>(q.ls + GL_SLOTS * s), sb = *reinterpret_cast<const uint4*>(q.ls + GL_SLOTS * s + 4);
        const uint32_t s8[8] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w};
        unsigned long long k8[8];
        ps_keys8(q.sorted, s8, nsl, lq2.x, lq2.y, lq2.z, k8);
        uint32_t t[4];
        float d5;
        ps_top4_of8(k8, s8, t, d5);
        const float r5 = sqrtf(d5);
        lqv.w = r5 < lq2.w ? r5 : lq2.w;
        sv = make_uint4(t[0], t[1], t[2], t[3]);
      }
    }
    L.lq[slot] = lqv;
    *reinterpret_cast<uint4*>(&L.ls[slot][0]) = sv;
    *reinterpret_cast<uint4*>(&L.ls[slot][4]) = make_uint4(~0u, ~0u, ~0u, ~0u);
    L.relist_s[slot] = 0;
  }
  if (threadIdx.x < 3 * GL_STAT_LAUNCHES) (&L.lstat_blk[0][0])[threadIdx.x] = 0;
  ps_list_gather(q, L, slot < PS_NQ ? slot : 0, live, nsl, r);
  __syncthreads();
  unsigned target = 0;
  for (int h = pb.h0; h < h_end; ++h) {
    unsigned long long* tl = nullptr;
#ifdef GS_ICP_TIMELINE
    if (pb.timeline && (h == pb.tl_h || h == pb.tl_h + 1)) tl = pb.timeline + 72 * (size_t)(xcc * PS_CUS_PER_XCD + lb) + 12 * (h - pb.tl_h) + 8;
    if (tl && threadIdx.x == 0) tl[6] = wall_clock64();   // at the barrier
#endif
    if (h > pb.h0) {   // the rows of half h - 1: every block of the sequence has arrived
      target += (unsigned)pb.nb;
      if (threadIdx.x == 0) {
        unsigned n = 0;
        while (ps_load_u32_sc1(q.sync + PS_ARRIVED) < target) {
          if (++n > PS_SPIN_LIMIT || ((n & 1023u) == 0u && ps_load_u32_sc1(q.sync + PS_ERROR) != 0u)) {
            L.ctl[1] = 1;
            atomicExch(q.sync + PS_ERROR, 1u);
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      gs_barrier_lds();
      if (L.ctl[1]) return;
    }
    PS_STAMP(0);
#ifdef GS_ICP_TIMELINE
    if (pb.timeline && threadIdx.x == 0 && h - pb.h0 < 39)   // release time of every half-iteration
      pb.timeline[72 * (size_t)(xcc * PS_CUS_PER_XCD + lb) + 32 + (h - pb.h0)] = wall_clock64();
#endif
    if ((h & 1) == 0) ps_half<true>(q, L, r, n_src, dist_thresh, prm, h >> 1, h, lb, u_first, u_last, nunits, nsl, tl);
    else ps_half<false>(q, L, r, n_src, dist_thresh, prm, h >> 1, h, lb, u_first, u_last, nunits, nsl, tl);
  }
  if (q.lstat && threadIdx.x < 3 * GL_STAT_LAUNCHES) {
    const int c = (&L.lstat_blk[0][0])[threadIdx.x];
    if (c) atomicAdd(q.lstat + threadIdx.x, c);
  }
  // the state behind the last look-ahead, for the finish launch
  if (lb == 0 && threadIdx.x < (int)(sizeof(IcpSmall) / 4))
    reinterpret_cast<float*>(&q.state->s[h_end & 1])[threadIdx.x] = reinterpret_cast<const float*>(&L.sm)[threadIdx.x];
#ifdef GS_ICP_TIMELINE
  if (pb.timeline && threadIdx.x == 0) {
    unsigned long long* t0 = pb.timeline + 72 * (size_t)(xcc * PS_CUS_PER_XCD + lb);
    t0[0] = 1; t0[1] = xcc; t0[2] = (unsigned long long)lb; t0[3] = wall_clock64(); t0[4] = t_start;
    if (h_end - pb.h0 < 39) t0[32 + (h_end - pb.h0)] = t0[3];
  }
#endif
}
