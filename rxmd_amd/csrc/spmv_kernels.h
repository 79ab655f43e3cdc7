// spmv_kernels.h -- the two forms of the QEq matrix pass (get_hsh / get_gradient, reference src/qeq.F90:271-363) and what they share with the
// vector kernels of the CG: k_spmv (one wavefront per row, 16-byte gather per entry) and k_spmv_win (the window pass: 16 cell-sorted rows per
// workgroup, partners staged in LDS, 16-bit slots).  Included by qeq.hip (the product) and by experiments.hip (timing probes and variants that
// launch the same kernels; compiled only by `make experiments`).
#pragma once
#include "engine.h"

#include <type_traits>

namespace rxmd {

static inline int nblk(long long n, int b) { return n > 0 ? static_cast<int>((n + b - 1) / b) : 1; }   // an empty rank still launches (kernels guard their range)

enum { S_MU = 0, S_LMIN_S, S_LMIN_T, S_GOLD_S, S_GOLD_T, S_GNEW_S, S_GNEW_T, S_EST, S_GH_S, S_GH_T, S_HSH_S, S_HSH_T, S_SSUM, S_TSUM, S_BETA_S, S_BETA_T, S_RAW0, S_RAW1, S_RAW2, S_RAW3, S_RAW4, S_RAW5, S_RAW6, S_RAW7, S_STOP, S_STOP1, S_TOL, S_COUNT };
enum { MODE_HSH = 0, MODE_GRAD = 1 };
constexpr int S_SNAP = 128;      // scal[S_SNAP + 32 p ..]: snapshot of the scalars of an iteration of parity p (scalar_algebra stage 6)
typedef double f64x2 __attribute__((ext_vector_type(2)));
#ifndef SPMV_UNR
#define SPMV_UNR 4
#endif
constexpr int UNR = SPMV_UNR;
__device__ inline double wave_sum(double v) { return wave_sum64(v); }   // DPP reduction, engine.h

template <int NC>
__device__ inline void block_store_partials(double (&acc)[NC], double *partials, int ncomp_stride) {
  // acc holds lane-0-of-wave partials; combine the block's waves in wave order, then one store per component
  __shared__ double sm[16][NC];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  if (lane == 0)
    for (int c = 0; c < NC; ++c) sm[w][c] = acc[c];
  __syncthreads();
  if (threadIdx.x < NC) {
    double s = 0.0;
    for (int k = 0; k < nw; ++k) s += sm[k][threadIdx.x];
    // device-scope store: written through to the coherence point, so the in-kernel tail (block_finish) needs no L2 write-back
    __hip_atomic_store(partials + static_cast<size_t>(blockIdx.x) * ncomp_stride + threadIdx.x, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// PQEq (pqrow != nullptr): the second pair of sums (gs_,gt_) is over the shell-core matrix hsc instead of the ghost columns;
// gradient gets the field term fpqeq (pqeq.F90:466), Est the core/shell terms of pqeq.F90:381-411 without the resident doubling
__device__ inline double pq_est_row(const DevAtomP &ap, double Zi, const double4 &pr, double qi, double hq, double bq) {
  return ap.chi * qi + 0.5 * ap.eta * qi * qi + 0.5 * (qi + Zi) * (hq + pr.y) + pr.w - (bq + pr.z);
}
// STORE (qeq_mode 1): additionally keep the raw row sums (all columns / ghost columns) so that the next gradient and Est
// follow from  H.(q + l h) = H.q + l H.h  with N-sized vector kernels instead of a second matrix pass.
// One wavefront = one row (the launch covers the rows exactly), 16 rows per workgroup: workgroup dispatch is not free at a million
// wavefronts per pass (measured 1.25 / 1.24 / 1.15-1.19 / 1.13 / 1.08-1.11 ms at 1 / 2 / 4 / 8 / 16 wavefronts per workgroup; splitting
// a row over 2 or 4 wavefronts instead: 1.22 / 2.19 ms).  The pass is bound by
// LATENCY x occupancy, not by instruction issue: a wavefront lives for a chain of dependent memory round trips (row length ->
// the two streams -> the gathers -> the operands of the tail), so (a) the tail operands are requested first, next to the row
// length, (b) four batches of 64 entries are in flight instead of eight, which brings the kernel from 99 to <= 80 VGPRs and from
// 4 to 6 wavefronts per SIMD (measured on one box: 1.276 ms -> 1.21 with (a), 1.157 with (b), 1.06-1.12 with both), (c) PQEq is
// a template parameter so that the plain kernel does not carry its code.
// PIPE (default; RXMD_SPMV_NO_PIPE=1 switches it off): the first batch of the two streams is requested BEFORE the row length is known (it lies
// inside the row's S10-entry slot whatever the length; entries behind the row's end get weight 0 once the length has arrived), one
// dependent round trip less per wavefront.  Measured 0.98 -> 0.93 ms per pass on one box and no difference on another (0.941 / 0.949 /
// 0.938): the pass is insensitive to its latency chain.  It is insensitive to the BYTES of its streams too: a 16-bit column stream
// (stencil column + offset in the column's run, 10 instead of 12 bytes per entry) gave 0.968 vs 0.988 ms without the early request and
// 0.931 vs 0.921 ms with it; requesting every batch ahead of the previous one's gathers, a tighter row stride (448 or 512 instead of
// 640 entries) and skipping the ghost-column sums on the three rows in four that have no ghost partner (7 % SLOWER: the flag is one
// more scalar round trip in front of the loop) changed nothing or lost.  All were dropped again; see NOTES.md 3.
template <int MODE, bool STORE, bool PQ, int PIPE>
__global__ void __launch_bounds__(1024) k_spmv(int N, int S10, DevFF ff, const int *__restrict__ nb10, const double *__restrict__ hess, const int *__restrict__ n10,
                                               const double2 *__restrict__ xv, const double2 *__restrict__ hst, double2 *__restrict__ gst,
                                               const double2 *__restrict__ qst, const double *__restrict__ q, const int *__restrict__ type,
                                               const double *__restrict__ scal, double *__restrict__ partials,
                                               double2 *__restrict__ rs_all, double2 *__restrict__ rs_gh,
                                               const double *__restrict__ hsc, const double4 *__restrict__ pqrow, int swz,
                                               const int *__restrict__ rowlist, int nrows, int pbase, const double *__restrict__ stopflag) {
  if (stopflag && *stopflag != 0.0) return;        // run-ahead CG loop: the iteration this launch belongs to was decided not to happen (scalar_algebra stage 6)
  // rowlist != nullptr: this launch covers nrows rows named by the list (interior or boundary rows of a multi-rank domain);
  // its workgroups write their partial sums behind the pbase workgroups of the other launch
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  // the row is the same for the 64 lanes: say so (readfirstlane), and the row's base addresses, its length and the loop bounds live in
  // scalar registers -- 44 instead of 57 VGPRs and scalar address arithmetic: 1.03-1.09 -> 0.95 ms per pass on the same box
  const int widx = (swz ? xcd_swizzle(blockIdx.x, gridDim.x) : static_cast<int>(blockIdx.x)) * wpb + __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int row = rowlist ? (widx < nrows ? rowlist[widx] : N) : widx;
  // the row tails of a workgroup are run by the first lanes of its wavefront 0 after the barrier the partial sums need anyway -- the
  // operands of consecutive rows (type, hst / qst, gst / q) and their results (row sums, gradient) are then a handful of coalesced requests
  // per workgroup instead of five per row (k_spmv_bisect: tail operands, row stores and partials are 6-8 % of the pass)
  __shared__ double s_row[16][4];
  const int wave_in_wg = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  if (row < N) {
    const size_t base = static_cast<size_t>(row) * S10;
    unsigned e[UNR];
    double h[UNR], c[UNR];
    // one batch of the row's streams (entry, hessian value, PQEq: shell-core value) for entries [kb, kb + 256) below `bound`
    auto request = [&](int kb, int bound, unsigned (&ee)[UNR], double (&hh)[UNR], double (&cc)[UNR]) {
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int k = kb + lane + 64 * u;
        const bool ok = k < bound;
        ee[u] = ok ? static_cast<unsigned>(__builtin_nontemporal_load(nb10 + base + k)) : 0u;   // streamed once: keep it out of
        hh[u] = ok ? __builtin_nontemporal_load(hess + base + k) : 0.0;                        // the caches that hold the vector
        if (PQ && (MODE == MODE_GRAD || STORE)) cc[u] = ok ? __builtin_nontemporal_load(hsc + base + k) : 0.0;
      }
    };
    if (PIPE) request(0, S10, e, h, c);
    const int n = n10[row] & N10_COUNT;
    double as = 0.0, at = 0.0, gs_ = 0.0, gt_ = 0.0;
    if (PIPE) {                                    // entries behind the row's end get weight 0
#pragma unroll
      for (int u = 0; u < UNR; ++u) { const bool ok = lane + 64 * u < n; e[u] = ok ? e[u] : 0u; h[u] = ok ? h[u] : 0.0; if (PQ) c[u] = ok ? c[u] : 0.0; }
    }
    for (int kb = 0; kb < n; kb += 64 * UNR) {   // wave-uniform trip count
      if (PIPE == 0 || (PIPE == 1 && kb > 0)) request(kb, n, e, h, c);
#pragma unroll
      for (int u = 0; u < UNR; ++u) {            // one 16-byte gather per entry from the cell-sorted vector copy
        const double2 v = xv[e[u] & NB10_IDX_MASK];
        as += h[u] * v.x;
        at += h[u] * v.y;
        if ((MODE == MODE_GRAD || STORE) && !PQ) { const double hg = (e[u] & NB10_GHOST) ? h[u] : 0.0; gs_ += hg * v.x; gt_ += hg * v.y; }   // select the weight, not the sums
        if ((MODE == MODE_GRAD || STORE) && PQ) { gs_ += c[u] * v.x; gt_ += c[u] * v.y; }      // PQEq: second matrix (shell-core) over the same columns
      }
    }
    as = wave_sum(as); at = wave_sum(at);
    if (MODE == MODE_GRAD || STORE) { gs_ = wave_sum(gs_); gt_ = wave_sum(gt_); }
    if (lane == 0) { s_row[wave_in_wg][0] = as; s_row[wave_in_wg][1] = at; s_row[wave_in_wg][2] = gs_; s_row[wave_in_wg][3] = gt_; }
  }
  __syncthreads();
  if (wave_in_wg != 0) return;
  {
    const int r_idx = (swz ? xcd_swizzle(blockIdx.x, gridDim.x) : static_cast<int>(blockIdx.x)) * wpb + lane;     // lane r = the row of wavefront r
    const int r = (lane < wpb) ? (rowlist ? (r_idx < nrows ? rowlist[r_idx] : N) : r_idx) : N;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (r < N) {
      const double as = s_row[lane][0], at = s_row[lane][1], gs_ = s_row[lane][2], gt_ = s_row[lane][3];
      const int t = type[r];
      const double2 pa = (MODE == MODE_HSH) ? hst[r] : qst[r];
      const DevAtomP ap = ff.atom[t];
      if (STORE) { rs_all[r] = make_double2(as, at); rs_gh[r] = make_double2(gs_, gt_); }
      if (MODE == MODE_HSH) {
        const double2 pb = gst[r];
        const double ts = ap.eta * pa.x + as, tt = ap.eta * pa.y + at;          // qeq.F90:294-302
        a0 = ts * pa.x; a1 = tt * pa.y;                                         // hshs_sum, hsht_sum (:309-310)
        a2 = pb.x * pa.x; a3 = pb.y * pa.y;                                     // g.h (:119,123)
      } else {
        const double mu = scal[S_MU];
        const double fpq = PQ ? pqrow[r].x : 0.0;
        const double g1 = -ap.chi - ap.eta * pa.x - as - fpq;                   // qeq.F90:349-350 (pqeq.F90:466)
        const double g2 = -1.0 - ap.eta * pa.y - at;
        gst[r] = make_double2(g1, g2);
        a0 = g1 * g1; a1 = g2 * g2;                                             // Gnew (:355-356)
        const double qi = q[r];
        const double hq_all = as - mu * at, hq_res = (as - gs_) - mu * (at - gt_);
        if (PQ) a2 = pq_est_row(ap, ff.Zpq[t], pqrow[r], qi, hq_all, gs_ - mu * gt_);
        else a2 = ap.chi * qi + 0.5 * ap.eta * qi * qi + 0.5 * qi * (hq_all + hq_res);  // Est (:297-306)
      }
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2); a3 = wave_sum(a3);           // fixed order over the workgroup's rows
    if (lane < 4) {
      const double v = lane == 0 ? a0 : (lane == 1 ? a1 : (lane == 2 ? a2 : a3));
      __hip_atomic_store(partials + (static_cast<size_t>(pbase) + blockIdx.x) * 4 + lane, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ---- window pass (k_spmv_win) -------------------------------------------------------------------------------------------------------------
// The 16 rows of a group (consecutive residents in cell-sorted order, engine.h WIN_*) couple to nearly the same partners: ~1,500 cell-sorted
// positions for ~430 entries per row.  The workgroup copies the vector entries of that window into LDS with coalesced loads (8 consecutive
// positions = one 128-byte line per unit) and every row reads its partners from there: a 16-bit slot per entry (bit 15: ghost column) replaces
// the 4-byte entry and the 16-byte gather per entry through the vector memory path disappears.  Streams: value 8 + slot 2 bytes per entry, two
// entries per lane and request (16-byte / 4-byte loads), 256 entries of a row in flight.  Same sums in the same per-row roles as k_spmv; the
// order in which a row's products are added differs (lane = entry pair), i.e. the last bits of a row sum do.
// (A timing probe with synthetic slots promised 0.76 against 0.93 ms of k_spmv before anything real was built; the real pass: 0.80 against 0.89 ms
// back to back in one process, NOTES.md 3.  Variants are compared compiled side by side through VAR and debug tap 104.)
// FORM (bit set): WIN_PREFETCH = the second batch of a row is requested before the workgroup's barrier; WIN_LEAN = groups without a ghost partner skip the
// ghost-column sums; WIN_RANKROWS = experiments build only.  Plain QEq runs WIN_PREFETCH | WIN_LEAN (RXMD_SPMV_ONE_TRIP=1 and rows of 257-384 entries: NSTEP 3, WIN_LEAN), PQEq NSTEP 1 with WIN_PREFETCH (RXMD_PQ_PREFETCH=0: NSTEP 2, no prefetch).
[[maybe_unused]] constexpr int WIN_RANKROWS = 1;
constexpr int WIN_PREFETCH = 2, WIN_LEAN = 4;
template <int MODE, bool STORE, bool PQ, int NSTEP = 2, int VAR = 0>      // VAR = FORM bits; VAR & 2: the second batch of a row is requested before the workgroup's barrier (below); VAR & 4: groups without a ghost partner skip the ghost-column sums (below); the default of plain QEq is 6.  NSTEP x 128 entries of a row in flight per register set (2; 3 = every row of at most 384 entries is one trip, the choice for water in rounds 4-5 -- 1.478 against 1.529 ms then, 1.61 against 1.58 since round 6, no longer the default; 1 for PQEq, whose third stream doubles a set); VAR: variants under measurement, compiled side by side and timed by debug tap 104
__global__ void __launch_bounds__(64 * WIN_ROWS, 8) k_spmv_win(int N, int G, int S10, DevFF ff, const unsigned short *__restrict__ sl10, const double *__restrict__ hess, const int *__restrict__ n10,
                                                            const int *__restrict__ rows_sorted, const int *__restrict__ win_k, const int *__restrict__ win_cnt,
                                                            const double2 *__restrict__ xv, const double2 *__restrict__ hst, double2 *__restrict__ gst,
                                                            const double2 *__restrict__ qst, const double *__restrict__ q, const int *__restrict__ type,
                                                            const double *__restrict__ scal, double *__restrict__ partials,
                                                            double2 *__restrict__ rs_all, double2 *__restrict__ rs_gh,
                                                            const double *__restrict__ hsc, const double4 *__restrict__ pqrow,
                                                            const int *__restrict__ grouplist, int ngroups, int pbase, const double *__restrict__ stopflag,
                                                            const int *__restrict__ gflags = nullptr, int roword = 0) {
  // roword (round 6, run-ahead CG loop): the per-row operands and results of the tail -- type, (hs,ht), (gs,gt), row length in; the two row sums out -- are
  // indexed by the row's PLACE in rows_sorted (group * WIN_ROWS + wavefront) instead of by its atom: the 16 wavefronts of a workgroup then touch one
  // contiguous run per array instead of 16 scattered 64-byte sectors (the caller hands over the row-ordered arrays)
  if (stopflag && *stopflag != 0.0) return;
  extern __shared__ double2 s_x[];                  // the window: slot -> (xs, xt)
  __shared__ double s_row[WIN_ROWS][4];
  __shared__ int s_arrived;
  constexpr int STEPS = NSTEP;
  constexpr int NT = 64 * WIN_ROWS;
  typedef double d2v __attribute__((ext_vector_type(2)));
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int gidx = xcd_swizzle(blockIdx.x, gridDim.x);
  const int grp = grouplist ? (gidx < ngroups ? grouplist[gidx] : -1) : (gidx < ngroups ? gidx : -1);
  if (grp < 0) return;                              // (whole workgroup)
  // Round trip 1: everything that needs only the group number -- the row of this wavefront, the window's size, the first positions of the
  // window units this thread will copy (two rounds of 1,024 slots cover 256 units; the descriptor row is WIN_MAXUNITS long whatever the count).
  const int ridx = grp * WIN_ROWS + wave;
  const int row = min(rows_sorted[ridx], N);        // (a cell column's last group may be short: those slots hold a sentinel >= N)
  const int nslots = WIN_UNIT * win_cnt[grp];
  const int *wk = win_k + static_cast<size_t>(grp) * WIN_MAXUNITS;
  const int t0 = threadIdx.x, t1 = threadIdx.x + NT;
  const int wk0 = wk[t0 / WIN_UNIT], wk1 = wk[t1 / WIN_UNIT];                      // (t1 / 8 < 256 <= WIN_MAXUNITS)
  const bool live = row < N;
#ifdef RXMD_EXPERIMENTS
  const size_t base = static_cast<size_t>(live ? ((VAR & WIN_RANKROWS) ? ridx : row) : 0) * S10;       // (VAR & 1, experiment: the streams' rows in cell-sorted order)
#else
  const size_t base = static_cast<size_t>(live ? row : 0) * S10;
#endif
  const d2v *hv2 = reinterpret_cast<const d2v *>(hess + base);
  const d2v *cv2 = reinterpret_cast<const d2v *>((PQ ? hsc : hess) + base);
  const unsigned *sl2 = reinterpret_cast<const unsigned *>(sl10 + base);
  // VAR & 4: a group none of whose rows has a ghost partner (gflags[grp] == 0: three groups in four of a large domain) takes a body WITHOUT the
  // ghost-column sums -- two selects and four FMAs of the twelve vector instructions per pair of entries, and two of the four reductions.  The flag
  // rides with round trip 1 and the choice is ONE scalar branch per workgroup around two complete copies of everything below (round 6: as two loops
  // inside one body the compiler kept the first batches alive for "the other loop" and paid with folded spills under the 64-register cap -- 48 B of
  // scratch per lane, 20 after the register sets took turns, 0 now).
  const bool gh = !((VAR & WIN_LEAN) != 0 && !PQ && gflags != nullptr) || gflags[grp] != 0;
  auto whole = [&](auto ghc) {
  constexpr bool GHC = decltype(ghc)::value;
  double2 v[STEPS], c[STEPS]; unsigned ss[STEPS];
  auto request = [&](int kb, int bound) {          // entries kb + 128 u + 2 lane and the next one
#pragma unroll
    for (int u = 0; u < STEPS; ++u) {
      const int k = kb + 128 * u + 2 * lane;
      const bool ok = k < bound;
      if (ok) { const d2v t2 = __builtin_nontemporal_load(hv2 + (k >> 1)); v[u] = make_double2(t2.x, t2.y); } else v[u] = make_double2(0.0, 0.0);
      ss[u] = ok ? __builtin_nontemporal_load(sl2 + (k >> 1)) : 0u;
      if (PQ && (MODE == MODE_GRAD || STORE)) { if (ok) { const d2v t2 = __builtin_nontemporal_load(cv2 + (k >> 1)); c[u] = make_double2(t2.x, t2.y); } else c[u] = make_double2(0.0, 0.0); }
    }
  };
  // Round trip 2: the window's vector entries FIRST (they return first, and the workgroup's barrier waits for them only), then the row's
  // first batch -- before the row length is known: it lies inside the row's slot whatever the length -- the length, and the operands of the
  // row's tail (the row is the same for the 64 lanes: scalar loads, no vector registers).
  double2 x0 = make_double2(0.0, 0.0), x1 = x0;
  if (t0 < nslots) x0 = xv[min(wk0 + (t0 & (WIN_UNIT - 1)), G - 1)];
  if (t1 < nslots) x1 = xv[min(wk1 + (t1 & (WIN_UNIT - 1)), G - 1)];
  request(0, live ? S10 : 0);
  const int rowc = live ? (roword ? ridx : row) : 0;         // index of this row in the per-row arrays
  const int n = live ? (n10[rowc] & N10_COUNT) : 0;
  const int tl_t = type[rowc];
  const double2 tl_a = (MODE == MODE_HSH) ? hst[rowc] : qst[rowc];
  const double2 tl_b = (MODE == MODE_HSH) ? const_cast<const double2 *>(gst)[rowc] : make_double2(q[rowc], 0.0);
  if (threadIdx.x == 0) s_arrived = 0;
  if (t0 < nslots) s_x[t0] = x0;
  if (t1 < nslots) s_x[t1] = x1;
  for (int t = threadIdx.x + 2 * NT; t < nslots; t += NT) s_x[t] = xv[min(wk[t / WIN_UNIT] + (t & (WIN_UNIT - 1)), G - 1)];   // a window of more than 256 units
#pragma unroll
  for (int u = 0; u < STEPS; ++u) {                // entries behind the row's end: weight 0, slot 0
    const int k = 128 * u + 2 * lane;
    if (k >= n) { v[u].x = 0.0; ss[u] &= 0xffff0000u; if (PQ) c[u].x = 0.0; }
    if (k + 1 >= n) { v[u].y = 0.0; ss[u] &= 0x0000ffffu; if (PQ) c[u].y = 0.0; }
  }
  // VAR & 2 (plain QEq): the row's NEXT batch is requested before the barrier, behind the window's data (a wavefront's loads return in order:
  // the window does not wait for it) -- the round trip of the second batch runs under the barrier and the first batch's arithmetic instead of
  // after it; the batch after that is requested before the current one is used, and so on.
  // PQEq takes the prefetch form with ONE batch of 128 entries per register set (NSTEP 1): its third stream (the shell-core values) doubles the registers of a
  // set, and two sets of two batches do not fit the 64 registers of eight workgroups per CU
  constexpr bool PRE = (VAR & WIN_PREFETCH) != 0 && (!PQ || NSTEP == 1);
  double2 vn[STEPS], cn[STEPS]; unsigned sn[STEPS];
  auto request_next = [&](int kb) {
#pragma unroll
    for (int u = 0; u < STEPS; ++u) {
      const int k = kb + 128 * u + 2 * lane;
      const bool ok = k < n;
      if (ok) { const d2v t2 = __builtin_nontemporal_load(hv2 + (k >> 1)); vn[u] = make_double2(t2.x, t2.y); } else vn[u] = make_double2(0.0, 0.0);
      sn[u] = ok ? __builtin_nontemporal_load(sl2 + (k >> 1)) : 0u;
      if (PQ && (MODE == MODE_GRAD || STORE)) { if (ok) { const d2v t2 = __builtin_nontemporal_load(cv2 + (k >> 1)); cn[u] = make_double2(t2.x, t2.y); } else cn[u] = make_double2(0.0, 0.0); }
    }
  };
  if (PRE && n > 128 * STEPS) request_next(128 * STEPS);
  __syncthreads();
  double as = 0.0, at = 0.0, gs_ = 0.0, gt_ = 0.0;
    auto batch = [&](const double2 (&vv)[STEPS], const unsigned (&sv)[STEPS], const double2 (&cv)[STEPS]) {      // one batch of 128 x STEPS entries out of registers and the LDS window
#pragma unroll
      for (int u = 0; u < STEPS; ++u) {
        const double2 y0 = s_x[sv[u] & 0x7fffu], y1 = s_x[(sv[u] >> 16) & 0x7fffu];
        as += vv[u].x * y0.x; at += vv[u].x * y0.y; as += vv[u].y * y1.x; at += vv[u].y * y1.y;
        if (GHC && (MODE == MODE_GRAD || STORE) && !PQ) {
          const double g0 = (sv[u] & 0x8000u) ? vv[u].x : 0.0, g1 = (sv[u] & 0x80000000u) ? vv[u].y : 0.0;     // select the weight, not the sums
          gs_ += g0 * y0.x; gt_ += g0 * y0.y; gs_ += g1 * y1.x; gt_ += g1 * y1.y;
        }
        if ((MODE == MODE_GRAD || STORE) && PQ) { gs_ += cv[u].x * y0.x; gt_ += cv[u].x * y0.y; gs_ += cv[u].y * y1.x; gt_ += cv[u].y * y1.y; }
      }
    };
    constexpr int B = 128 * STEPS;
    if (PRE) {
      // Two register sets take turns (round 6): batch k sits in (v, ss) for even k and in (vn, sn) for odd k, and the batch after the next one is
      // requested into the set that has just been used.  Until round 6 the loop rotated `v = vn` at its head: under the 64-register cap of eight
      // workgroups per CU that copy went through 48 bytes of scratch per lane on every row longer than two batches (rows > 512 entries: any
      // condensed system at PQEq's 12.5 A cut-off, dense metals) -- and cost ten moves per row on the others.  Same products in the same order.
      for (int kb = 0;;) {                               // wave-uniform trip count
        batch(v, ss, c);
        kb += B; if (kb >= n) break;
        if (kb + B < n) request(kb + B, n);               // (an odd row end: entry n is the zero padding of the row, slot 0)
        batch(vn, sn, cn);
        kb += B; if (kb >= n) break;
        if (kb + B < n) request_next(kb + B);
      }
    } else {
      for (int kb = 0; kb < n; kb += B) {                 // wave-uniform trip count
        if (kb > 0) request(kb, n);
        batch(v, ss, c);
      }
    }

  as = wave_sum(as); at = wave_sum(at);
  if ((MODE == MODE_GRAD || STORE) && GHC) { gs_ = wave_sum(gs_); gt_ = wave_sum(gt_); }
  // The row's tail by its own wavefront (the rows of a group are scattered residents: nothing would coalesce if one wavefront ran all of them,
  // and a workgroup whose last wavefront works alone keeps 15 wavefront slots of the CU empty); a wavefront leaves when it is done.  The
  // workgroup's four partial sums: every wavefront leaves its terms in LDS, the LAST one to arrive adds them in wavefront order.
  // the lane number afresh (two mbcnt instructions): `lane` kept alive across the loop for these four uses was the one register the PQ instances spilled
  const int lane_t = static_cast<int>(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)));
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  if (live) {
    const DevAtomP ap = ff.atom[tl_t];
    if (MODE == MODE_HSH) {
      const double ts = ap.eta * tl_a.x + as, tt = ap.eta * tl_a.y + at;      // qeq.F90:294-302
      a0 = ts * tl_a.x; a1 = tt * tl_a.y;                                     // hshs_sum, hsht_sum (:309-310)
      a2 = tl_b.x * tl_a.x; a3 = tl_b.y * tl_a.y;                             // g.h (:119,123)
      if (STORE && lane_t == 0) { rs_all[rowc] = make_double2(as, at); rs_gh[rowc] = make_double2(gs_, gt_); }
    } else {
      const double mu = scal[S_MU];
      const double4 pr = PQ ? pqrow[row] : make_double4(0.0, 0.0, 0.0, 0.0);
      const double g1 = -ap.chi - ap.eta * tl_a.x - as - pr.x;                // qeq.F90:349-350 (pqeq.F90:466)
      const double g2 = -1.0 - ap.eta * tl_a.y - at;
      a0 = g1 * g1; a1 = g2 * g2;                                             // Gnew (:355-356)
      const double qi = tl_b.x;
      const double hq_all = as - mu * at, hq_res = (as - gs_) - mu * (at - gt_);
      if (PQ) a2 = pq_est_row(ap, ff.Zpq[tl_t], pr, qi, hq_all, gs_ - mu * gt_);
      else a2 = ap.chi * qi + 0.5 * ap.eta * qi * qi + 0.5 * qi * (hq_all + hq_res);  // Est (:297-306)
      if (lane_t == 0) { gst[rowc] = make_double2(g1, g2); if (STORE) { rs_all[rowc] = make_double2(as, at); rs_gh[rowc] = make_double2(gs_, gt_); } }
    }
  }
  int arrived = 0;
  if (lane_t == 0) {
    s_row[wave][0] = a0; s_row[wave][1] = a1; s_row[wave][2] = a2; s_row[wave][3] = a3;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    arrived = __hip_atomic_fetch_add(&s_arrived, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  arrived = __builtin_amdgcn_readfirstlane(arrived);
  if (arrived != WIN_ROWS - 1) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  if (lane_t < 4) {
    double sum = 0.0;
#pragma unroll
    for (int w = 0; w < WIN_ROWS; ++w) sum += s_row[w][lane_t];                 // fixed order: the result does not depend on which wavefront is last
    __hip_atomic_store(partials + (static_cast<size_t>(pbase) + blockIdx.x) * 4 + lane_t, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  };
  if (gh) whole(std::true_type{}); else whole(std::false_type{});
}

}  // namespace rxmd
