// qeq.hip -- two-vector QEq conjugate gradient on the device (reference src/qeq.F90:2-178).
//   get_hsh      (qeq.F90:271-318) -> k_spmv<MODE_HSH>    matrix pass over (hs,ht)
//   get_gradient (qeq.F90:321-363) -> k_spmv<MODE_GRAD>   matrix pass over (qs,qt)
//   driver loop  (qeq.F90:96-166)  -> Engine::qeq()       REAL(4) step lengths kept (qeq.F90:23,133)
// The matrix is the ELL list built in lists.hip: per row a contiguous FP64 value stream and an INT32
// column stream (12 bytes per entry, the algorithmic bytes of SURVEY 8d), one wavefront per row.
// The two right-hand vectors are interleaved (double2) so a column costs ONE 16-byte gather, and the gather
// source is a CELL-SORTED copy (ghosts resolved to their owners while copying = the reference's QCOPY halo),
// so the 64 lanes of a wavefront touch a few cache lines instead of 64.
//
// Est (the convergence quantity, qeq.F90:297-306) is produced by the GRADIENT pass of the previous
// iteration: sum_j H_ij q_j = sum_j H_ij qs_j - mu sum_j H_ij qt_j, with the reference's
// "count resident partners twice" rule kept through a second accumulator over ghost columns.
#include "engine.h"

#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace rxmd {

static inline int nblk(long long n, int b) { return n > 0 ? static_cast<int>((n + b - 1) / b) : 1; }   // an empty rank still launches (kernels guard their range)

enum { S_MU = 0, S_LMIN_S, S_LMIN_T, S_GOLD_S, S_GOLD_T, S_GNEW_S, S_GNEW_T, S_EST, S_GH_S, S_GH_T, S_HSH_S, S_HSH_T, S_SSUM, S_TSUM, S_BETA_S, S_BETA_T, S_RAW0, S_RAW1, S_RAW2, S_RAW3, S_RAW4, S_RAW5, S_RAW6, S_RAW7, S_STOP, S_STOP1, S_TOL, S_COUNT };
enum { MODE_HSH = 0, MODE_GRAD = 1 };
typedef double f64x2 __attribute__((ext_vector_type(2)));
#ifndef SPMV_UNR
#define SPMV_UNR 4
#endif
constexpr int UNR = SPMV_UNR;
#ifndef SPMV_LEAN_GH
#define SPMV_LEAN_GH 0
#endif
#ifndef SPMV_LEAN_FULL
#define SPMV_LEAN_FULL 0
#endif
#ifndef SPMV_WG_TAIL
#define SPMV_WG_TAIL 1
#endif
constexpr bool WG_TAIL = SPMV_WG_TAIL != 0;   // row tails of a workgroup run together by its wavefront 0 (k_spmv)
constexpr bool LEAN_GH = SPMV_LEAN_GH != 0, LEAN_FULL = SPMV_LEAN_FULL != 0;   // measured and left off: ghost sums only on boundary rows / complete batches without bounds -- fewer vector instructions, not faster   // 4 x 64 = 256 entries in flight per wavefront and pass of the row loop (see k_spmv)

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
// more scalar round trip in front of the loop) changed nothing or lost.  All were dropped again; see DESIGN.md 3.
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
  // WG_TAIL: the row tails of a workgroup are run by the first lanes of its wavefront 0 after the barrier the partial sums need anyway -- the
  // operands of consecutive rows (type, hst / qst, gst / q) and their results (row sums, gradient) are then a handful of coalesced requests
  // per workgroup instead of five per row (k_spmv_bisect: tail operands, row stores and partials are 6-8 % of the pass)
  __shared__ double s_row[16][4];
  const int wave_in_wg = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
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
    auto request_full = [&](int kb, unsigned (&ee)[UNR], double (&hh)[UNR], double (&cc)[UNR]) {      // a complete batch needs no bounds
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int k = kb + lane + 64 * u;
        ee[u] = static_cast<unsigned>(__builtin_nontemporal_load(nb10 + base + k));
        hh[u] = __builtin_nontemporal_load(hess + base + k);
        if (PQ && (MODE == MODE_GRAD || STORE)) cc[u] = __builtin_nontemporal_load(hsc + base + k);
      }
    };
    if (PIPE) { if (LEAN_FULL && S10 >= 64 * UNR) request_full(0, e, h, c); else request(0, S10, e, h, c); }
    const int nraw = n10[row];
    const int n = nraw & N10_COUNT;
    // plain QEq: the sums over ghost columns exist only on boundary rows of the domain (the flag rides in the row length, engine.h); on the
    // other rows -- three in four at 979,776 atoms -- the six vector instructions per batch that select and add them, and their two
    // wavefront reductions, are not executed (counters: 306 vector instructions per row before, the vector unit busy 51 % of the time)
    const bool gh_row = PQ || !LEAN_GH || (nraw & N10_GHOST_ROW) != 0;
    // per-wavefront tail (WG_TAIL off): its operands are requested before the streams so that they are not a further dependent round trip
    int pf_t = 0; double2 pf_a = make_double2(0.0, 0.0), pf_b = make_double2(0.0, 0.0);
    if (!WG_TAIL) { pf_t = type[row]; pf_a = (MODE == MODE_HSH) ? hst[row] : qst[row]; pf_b = (MODE == MODE_HSH) ? gst[row] : make_double2(q[row], 0.0); }
    const double mu = (MODE == MODE_GRAD) ? scal[S_MU] : 0.0;
    double as = 0.0, at = 0.0, gs_ = 0.0, gt_ = 0.0;
    if (PIPE && (!LEAN_FULL || n < 64 * UNR)) {                  // a row shorter than the first batch: entries behind its end get weight 0 (longer rows skip the re-masking)
#pragma unroll
      for (int u = 0; u < UNR; ++u) { const bool ok = lane + 64 * u < n; e[u] = ok ? e[u] : 0u; h[u] = ok ? h[u] : 0.0; if (PQ) c[u] = ok ? c[u] : 0.0; }
    }
    auto accumulate = [&](auto ghc) {
      constexpr bool GH = decltype(ghc)::value;
#pragma unroll
      for (int u = 0; u < UNR; ++u) {            // one 16-byte gather per entry from the cell-sorted vector copy
        const double2 v = xv[e[u] & NB10_IDX_MASK];
        as += h[u] * v.x;
        at += h[u] * v.y;
        if (GH && (MODE == MODE_GRAD || STORE) && !PQ) { const double hg = (e[u] & NB10_GHOST) ? h[u] : 0.0; gs_ += hg * v.x; gt_ += hg * v.y; }   // select the weight, not the sums
        if ((MODE == MODE_GRAD || STORE) && PQ) { gs_ += c[u] * v.x; gt_ += c[u] * v.y; }      // PQEq: second matrix (shell-core) over the same columns
      }
    };
    for (int kb = 0; kb < n; kb += 64 * UNR) {   // wave-uniform trip count
      if (PIPE == 0 || (PIPE == 1 && kb > 0)) {
        if (LEAN_FULL && kb + 64 * UNR <= n) request_full(kb, e, h, c); else request(kb, n, e, h, c);
      }
      if (!LEAN_GH || gh_row) accumulate(std::true_type{}); else accumulate(std::false_type{});
    }
    as = wave_sum(as); at = wave_sum(at);
    if ((MODE == MODE_GRAD || STORE) && (!LEAN_GH || gh_row)) { gs_ = wave_sum(gs_); gt_ = wave_sum(gt_); }
    if (WG_TAIL) { if (lane == 0) { s_row[wave_in_wg][0] = as; s_row[wave_in_wg][1] = at; s_row[wave_in_wg][2] = gs_; s_row[wave_in_wg][3] = gt_; } }
    else if (lane == 0) {
      if (STORE) { rs_all[row] = make_double2(as, at); rs_gh[row] = make_double2(gs_, gt_); }
      const DevAtomP ap = ff.atom[pf_t];
      if (MODE == MODE_HSH) {
        const double ts = ap.eta * pf_a.x + as, tt = ap.eta * pf_a.y + at;      // qeq.F90:294-302
        acc[0] = ts * pf_a.x; acc[1] = tt * pf_a.y;                             // hshs_sum, hsht_sum (:309-310)
        acc[2] = pf_b.x * pf_a.x; acc[3] = pf_b.y * pf_a.y;                     // g.h (:119,123)
      } else {
        const double fpq = PQ ? pqrow[row].x : 0.0;
        const double g1 = -ap.chi - ap.eta * pf_a.x - as - fpq;                 // qeq.F90:349-350 (pqeq.F90:466)
        const double g2 = -1.0 - ap.eta * pf_a.y - at;
        gst[row] = make_double2(g1, g2);
        acc[0] = g1 * g1; acc[1] = g2 * g2;                                     // Gnew (:355-356)
        const double qi = pf_b.x;
        const double hq_all = as - mu * at, hq_res = (as - gs_) - mu * (at - gt_);
        if (PQ) acc[2] = pq_est_row(ap, ff.Zpq[pf_t], pqrow[row], qi, hq_all, gs_ - mu * gt_);
        else acc[2] = ap.chi * qi + 0.5 * ap.eta * qi * qi + 0.5 * qi * (hq_all + hq_res);  // Est (:297-306)
      }
    }
  }
  if (!WG_TAIL) { block_store_partials<4>(acc, partials + static_cast<size_t>(pbase) * 4, 4); return; }
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
// Timing probe with synthetic slots before it was built (debug tap 103): 0.76 ms against 0.93 ms of k_spmv on the same box.
template <int MODE, bool STORE, bool PQ>
__global__ void __launch_bounds__(64 * WIN_ROWS, 8) k_spmv_win(int N, int G, int S10, DevFF ff, const unsigned short *__restrict__ sl10, const double *__restrict__ hess, const int *__restrict__ n10,
                                                            const int *__restrict__ rows_sorted, const int *__restrict__ win_k, const int *__restrict__ win_cnt,
                                                            const double2 *__restrict__ xv, const double2 *__restrict__ hst, double2 *__restrict__ gst,
                                                            const double2 *__restrict__ qst, const double *__restrict__ q, const int *__restrict__ type,
                                                            const double *__restrict__ scal, double *__restrict__ partials,
                                                            double2 *__restrict__ rs_all, double2 *__restrict__ rs_gh,
                                                            const double *__restrict__ hsc, const double4 *__restrict__ pqrow,
                                                            const int *__restrict__ grouplist, int ngroups, int pbase, const double *__restrict__ stopflag) {
  if (stopflag && *stopflag != 0.0) return;
  extern __shared__ double2 s_x[];                  // the window: slot -> (xs, xt)
  __shared__ double s_row[WIN_ROWS][4];
  constexpr int STEPS = 2;                          // 2 x 128 entries of the row in flight
  constexpr int NT = 64 * WIN_ROWS;
  typedef double d2v __attribute__((ext_vector_type(2)));
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int gidx = xcd_swizzle(blockIdx.x, gridDim.x);
  const int grp = grouplist ? (gidx < ngroups ? grouplist[gidx] : -1) : (gidx < ngroups ? gidx : -1);
  if (grp < 0) return;                              // (whole workgroup)
  // Round trip 1: everything that needs only the group number -- the row of this wavefront, the window's size, the first positions of the
  // window units this thread will copy (two rounds of 1,024 slots cover 256 units; the descriptor row is WIN_MAXUNITS long whatever the
  // count), and for wavefront 0 the rows whose tails it runs at the end.
  const int ridx = grp * WIN_ROWS + wave;
  const int row = ridx < N ? rows_sorted[ridx] : N;
  const int nslots = WIN_UNIT * win_cnt[grp];
  const int *wk = win_k + static_cast<size_t>(grp) * WIN_MAXUNITS;
  const int t0 = threadIdx.x, t1 = threadIdx.x + NT;
  const int wk0 = wk[t0 / WIN_UNIT], wk1 = wk[t1 / WIN_UNIT];                      // (t1 / 8 < 256 <= WIN_MAXUNITS)
  const int r_tail = (wave == 0 && lane < WIN_ROWS && grp * WIN_ROWS + lane < N) ? rows_sorted[grp * WIN_ROWS + lane] : N;
  const bool live = row < N;
  const size_t base = static_cast<size_t>(live ? row : 0) * S10;
  const d2v *hv2 = reinterpret_cast<const d2v *>(hess + base);
  const d2v *cv2 = reinterpret_cast<const d2v *>((PQ ? hsc : hess) + base);
  const unsigned *sl2 = reinterpret_cast<const unsigned *>(sl10 + base);
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
  // first batch -- before the row length is known: it lies inside the row's slot whatever the length -- the length, the tail operands.
  double2 x0 = make_double2(0.0, 0.0), x1 = x0;
  if (t0 < nslots) x0 = xv[min(wk0 + (t0 & (WIN_UNIT - 1)), G - 1)];
  if (t1 < nslots) x1 = xv[min(wk1 + (t1 & (WIN_UNIT - 1)), G - 1)];
  request(0, live ? S10 : 0);
  const int n = live ? (n10[row] & N10_COUNT) : 0;
  constexpr bool EARLY_TAIL = !PQ;                  // (the PQEq instances have no registers to spare: their tail operands are requested at the end)
  int tl_t = 0; double2 tl_a = make_double2(0.0, 0.0), tl_b = tl_a; double4 tl_p = make_double4(0.0, 0.0, 0.0, 0.0);
  auto tail_operands = [&]() {
    if (r_tail < N) {
      tl_t = type[r_tail];
      tl_a = (MODE == MODE_HSH) ? hst[r_tail] : qst[r_tail];
      tl_b = (MODE == MODE_HSH) ? gst[r_tail] : make_double2(q[r_tail], 0.0);
      if (PQ && MODE == MODE_GRAD) tl_p = pqrow[r_tail];
    }
  };
  if (EARLY_TAIL) tail_operands();
  if (t0 < nslots) s_x[t0] = x0;
  if (t1 < nslots) s_x[t1] = x1;
  for (int t = threadIdx.x + 2 * NT; t < nslots; t += NT) s_x[t] = xv[min(wk[t / WIN_UNIT] + (t & (WIN_UNIT - 1)), G - 1)];   // a window of more than 256 units
#pragma unroll
  for (int u = 0; u < STEPS; ++u) {                // entries behind the row's end: weight 0, slot 0
    const int k = 128 * u + 2 * lane;
    if (k >= n) { v[u].x = 0.0; ss[u] &= 0xffff0000u; if (PQ) c[u].x = 0.0; }
    if (k + 1 >= n) { v[u].y = 0.0; ss[u] &= 0x0000ffffu; if (PQ) c[u].y = 0.0; }
  }
  __syncthreads();
  double as = 0.0, at = 0.0, gs_ = 0.0, gt_ = 0.0;
  for (int kb = 0; kb < n; kb += 128 * STEPS) {     // wave-uniform trip count
    if (kb > 0) request(kb, n);                     // (an odd row end: entry n is the zero padding of the row, slot 0)
#pragma unroll
    for (int u = 0; u < STEPS; ++u) {
      const double2 y0 = s_x[ss[u] & 0x7fffu], y1 = s_x[(ss[u] >> 16) & 0x7fffu];
      as += v[u].x * y0.x; at += v[u].x * y0.y; as += v[u].y * y1.x; at += v[u].y * y1.y;
      if ((MODE == MODE_GRAD || STORE) && !PQ) {
        const double g0 = (ss[u] & 0x8000u) ? v[u].x : 0.0, g1 = (ss[u] & 0x80000000u) ? v[u].y : 0.0;     // select the weight, not the sums
        gs_ += g0 * y0.x; gt_ += g0 * y0.y; gs_ += g1 * y1.x; gt_ += g1 * y1.y;
      }
      if ((MODE == MODE_GRAD || STORE) && PQ) { gs_ += c[u].x * y0.x; gt_ += c[u].x * y0.y; gs_ += c[u].y * y1.x; gt_ += c[u].y * y1.y; }
    }
  }
  as = wave_sum(as); at = wave_sum(at);
  if (MODE == MODE_GRAD || STORE) { gs_ = wave_sum(gs_); gt_ = wave_sum(gt_); }
  if (lane == 0) { s_row[wave][0] = as; s_row[wave][1] = at; s_row[wave][2] = gs_; s_row[wave][3] = gt_; }
  __syncthreads();
  if (wave != 0) return;
  {                                                 // the row tails of the group, lane r = the row of wavefront r (as k_spmv); operands are here already
    const int r = r_tail;
    if (!EARLY_TAIL) tail_operands();
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (r < N) {
      const double ras = s_row[lane][0], rat = s_row[lane][1], rgs = s_row[lane][2], rgt = s_row[lane][3];
      const double2 pa = tl_a;
      const DevAtomP ap = ff.atom[tl_t];
      if (STORE) { rs_all[r] = make_double2(ras, rat); rs_gh[r] = make_double2(rgs, rgt); }
      if (MODE == MODE_HSH) {
        const double2 pb = tl_b;
        const double ts = ap.eta * pa.x + ras, tt = ap.eta * pa.y + rat;        // qeq.F90:294-302
        a0 = ts * pa.x; a1 = tt * pa.y;                                         // hshs_sum, hsht_sum (:309-310)
        a2 = pb.x * pa.x; a3 = pb.y * pa.y;                                     // g.h (:119,123)
      } else {
        const double mu = scal[S_MU];
        const double fpq = PQ ? tl_p.x : 0.0;
        const double g1 = -ap.chi - ap.eta * pa.x - ras - fpq;                  // qeq.F90:349-350 (pqeq.F90:466)
        const double g2 = -1.0 - ap.eta * pa.y - rat;
        gst[r] = make_double2(g1, g2);
        a0 = g1 * g1; a1 = g2 * g2;                                             // Gnew (:355-356)
        const double qi = tl_b.x;
        const double hq_all = ras - mu * rat, hq_res = (ras - rgs) - mu * (rat - rgt);
        if (PQ) a2 = pq_est_row(ap, ff.Zpq[tl_t], tl_p, qi, hq_all, rgs - mu * rgt);
        else a2 = ap.chi * qi + 0.5 * ap.eta * qi * qi + 0.5 * qi * (hq_all + hq_res);  // Est (:297-306)
      }
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2); a3 = wave_sum(a3);           // fixed order over the group's rows
    if (lane < 4) {
      const double val = lane == 0 ? a0 : (lane == 1 ? a1 : (lane == 2 ? a2 : a3));
      __hip_atomic_store(partials + (static_cast<size_t>(pbase) + blockIdx.x) * 4 + lane, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ---- two rows per wavefront, SIDE BY SIDE (k_spmv2) -----------------------------------------------------------------------------
// Every form of the pass measured so far turns a row around in 0.92-0.97 ns chip-wide whatever the row holds -- RDX 430 entries, water 357,
// the SiC + O2 system 212 with a third stream -- i.e. a CU finishes a row every 240 ns with its 32 wavefronts, each alive 6.5 us: the fixed
// part of a wavefront's life (launch, kernel arguments, row length, first batch, gathers, reduction, tail, the workgroup barrier of the
// partial sums) bounds the pass, and 32 wavefronts per CU is the hardware's limit.  Here a wavefront holds TWO rows at the same time, lanes
// 0-31 one and lanes 32-63 the other: 64 rows in flight per CU, the fixed part paid once per pair.  (Two rows one AFTER the other in a
// wavefront -- round 2 -- left the rows in flight at 32 and lost.)  A lane walks its row with stride 32; the four sums of a row are reduced
// inside its half (five DPP steps: the four inside a row of 16 lanes, then row 0 -> 1 and 2 -> 3); lane 16 of each half runs the row tail.
__device__ inline double half_sum32(double v) {    // sum over the 32 lanes of a half; valid in lanes 16-31 (first half) and 48-63 (second half)
  v += dpp_move<0xb1, 0xf>(v);
  v += dpp_move<0x4e, 0xf>(v);
  v += dpp_move<0x141, 0xf>(v);
  v += dpp_move<0x140, 0xf>(v);
  v += dpp_move<0x142, 0xa>(v);                    // row_bcast:15 into rows 1 and 3
  return v;
}
template <int MODE, bool STORE, bool PQ>
__global__ void __launch_bounds__(1024) k_spmv2(int N, int S10, DevFF ff, const int *__restrict__ nb10, const double *__restrict__ hess, const int *__restrict__ n10,
                                                const double2 *__restrict__ xv, const double2 *__restrict__ hst, double2 *__restrict__ gst,
                                                const double2 *__restrict__ qst, const double *__restrict__ q, const int *__restrict__ type,
                                                const double *__restrict__ scal, double *__restrict__ partials,
                                                double2 *__restrict__ rs_all, double2 *__restrict__ rs_gh,
                                                const double *__restrict__ hsc, const double4 *__restrict__ pqrow, int swz,
                                                const int *__restrict__ rowlist, int nrows, int pbase, const double *__restrict__ stopflag) {
  if (stopflag && *stopflag != 0.0) return;
#ifndef SPMV2_U
#define SPMV2_U 8
#endif
  constexpr int U2 = SPMV2_U;                       // U2 x 32 entries of each row in flight per trip
  const int lane = threadIdx.x & 63, hl = lane & 31, half = lane >> 5;
  const int wpb = blockDim.x >> 6;
  const int widx = (swz ? xcd_swizzle(blockIdx.x, gridDim.x) : static_cast<int>(blockIdx.x)) * wpb + __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int ridx = 2 * widx + half;
  const int row = rowlist ? (ridx < nrows ? rowlist[ridx] : N) : (ridx < nrows ? ridx : N);
  const bool live = row < N;
  const int rowc = live ? row : 0;
  const size_t base = static_cast<size_t>(rowc) * S10;
  unsigned e[U2];
  double h[U2], c[U2];
  auto request = [&](int kb, int bound) {
#pragma unroll
    for (int u = 0; u < U2; ++u) {
      const int k = kb + hl + 32 * u;
      const bool ok = k < bound;
      e[u] = ok ? static_cast<unsigned>(__builtin_nontemporal_load(nb10 + base + k)) : 0u;
      h[u] = ok ? __builtin_nontemporal_load(hess + base + k) : 0.0;
      if (PQ && (MODE == MODE_GRAD || STORE)) c[u] = ok ? __builtin_nontemporal_load(hsc + base + k) : 0.0;
    }
  };
  request(0, live ? S10 : 0);                       // before the row length is known: the first batch lies inside the row's slot whatever the length
  const int n = live ? (n10[rowc] & N10_COUNT) : 0;
  const int pf_t = type[rowc];
  const double2 pf_a = (MODE == MODE_HSH) ? hst[rowc] : qst[rowc];
  const double2 pf_b = (MODE == MODE_HSH) ? gst[rowc] : make_double2(q[rowc], 0.0);
  const double mu = (MODE == MODE_GRAD) ? scal[S_MU] : 0.0;
  const int nmax = max(__builtin_amdgcn_readlane(n, 0), __builtin_amdgcn_readlane(n, 32));
#pragma unroll
  for (int u = 0; u < U2; ++u) { const bool ok = hl + 32 * u < n; e[u] = ok ? e[u] : 0u; h[u] = ok ? h[u] : 0.0; if (PQ) c[u] = ok ? c[u] : 0.0; }
  double as = 0.0, at = 0.0, gs_ = 0.0, gt_ = 0.0;
  for (int kb = 0; kb < nmax; kb += 32 * U2) {     // wave-uniform trip count: the longer of the two rows
    if (kb > 0) request(kb, n);
#pragma unroll
    for (int u = 0; u < U2; ++u) {
      const double2 v = xv[e[u] & NB10_IDX_MASK];
      as += h[u] * v.x;
      at += h[u] * v.y;
      if ((MODE == MODE_GRAD || STORE) && !PQ) { const double hg = (e[u] & NB10_GHOST) ? h[u] : 0.0; gs_ += hg * v.x; gt_ += hg * v.y; }
      if ((MODE == MODE_GRAD || STORE) && PQ) { gs_ += c[u] * v.x; gt_ += c[u] * v.y; }
    }
  }
  as = half_sum32(as); at = half_sum32(at);
  if (MODE == MODE_GRAD || STORE) { gs_ = half_sum32(gs_); gt_ = half_sum32(gt_); }
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  if (hl == 16 && live) {
    if (STORE) { rs_all[row] = make_double2(as, at); rs_gh[row] = make_double2(gs_, gt_); }
    const DevAtomP ap = ff.atom[pf_t];
    if (MODE == MODE_HSH) {
      const double ts = ap.eta * pf_a.x + as, tt = ap.eta * pf_a.y + at;      // qeq.F90:294-302
      acc[0] = ts * pf_a.x; acc[1] = tt * pf_a.y;                             // hshs_sum, hsht_sum (:309-310)
      acc[2] = pf_b.x * pf_a.x; acc[3] = pf_b.y * pf_a.y;                     // g.h (:119,123)
    } else {
      const double fpq = PQ ? pqrow[row].x : 0.0;
      const double g1 = -ap.chi - ap.eta * pf_a.x - as - fpq;                 // qeq.F90:349-350 (pqeq.F90:466)
      const double g2 = -1.0 - ap.eta * pf_a.y - at;
      gst[row] = make_double2(g1, g2);
      acc[0] = g1 * g1; acc[1] = g2 * g2;                                     // Gnew (:355-356)
      const double qi = pf_b.x;
      const double hq_all = as - mu * at, hq_res = (as - gs_) - mu * (at - gt_);
      if (PQ) acc[2] = pq_est_row(ap, ff.Zpq[pf_t], pqrow[row], qi, hq_all, gs_ - mu * gt_);
      else acc[2] = ap.chi * qi + 0.5 * ap.eta * qi * qi + 0.5 * qi * (hq_all + hq_res);  // Est (:297-306)
    }
  }
  // the wavefront's partial = first row + second row (lanes 16 and 48), handed to lane 0 for the workgroup's fixed-order sum
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double a16 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(acc[k]), 16), __builtin_amdgcn_readlane(__double2loint(acc[k]), 16));
    const double a48 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(acc[k]), 48), __builtin_amdgcn_readlane(__double2loint(acc[k]), 48));
    acc[k] = a16 + a48;
  }
  block_store_partials<4>(acc, partials + static_cast<size_t>(pbase) * 4, 4);
}

// ---- the matrix pass as a packed stream through an LDS ring (LDS-DMA) ---------------------------------------------------------
// One persistent workgroup per CU owns a contiguous range of rows.  Wave 0 (the LOADER) streams the used part of consecutive rows --
// hessian values, packed entries, PQEq: shell-core values -- from their ELL slots straight into LDS rings with
// `global_load_lds_dwordx4 ... nt` (16 B per lane, 1 KiB per instruction, no VGPR destination; the per-lane SOURCE address skips the unused
// tail of every slot, the LDS destination is base + lane * 16, so the rows lie packed end to end in the ring).  Waves 1..C (the CONSUMERS)
// take the rows round-robin: wait until their row has landed, read entry + value from LDS, gather (hs,ht) / (qs,qt) from the
// cell-sorted copy, FMA, reduce, run the row tail.  What this removes from the wavefront-per-row kernel above: a million wavefront
// launches per pass, the row length as a dependent round trip in front of every row's loads, and the cap that VGPR-staged loads put on
// the bytes a CU keeps in flight (here: what the ring holds; the loader runs up to RING_V instructions = RING_V KiB ahead of the last
// row it has published).  One partial sum per workgroup (<= #CUs) instead of one per sixteen rows.
//   protocol (all in LDS): tab[t & 63] = (row, ring position, length) written by the loader before it issues row t of its sequence;
//   `landed` = number of rows whose data is in LDS (the loader counts its DMA instructions; after `s_waitcnt vmcnt(V)` all but the V
//   youngest have landed); crow[c] = sequence number of the oldest row consumer c still needs -- the loader re-uses ring space
//   behind min_c crow[c].  A row never wraps: when it does not fit before the ring's end it starts at 0.
//   Before the loader waits for space it drains its DMAs and publishes everything it has issued (no consumer can then wait for a row
//   that only further issues would publish).  One row always fits (host: S10 <= R), so the oldest row in flight is always processed.
#ifndef RING_V_DEF
#define RING_V_DEF 48
#endif
constexpr int RING_V = RING_V_DEF;            // DMA instructions the loader leaves in flight behind its publication point (vmcnt is 6 bits)
constexpr int RING_NL = 1;            // loader waves per workgroup: each issues every other DMA instruction of a row (measured: a second loader per workgroup does not raise the stream rate, 1.05 against 0.92 ms for the bare stream -- two workgroups per CU with one loader each do)
constexpr int RING_MAXC = 16 - RING_NL;   // consumer waves (with the loaders: 16 waves = the 1024-thread workgroup limit)
struct RingCtl { int landed[RING_NL]; int pad_[16 - RING_NL]; int crow[16]; int tab[64 * 4]; double acc[16][4]; double chi[16], eta[16], Zpq[16]; double res[16][6]; int resrow[16]; };   // chi/eta/Z per atom type: the row tail reads LDS, not a type -> parameter chain in HBM

__device__ inline void glds16_nt(const void *gsrc, unsigned lds_byte_addr) {   // M0 = LDS destination of lane 0; written in the statement that reads it
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}
// NI consecutive LDS-DMA instructions of one stream of one row: global source = sbase + voff (SGPR base + 32-bit lane offset, lane * 16)
// + 1024 * k, LDS destination = M0 + lane * 16 + 1024 * k (the immediate offset moves BOTH addresses: scripts/micro/glds_offset.hip),
// EXEC of instruction k = the first min(max(cnt - 64 k, 0), 64) lanes.  Seven scalar instructions and one DMA per KiB, no vector ALU;
// the caller runs with all lanes on.  An instruction whose EXEC is 0 moves nothing and still counts in vmcnt.
#ifndef RING_DMA_POLICY
#define RING_DMA_POLICY "nt"       // cache policy bits of the stream's DMA instructions (experiments: -DRING_DMA_POLICY='"sc1 nt"')
#endif
#define RX_GLDS_STEP(OFF)                                                                                                      \
  "s_max_i32 %[t], %[c], 0\n\ts_bfm_b64 exec, %[t], 0\n\ts_cmp_gt_i32 %[t], 63\n\ts_cmov_b64 exec, -1\n\t"                      \
  "global_load_lds_dwordx4 %[v], %[sb] offset:" #OFF " " RING_DMA_POLICY "\n\ts_sub_i32 %[c], %[c], 64\n\t"
template <int NI>
__device__ inline void glds16_nt_group(unsigned voff, const void *sbase, unsigned lds_byte_addr, int cnt) {
  static_assert(NI >= 1 && NI <= 4, "the immediate offset has 13 bits");
  int t;
  if (NI == 4) asm volatile("s_mov_b32 m0, %[dst]\n\t" RX_GLDS_STEP(0) RX_GLDS_STEP(1024) RX_GLDS_STEP(2048) RX_GLDS_STEP(3072) "s_mov_b64 exec, -1"
                            : [t] "=&s"(t), [c] "+s"(cnt) : [v] "v"(voff), [sb] "s"(sbase), [dst] "s"(lds_byte_addr) : "memory", "scc");
  if (NI == 3) asm volatile("s_mov_b32 m0, %[dst]\n\t" RX_GLDS_STEP(0) RX_GLDS_STEP(1024) RX_GLDS_STEP(2048) "s_mov_b64 exec, -1"
                            : [t] "=&s"(t), [c] "+s"(cnt) : [v] "v"(voff), [sb] "s"(sbase), [dst] "s"(lds_byte_addr) : "memory", "scc");
  if (NI == 2) asm volatile("s_mov_b32 m0, %[dst]\n\t" RX_GLDS_STEP(0) RX_GLDS_STEP(1024) "s_mov_b64 exec, -1"
                            : [t] "=&s"(t), [c] "+s"(cnt) : [v] "v"(voff), [sb] "s"(sbase), [dst] "s"(lds_byte_addr) : "memory", "scc");
  if (NI == 1) asm volatile("s_mov_b32 m0, %[dst]\n\t" RX_GLDS_STEP(0) "s_mov_b64 exec, -1"
                            : [t] "=&s"(t), [c] "+s"(cnt) : [v] "v"(voff), [sb] "s"(sbase), [dst] "s"(lds_byte_addr) : "memory", "scc");
}
template <class F, int... I> __device__ inline void static_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ inline void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }   // f(integral_constant<0>) ... f(integral_constant<N-1>)
__device__ inline unsigned lds_offset(const void *p) { return __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<size_t>(p))); }   // low 32 bits of a flat LDS address
__device__ inline int wave_min_int(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
  return v;
}
// ---- wavefront per row, the row's streams brought in by LDS-DMA (k_spmv_dma) -----------------------------------------------------
// The row kernel above keeps 4 x 64 entries of a row in flight because every staged load costs VGPRs.  Here a wavefront issues its WHOLE
// row at once -- value stream, entry words (PQEq: shell-core values) by `global_load_lds_dwordx4 ... nt` into its own LDS slot, no VGPR
// destination -- waits for it, and then runs the same gather / FMA / reduce / tail over LDS reads.  No roles, no flags: the only
// difference to k_spmv is where the streams land.  LDS bounds the residency: slot = rowcap * 12 (20) bytes per wavefront.
template <int MODE, bool STORE, bool PQ, int KH>
__global__ void __launch_bounds__(1024) k_spmv_dma(int N, int S10, DevFF ff, const int *__restrict__ nb10, const double *__restrict__ hess, const int *__restrict__ n10,
                                                   const double2 *__restrict__ xv, const double2 *__restrict__ hst, double2 *__restrict__ gst,
                                                   const double2 *__restrict__ qst, const double *__restrict__ q, const int *__restrict__ type,
                                                   const double *__restrict__ scal, double *__restrict__ partials,
                                                   double2 *__restrict__ rs_all, double2 *__restrict__ rs_gh,
                                                   const double *__restrict__ hsc, const double4 *__restrict__ pqrow, int swz,
                                                   const int *__restrict__ rowlist, int nrows, int pbase, int dma_n4) {
  extern __shared__ __attribute__((aligned(16))) char dma_smem[];
  constexpr bool PQS = PQ && (MODE == MODE_GRAD || STORE);
  constexpr int CAP = 128 * KH;                                     // entries per slot
  constexpr int SLOT = CAP * (PQS ? 20 : 12);                       // bytes: values, (shell-core values,) entry words
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  const int wv = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int widx = (swz ? xcd_swizzle(blockIdx.x, gridDim.x) : static_cast<int>(blockIdx.x)) * wpb + wv;
  const int row = rowlist ? (widx < nrows ? rowlist[widx] : N) : widx;
  char *slot = dma_smem + static_cast<size_t>(wv) * SLOT;
  const double *hl = reinterpret_cast<const double *>(slot);
  const double *cl = hl + (PQS ? CAP : 0);
  const int *il = reinterpret_cast<const int *>(hl + (PQS ? 2 : 1) * CAP);
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  if (row < N) {
    // dma_n4 > 0 (the rows of this build are nearly equally long: crystals, liquids): the streams are requested for the LONGEST row's length
    // without waiting for this row's -- the few entries behind its end lie inside its own ELL slot and get weight 0 below -- so nothing
    // stands between the wavefront's launch and its stream (the row length arrives while the stream is in flight)
    int nspec = dma_n4;
    const int n = (nspec > 0) ? 0 : (n10[row] & N10_COUNT);
    const int nd4 = (nspec > 0) ? nspec : ((n + 3) & ~3);
    int rowv = row;
    asm volatile("" : "+v"(rowv));                // tail operands by vector loads, requested before the streams
    const int pf_t = type[rowv];
    const double2 pf_a = (MODE == MODE_HSH) ? hst[rowv] : qst[rowv];
    const double2 pf_b = (MODE == MODE_HSH) ? gst[rowv] : make_double2(q[rowv], 0.0);
    const double mu = (MODE == MODE_GRAD) ? scal[S_MU] : 0.0;
    {
      const size_t base = static_cast<size_t>(row) * S10;
      const unsigned voff = static_cast<unsigned>(lane) * 16u;
      const unsigned hb = lds_offset(hl), cb = lds_offset(cl), ib = lds_offset(il);
      const int gh = nd4 >> 1, gi = nd4 >> 2;
#pragma unroll
      for (int g = 0; g < KH; g += 4) glds16_nt_group<4>(voff, hess + base + 128 * g, hb + 1024u * g, gh - 64 * g);
      if (PQS) {
#pragma unroll
        for (int g = 0; g < KH; g += 4) glds16_nt_group<4>(voff, hsc + base + 128 * g, cb + 1024u * g, gh - 64 * g);
      }
#pragma unroll
      for (int g = 0; g < KH / 2; g += 4) glds16_nt_group<(KH / 2 >= 4 ? 4 : KH / 2)>(voff, nb10 + base + 256 * g, ib + 1024u * g, gi - 64 * g);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const int n4 = (((nspec > 0) ? (n10[row] & N10_COUNT) : n) + 3) & ~3;
    double as = 0.0, at = 0.0, gs_ = 0.0, gt_ = 0.0;
    for (int kb = 0; kb < n4; kb += 64 * UNR) {
      unsigned e[UNR];
      double h[UNR], c[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int k = kb + lane + 64 * u;
        const bool ok = k < n4;
        e[u] = ok ? static_cast<unsigned>(il[k]) : 0u;
        h[u] = ok ? hl[k] : 0.0;
        if (PQS) c[u] = ok ? cl[k] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const double2 v = xv[e[u] & NB10_IDX_MASK];
        as += h[u] * v.x;
        at += h[u] * v.y;
        if ((MODE == MODE_GRAD || STORE) && !PQ) { const double hg = (e[u] & NB10_GHOST) ? h[u] : 0.0; gs_ += hg * v.x; gt_ += hg * v.y; }
        if ((MODE == MODE_GRAD || STORE) && PQ) { gs_ += c[u] * v.x; gt_ += c[u] * v.y; }
      }
    }
    as = wave_sum(as); at = wave_sum(at);
    if (MODE == MODE_GRAD || STORE) { gs_ = wave_sum(gs_); gt_ = wave_sum(gt_); }
    if (lane == 0) {
      if (STORE) { rs_all[row] = make_double2(as, at); rs_gh[row] = make_double2(gs_, gt_); }
      const DevAtomP ap = ff.atom[pf_t];
      if (MODE == MODE_HSH) {
        const double ts = ap.eta * pf_a.x + as, tt = ap.eta * pf_a.y + at;
        acc[0] = ts * pf_a.x; acc[1] = tt * pf_a.y;
        acc[2] = pf_b.x * pf_a.x; acc[3] = pf_b.y * pf_a.y;
      } else {
        const double fpq = PQ ? pqrow[row].x : 0.0;
        const double g1 = -ap.chi - ap.eta * pf_a.x - as - fpq;
        const double g2 = -1.0 - ap.eta * pf_a.y - at;
        gst[row] = make_double2(g1, g2);
        acc[0] = g1 * g1; acc[1] = g2 * g2;
        const double qi = pf_b.x;
        const double hq_all = as - mu * at, hq_res = (as - gs_) - mu * (at - gt_);
        if (PQ) acc[2] = pq_est_row(ap, ff.Zpq[pf_t], pqrow[row], qi, hq_all, gs_ - mu * gt_);
        else acc[2] = ap.chi * qi + 0.5 * ap.eta * qi * qi + 0.5 * qi * (hq_all + hq_res);
      }
    }
  }
  block_store_partials<4>(acc, partials + static_cast<size_t>(pbase) * 4, 4);
}

// Which rows a workgroup of the ring pass streams.  contiguous: workgroup w takes rows [w * per, (w + 1) * per).  cyclic (default when
// the grid is a multiple of 8): the 8 XCDs each own a contiguous eighth of the rows -- the gather vector of an XCD stays in its own L2 --
// and inside an XCD's range the rows go round-robin over its workgroups, so that at any moment the XCD reads ONE advancing window of the
// matrix (q consecutive rows) instead of q streams far apart: DRAM sees 8 sequential streams, not 512.
__host__ __device__ inline int ring_rows_per_wg(int nrows, int nwg) { return (nrows + nwg - 1) / nwg + 1; }
__host__ __device__ inline int ring_rows_of_wg(int nrows, int nwg, int w, int cyclic) {
  if (!cyclic) { const int per = (nrows + nwg - 1) / nwg; const int i0 = min(nrows, w * per); return min(nrows, i0 + per) - i0; }
  const int q = nwg >> 3, x = w / q, j = w - x * q;
  const int gb = static_cast<int>(static_cast<long long>(nrows) * x / 8), ge = static_cast<int>(static_cast<long long>(nrows) * (x + 1) / 8);
  return ge - gb > j ? (ge - gb - j + q - 1) / q : 0;
}
__global__ void __launch_bounds__(256) k_ring_schedule(int nrows, int nwg, int cyclic, const int *__restrict__ rowlist, const int *__restrict__ n10, int2 *__restrict__ sched) {
  const int per = ring_rows_per_wg(nrows, nwg);
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nwg * per) return;
  const int w = idx / per, k = idx - w * per;
  int2 o = make_int2(0, 0);
  if (k < ring_rows_of_wg(nrows, nwg, w, cyclic)) {
    int i;
    if (!cyclic) i = w * ((nrows + nwg - 1) / nwg) + k;
    else { const int q = nwg >> 3, x = w / q, j = w - x * q; i = static_cast<int>(static_cast<long long>(nrows) * x / 8) + j + q * k; }
    const int row = rowlist ? rowlist[i] : i;
    o = make_int2(row, n10[row] & N10_COUNT);
  }
  sched[idx] = o;
}
template <int MODE, bool STORE, bool PQ, int KH>     // KH: value-stream DMA instructions per row = rows of up to 128 * KH entries
__global__ void __launch_bounds__(1024, 8) k_spmv_ring(int N, int S10, DevFF ff, const int *__restrict__ nb10, const double *__restrict__ hess, const int *__restrict__ n10,
                                                    const double2 *__restrict__ xv, const double2 *__restrict__ hst, double2 *__restrict__ gst,
                                                    const double2 *__restrict__ qst, const double *__restrict__ q, const int *__restrict__ type,
                                                    const double *__restrict__ scal, double *__restrict__ partials,
                                                    double2 *__restrict__ rs_all, double2 *__restrict__ rs_gh,
                                                    const double *__restrict__ hsc, const double4 *__restrict__ pqrow, int swz,
                                                    const int2 *__restrict__ sched, int sched_cyclic, int nrows, int pbase, int R, int C) {
  extern __shared__ __attribute__((aligned(16))) char ring_smem[];
  double *hring = reinterpret_cast<double *>(ring_smem);
  double *cring = hring + (PQ ? R : 0);
  int *iring = reinterpret_cast<int *>(hring + (PQ ? 2 : 1) * static_cast<size_t>(R));
  RingCtl *ctl = reinterpret_cast<RingCtl *>(iring + R);
  // flags: relaxed workgroup-scope atomics = plain ds_read / ds_write (a `volatile` access is not given its address space back and
  // becomes a flat system-scope load behind s_waitcnt vmcnt(0) lgkmcnt(0): the consumer would wait for its own global stores)
  auto flag_load = [](const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
  auto flag_store = [](int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int probe = (swz >> 8) & 0xff;     // timing experiments only (RXMD_RING_PROBE): 1 consumers only release, 2 no gathers, 3 no DMA
  const bool grouped = (swz >> 16) & 1;      // consumers start the C rows of a group TOGETHER (when the group's last row has landed): consecutive rows share their vector lines in L1
  swz &= 0xff;
  const int wg = swz ? xcd_swizzle(blockIdx.x, gridDim.x) : static_cast<int>(blockIdx.x);
  // rows of this workgroup: entries [wg * per, wg * per + total) of the schedule (k_ring_schedule): (row, length) pairs in the order it streams them
  const int per = ring_rows_per_wg(nrows, gridDim.x);
  const int total = ring_rows_of_wg(nrows, gridDim.x, wg, sched_cyclic);
  const int2 *__restrict__ my = sched + static_cast<size_t>(wg) * per;
  if (threadIdx.x < RING_NL) ctl->landed[threadIdx.x] = 0;
  if (threadIdx.x < 16) ctl->crow[threadIdx.x] = static_cast<int>(threadIdx.x) < C ? static_cast<int>(threadIdx.x) : 0x7fffffff;
  if (threadIdx.x < 64) { ctl->acc[threadIdx.x >> 2][threadIdx.x & 3] = 0.0; }
  if (threadIdx.x >= 64 && threadIdx.x < 80) {
    const int ty = threadIdx.x - 64;
    const bool has = ty <= ff.nso;
    ctl->chi[ty] = has ? ff.atom[ty].chi : 0.0; ctl->eta[ty] = has ? ff.atom[ty].eta : 0.0; ctl->Zpq[ty] = (PQ && has) ? ff.Zpq[ty] : 0.0;
  }
  __syncthreads();
  if (wv < RING_NL) {
    // ---------------- loaders ----------------
    // A lone wave issues an instruction every 8-10 cycles at best (branches cost a refill), and the budget is ~500 cycles per row: the loop
    // is straight-line.  Every row takes exactly KH + KI (+ KH) DMA instructions whatever its length -- instruction k of a stream runs
    // with EXEC = the granules the row has left for it, possibly none (an instruction with EXEC = 0 still counts in vmcnt and retires at
    // once) -- so the row whose data has landed follows from the instruction count alone: after s_waitcnt vmcnt(RING_V) all rows but the
    // youngest RING_V / K are in LDS.  Ring space is tracked in VIRTUAL (never wrapping) positions: a row occupies [vs, vs + n4), the tail
    // skipped at a wrap counts as occupied, and a row fits while vs + n4 <= vstart(oldest row in use) + R.
    __builtin_amdgcn_s_setprio(3);
    // Two loader waves (on different SIMDs) run the same bookkeeping and take alternate instructions of every row; each publishes the rows
    // ITS instructions have landed, a consumer waits for both.
    constexpr bool PQS = PQ && (MODE == MODE_GRAD || STORE);
    constexpr int KS = (PQS ? 2 * KH + KH / 2 : KH + KH / 2) / RING_NL;   // DMA instructions per row and loader
    constexpr int LAG = (RING_V + KS - 1) / KS;             // rows behind the newest issue that s_waitcnt vmcnt(LAG * KS) guarantees
    static_assert(((PQS ? 2 * KH + KH / 2 : KH + KH / 2) % RING_NL) == 0, "instructions per row must split evenly over the loaders");
    const int role = wv;
    const unsigned hbase = lds_offset(hring), cbase = lds_offset(cring), ibase = lds_offset(iring);
    int cur = 0, vcur = 0, t = 0, tail = 0, vlimit = R;
    int h_vs = 0;                                   // per-lane history: lane (t & 63) holds the virtual start of row t
    int2 nxt = make_int2(0, 0);
    if (total > 0) nxt = my[0];
    for (int i = 0; i < total; ++i) {
      const int row = __builtin_amdgcn_readfirstlane(nxt.x), n = __builtin_amdgcn_readfirstlane(nxt.y);
      if (i + 1 < total) nxt = my[i + 1];           // contiguous per workgroup: scalar loads that stay in the scalar cache
      const int n4 = max(4, (n + 3) & ~3);          // an empty row still owns a granule: ring positions stay distinct
      if (cur + n4 > R) { vcur += R - cur; cur = 0; }
      const int rp = cur, vs = vcur;
      if (vs + n4 > vlimit || t - tail >= 64) {     // slow path: find out how far the consumers have come
        bool drained = false;
        for (;;) {
          const int cr = lane < C ? flag_load(&ctl->crow[lane]) : 0x7fffffff;
          tail = __builtin_amdgcn_readfirstlane(min(t, wave_min_int(cr)));
          vlimit = (tail < t ? __builtin_amdgcn_readlane(h_vs, tail & 63) : vs) + R;
          if (vs + n4 <= vlimit && t - tail < 64) break;
          if (!drained) {                           // nothing a consumer waits for may depend on further issues
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) flag_store(&ctl->landed[role], t);
            drained = true;
          }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      if (lane == 0 && role == 0) *reinterpret_cast<int4 *>(&ctl->tab[(t & 63) * 4]) = make_int4(row, rp, n, 0);
      if (lane == (t & 63)) h_vs = vs;
      if (probe != 3) {
        const size_t base = static_cast<size_t>(row) * S10;
        const int gh = n > 0 ? n4 >> 1 : 0, gi = n > 0 ? n4 >> 2 : 0;      // 16-byte granules of the value / entry streams
        const unsigned voff = static_cast<unsigned>(lane) * 16u;
        // value stream: KH instructions of 128 entries; (PQEq) the same for the shell-core values; entry words: KH / 2 instructions of 256
        static_assert(RING_NL == 1, "the grouped issue is written for one loader per workgroup");
#pragma unroll
        for (int g = 0; g < KH; g += 4) glds16_nt_group<4>(voff, hess + base + 128 * g, hbase + static_cast<unsigned>(rp + 128 * g) * 8u, gh - 64 * g);
        if (PQS) {
#pragma unroll
          for (int g = 0; g < KH; g += 4) glds16_nt_group<4>(voff, hsc + base + 128 * g, cbase + static_cast<unsigned>(rp + 128 * g) * 8u, gh - 64 * g);
        }
#pragma unroll
        for (int g = 0; g < KH / 2; g += 4) glds16_nt_group<(KH / 2 >= 4 ? 4 : KH / 2)>(voff, nb10 + base + 256 * g, ibase + static_cast<unsigned>(rp + 256 * g) * 4u, gi - 64 * g);
      }
      cur += n4; vcur += n4; ++t;
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LAG * KS) : "memory");
      if (lane == 0) flag_store(&ctl->landed[role], max(0, t - LAG));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) flag_store(&ctl->landed[role], t);
  } else if (wv < RING_NL + C) {
    // ---------------- consumer ----------------
    const int c = wv - RING_NL;
    // The results of a row wait in LDS (ctl->res) and go to HBM only after the NEXT row's gathers are in flight: a store issued at the end
    // of its own row would have to complete before the next row may overwrite the registers it reads (s_waitcnt vmcnt(0) at the loop
    // head).  The four running sums of the consumer live in LDS as well (ctl->acc, ds_add_f64): the kernel has to fit 64 VGPRs.
    if (lane == 0) ctl->resrow[c] = -1;
    auto flush = [&]() {
      if (lane == 0) {
        const int st_row = ctl->resrow[c];
        if (st_row >= 0) {
          const double2 ra = make_double2(ctl->res[c][0], ctl->res[c][1]), rb = make_double2(ctl->res[c][2], ctl->res[c][3]);
          if (MODE == MODE_HSH) { if (STORE) { rs_all[st_row] = ra; rs_gh[st_row] = rb; } }
          else { gst[st_row] = ra; if (STORE) { rs_all[st_row] = rb; rs_gh[st_row] = make_double2(ctl->res[c][4], ctl->res[c][5]); } }
        }
      }
    };
    auto acc_add = [&](int k, double v) { __hip_atomic_fetch_add(&ctl->acc[c][k], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    for (int t = c; t < total; t += C) {
      for (;;) {
        int l = flag_load(&ctl->landed[0]);
#pragma unroll
        for (int r = 1; r < RING_NL; ++r) l = min(l, flag_load(&ctl->landed[r]));
        if (l > (grouped ? min(total - 1, (t / C) * C + C - 1) : t)) break;
        __builtin_amdgcn_s_sleep(1);
      }
      asm volatile("" ::: "memory");
      const int4 te = *reinterpret_cast<const int4 *>(&ctl->tab[(t & 63) * 4]);
      const int row = __builtin_amdgcn_readfirstlane(te.x), rp = __builtin_amdgcn_readfirstlane(te.y), n = __builtin_amdgcn_readfirstlane(te.z);
      const int n4 = (n + 3) & ~3;
      if (probe == 1) { if (lane == 0) flag_store(&ctl->crow[c], t + C); continue; }
      int rowv = row;                               // the row index as a VGPR: tail operands come by VECTOR loads (as scalar loads they would
      asm volatile("" : "+v"(rowv));                // share lgkmcnt with the LDS reads, and scalar loads return out of order)
      double as = 0.0, at = 0.0, gs_ = 0.0, gt_ = 0.0;
      int pf_t = 0;
      double2 pf_a = make_double2(0.0, 0.0), pf_b = make_double2(0.0, 0.0);
      double4 pf_pq = make_double4(0.0, 0.0, 0.0, 0.0);
      // One SECTION = up to 512 entries of the row (all of an RDX row).  Order: every LDS read of the section; the ring space is released as
      // soon as the last section's values sit in registers (the loader refills it while this row is still gathering); gathers in two groups of
      // four; the stores of the PREVIOUS row and the tail operands of this one are queued behind the first group -- vmcnt counts in order,
      // so nothing the arithmetic waits for stands behind an HBM round trip.
      for (int kb = 0; kb < n4 || kb == 0; kb += 512) {
        unsigned e[8];
        double h[8], cv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int k = kb + 64 * u + lane;
          const bool ok = k < n4;
          e[u] = ok ? static_cast<unsigned>(iring[rp + k]) : 0u;
          h[u] = ok ? hring[rp + k] : 0.0;
          if (PQ && (MODE == MODE_GRAD || STORE)) cv[u] = ok ? cring[rp + k] : 0.0;
        }
        const bool last = kb + 512 >= n4;
        if (last) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); if (lane == 0) flag_store(&ctl->crow[c], t + C); }
        auto group = [&](auto u0c) {
          constexpr int U0 = decltype(u0c)::value;
          double2 v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) v[u] = xv[probe == 2 ? (lane & 15) : (e[U0 + u] & NB10_IDX_MASK)];
          if (U0 == 0 && kb == 0) {
            asm volatile("" ::: "memory");
            flush();
            pf_t = type[rowv];
            pf_a = (MODE == MODE_HSH) ? hst[rowv] : qst[rowv];
            pf_b = (MODE == MODE_HSH) ? gst[rowv] : make_double2(q[rowv], 0.0);
            if (PQ && MODE == MODE_GRAD) pf_pq = pqrow[rowv];
            asm volatile("" ::: "memory");
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const double hh = h[U0 + u];
            as += hh * v[u].x;
            at += hh * v[u].y;
            if ((MODE == MODE_GRAD || STORE) && !PQ) { const double hg = (e[U0 + u] & NB10_GHOST) ? hh : 0.0; gs_ += hg * v[u].x; gt_ += hg * v[u].y; }
            if ((MODE == MODE_GRAD || STORE) && PQ) { gs_ += cv[U0 + u] * v[u].x; gt_ += cv[U0 + u] * v[u].y; }
          }
        };
        group(std::integral_constant<int, 0>{});
        if (kb + 256 < n4) group(std::integral_constant<int, 4>{});
      }
      const double mu = (MODE == MODE_GRAD) ? scal[S_MU] : 0.0;
      as = wave_sum(as); at = wave_sum(at);
      if (MODE == MODE_GRAD || STORE) { gs_ = wave_sum(gs_); gt_ = wave_sum(gt_); }
      const double chi_i = ctl->chi[pf_t], eta_i = ctl->eta[pf_t];
      if (lane == 0) {
        ctl->resrow[c] = row;
        if (MODE == MODE_HSH) {
          ctl->res[c][0] = as; ctl->res[c][1] = at; ctl->res[c][2] = gs_; ctl->res[c][3] = gt_;
          const double ts = eta_i * pf_a.x + as, tt = eta_i * pf_a.y + at;        // qeq.F90:294-302
          acc_add(0, ts * pf_a.x); acc_add(1, tt * pf_a.y);                       // hshs_sum, hsht_sum (:309-310)
          acc_add(2, pf_b.x * pf_a.x); acc_add(3, pf_b.y * pf_a.y);               // g.h (:119,123)
        } else {
          const double fpq = PQ ? pf_pq.x : 0.0;
          const double g1 = -chi_i - eta_i * pf_a.x - as - fpq;                   // qeq.F90:349-350 (pqeq.F90:466)
          const double g2 = -1.0 - eta_i * pf_a.y - at;
          ctl->res[c][0] = g1; ctl->res[c][1] = g2; ctl->res[c][2] = as; ctl->res[c][3] = at; ctl->res[c][4] = gs_; ctl->res[c][5] = gt_;
          acc_add(0, g1 * g1); acc_add(1, g2 * g2);                               // Gnew (:355-356)
          const double qi = pf_b.x;
          const double hq_all = as - mu * at, hq_res = (as - gs_) - mu * (at - gt_);
          if (PQ) { DevAtomP ap; ap.chi = chi_i; ap.eta = eta_i; acc_add(2, pq_est_row(ap, ctl->Zpq[pf_t], pf_pq, qi, hq_all, gs_ - mu * gt_)); }
          else acc_add(2, chi_i * qi + 0.5 * eta_i * qi * qi + 0.5 * qi * (hq_all + hq_res));  // Est (:297-306)
        }
      }
    }
    flush();
    if (lane == 0) flag_store(&ctl->crow[c], 0x7fffffff);
  }
  __syncthreads();
  if (threadIdx.x < 4) {                            // the workgroup's partial: its consumers in order (fixed summation order)
    double s = 0.0;
    for (int k = 0; k < C; ++k) s += ctl->acc[k][threadIdx.x];
    __hip_atomic_store(partials + (static_cast<size_t>(pbase) + wg) * 4 + threadIdx.x, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---- where does the row kernel's time go?  Stripped-down forms of k_spmv, timed in isolation (debug tap 102; experiments only) -------
//   LEVEL 0: the two streams of a row only (4 x 64 entries per trip as k_spmv), one sum, one wavefront reduction, no store
//   LEVEL 1: + the 16-byte gather per entry and the two FMAs
//   LEVEL 2: + the ghost-column sums and all four reductions
//   LEVEL 3: + the tail operands (type, hst, gst of the row)   LEVEL 4: + the two 16-byte row stores
//   LEVEL 5: + the per-workgroup partials with their barrier (= the work of k_spmv<HSH, STORE>)   LEVEL 6: as 5 with ONE 32-byte row store
template <int LEVEL>
__global__ void __launch_bounds__(1024) k_spmv_bisect(int N, int S10, const int *__restrict__ nb10, const double *__restrict__ hess, const int *__restrict__ n10,
                                                      const double2 *__restrict__ xv, const double2 *__restrict__ hst, const double2 *__restrict__ gst,
                                                      const int *__restrict__ type, double *__restrict__ partials, double2 *__restrict__ rs_all, double2 *__restrict__ rs_gh, double *__restrict__ sink) {
  const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
  const int row = xcd_swizzle(blockIdx.x, gridDim.x) * wpb + __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  if (row < N) {
    const size_t base = static_cast<size_t>(row) * S10;
    unsigned e[UNR];
    double h[UNR];
    auto request = [&](int kb, int bound) {
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int k = kb + lane + 64 * u;
        const bool ok = k < bound;
        e[u] = ok ? static_cast<unsigned>(__builtin_nontemporal_load(nb10 + base + k)) : 0u;
        h[u] = ok ? __builtin_nontemporal_load(hess + base + k) : 0.0;
      }
    };
    request(0, S10);
    const int n = n10[row] & N10_COUNT;
    int pf_t = 0; double2 pf_a = make_double2(0, 0), pf_b = make_double2(0, 0);
    if (LEVEL >= 3) { pf_t = type[row]; pf_a = hst[row]; pf_b = gst[row]; }
#pragma unroll
    for (int u = 0; u < UNR; ++u) { const bool ok = lane + 64 * u < n; e[u] = ok ? e[u] : 0u; h[u] = ok ? h[u] : 0.0; }
    double as = 0.0, at = 0.0, gs_ = 0.0, gt_ = 0.0;
    for (int kb = 0; kb < n; kb += 64 * UNR) {
      if (kb > 0) request(kb, n);
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        if (LEVEL == 0) { as += h[u] * static_cast<double>(e[u] & 255u); }
        else {
          const double2 v = xv[e[u] & NB10_IDX_MASK];
          as += h[u] * v.x; at += h[u] * v.y;
          if (LEVEL >= 2) { const double hg = (e[u] & NB10_GHOST) ? h[u] : 0.0; gs_ += hg * v.x; gt_ += hg * v.y; }
        }
      }
    }
    as = wave_sum(as);
    if (LEVEL >= 1) at = wave_sum(at);
    if (LEVEL >= 2) { gs_ = wave_sum(gs_); gt_ = wave_sum(gt_); }
    if (lane == 0) {
      if (LEVEL >= 3) {
        if (LEVEL == 4 || LEVEL == 5) { rs_all[row] = make_double2(as, at); rs_gh[row] = make_double2(gs_, gt_); }
        if (LEVEL == 6) reinterpret_cast<double4 *>(sink)[row + 1] = make_double4(as, at, gs_, gt_);      // (sink: a 32-byte-per-row buffer for this level)
        acc[0] = (0.5 * pf_a.x + as) * pf_a.x; acc[1] = (0.5 * pf_a.y + at) * pf_a.y; acc[2] = pf_b.x * pf_a.x + pf_t; acc[3] = pf_b.y * pf_a.y;
        if (LEVEL == 3 && acc[0] + acc[1] + acc[2] + acc[3] == 1.2345e-300) sink[0] = as;
      } else if (as + at + gs_ + gt_ == 1.2345e-300) sink[0] = as;       // keeps the sums alive
    }
  }
  if (LEVEL >= 5) block_store_partials<4>(acc, partials, 4);
}
void spmv_bisect_ms(Engine &e, double *out4) {
  const int rb = nblk(e.N, 16);
  double *buf32 = nullptr;
  if (hipMalloc(reinterpret_cast<void **>(&buf32), sizeof(double) * 4 * (static_cast<size_t>(e.N) + 2)) != hipSuccess) return;
  auto run = [&](auto lv) {
    constexpr int L = decltype(lv)::value;
    for (int r = 0; r < 11; ++r) {
      if (r == 1) hipEventRecord(e.ev[2], e.stream);
      k_spmv_bisect<L><<<rb, 1024, 0, e.stream>>>(e.N, e.S10, e.nb10, e.hess, e.n10, e.xs, e.hst, e.gst, e.type, e.partials, e.wall, e.wgh, buf32);
    }
    hipEventRecord(e.ev[3], e.stream); hipEventSynchronize(e.ev[3]);
    float ms = 0; hipEventElapsedTime(&ms, e.ev[2], e.ev[3]);
    return static_cast<double>(ms) / 10.0;
  };
  out4[0] = run(std::integral_constant<int, 0>{}); out4[1] = run(std::integral_constant<int, 1>{});
  out4[2] = run(std::integral_constant<int, 2>{}); out4[3] = run(std::integral_constant<int, 3>{});
  out4[4] = run(std::integral_constant<int, 4>{}); out4[5] = run(std::integral_constant<int, 5>{}); out4[6] = run(std::integral_constant<int, 6>{});
  (void)hipFree(buf32);
}

// ---- window pass, timing probe (debug tap 103; experiments only) -----------------------------------------------------------------------
// What would a pass cost that (a) holds the vector entries of a 16-row group's common partner window in LDS (staged with coalesced loads,
// read back conflict-free) instead of gathering 16 bytes per entry, (b) replaces the 4-byte entry stream by a bit per window slot and row
// (value k of the row belongs to the k-th set bit), (c) has two dependent global round trips per wavefront (bit words -> values) instead of
// four?  The bit words here are SYNTHETIC (hashed, ~31 % set like RDX: 420 of 1,344 slots) and the windows arbitrary runs of the sorted
// vector, so the sums mean nothing; bytes, instruction mix, LDS traffic and the workgroup structure are those of the real thing.
constexpr int WIN_NW = 21, WIN_NWS = 24, WIN_RUNS = 25, WIN_RUNLEN = 54;     // 21 words of 64 slots; 25 runs of 54 slots = 1,350 >= 1,344
__global__ void k_winprobe_setup(int N, unsigned long long *__restrict__ bm) {
  const size_t t = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= static_cast<size_t>(N) * WIN_NWS) return;
  auto mix = [](unsigned long long z) { z += 0x9e3779b97f4a7c15ULL; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; return z ^ (z >> 31); };
  const unsigned long long h1 = mix(4 * t), h2 = mix(4 * t + 1), h3 = mix(4 * t + 2), h4 = mix(4 * t + 3);
  bm[t] = (t % WIN_NWS) < WIN_NW ? ((h1 & h2) | (h1 & h3 & h4)) : 0ULL;
}
template <int VARIANT, bool NT>
__global__ void __launch_bounds__(1024) k_spmv_winprobe(int N, int G, int S10, const double *__restrict__ hess, const unsigned long long *__restrict__ bm,
                                                        const double2 *__restrict__ xv, const double2 *__restrict__ hst, const double2 *__restrict__ gst,
                                                        const int *__restrict__ type, DevFF ff, double *__restrict__ partials, double2 *__restrict__ rs_all, double2 *__restrict__ rs_gh) {
  __shared__ double2 s_x[WIN_NW * 64 + 64];
  __shared__ unsigned long long s_gm[WIN_NWS];
  __shared__ double s_row[16][4];
  const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int grp = xcd_swizzle(blockIdx.x, gridDim.x);
  const int row = grp * wpb + wave;
  const bool live = row < N;
  const size_t base = static_cast<size_t>(live ? row : 0) * S10;
  // bit words of the row: lane w holds word w
  unsigned long long m = 0ULL;
  if (live && lane < WIN_NW) m = __builtin_nontemporal_load(bm + static_cast<size_t>(row) * WIN_NWS + lane);
  // the group's window: run t by wavefront t % 16 (coalesced 16-byte loads of the sorted vector)
  for (int t = wave; t < WIN_RUNS; t += wpb) {
    const int k0 = static_cast<int>((static_cast<long long>(grp) * 16 + static_cast<long long>(t) * 509) % (G - 64));
    if (lane < WIN_RUNLEN && t * WIN_RUNLEN + lane < WIN_NW * 64 + 64) s_x[t * WIN_RUNLEN + lane] = xv[k0 + lane];
  }
  if (threadIdx.x < WIN_NWS) s_gm[threadIdx.x] = (threadIdx.x & 3) == 0 ? 0x00ff00ff00ff00ffULL : 0ULL;      // which slots are ghosts
  // values: the k-th set bit of the row names value k of the row's slot -- all words requested before the first is used
  double v[WIN_NW];
  int pw = 0;
  const unsigned bit_lo = lane < 32 ? (1u << lane) : 0u, bit_hi = lane < 32 ? 0u : (1u << (lane - 32));
#pragma unroll
  for (int w = 0; w < WIN_NW; ++w) {
    const unsigned lo = __builtin_amdgcn_readlane(static_cast<unsigned>(m), w), hi = __builtin_amdgcn_readlane(static_cast<unsigned>(m >> 32), w);
    const unsigned long long mw = (static_cast<unsigned long long>(hi) << 32) | lo;
    const int idx = pw + __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0));
    const bool on = ((lo & bit_lo) | (hi & bit_hi)) != 0u;
    v[w] = on ? (NT ? __builtin_nontemporal_load(hess + base + idx) : hess[base + idx]) : 0.0;
    pw += __popcll(mw);
  }
  __syncthreads();
  unsigned long long gmv = lane < WIN_NWS ? s_gm[lane] : 0ULL;
  double as = 0.0, at = 0.0, gs_ = 0.0, gt_ = 0.0;
#pragma unroll
  for (int w = 0; w < WIN_NW; ++w) {
    const double2 x = s_x[64 * w + lane];
    as += v[w] * x.x; at += v[w] * x.y;
    if (VARIANT >= 1) {
      const unsigned glo = __builtin_amdgcn_readlane(static_cast<unsigned>(gmv), w), ghi = __builtin_amdgcn_readlane(static_cast<unsigned>(gmv >> 32), w);
      const double hg = ((glo & bit_lo) | (ghi & bit_hi)) != 0u ? v[w] : 0.0;
      gs_ += hg * x.x; gt_ += hg * x.y;
    }
  }
  as = wave_sum(as); at = wave_sum(at); gs_ = wave_sum(gs_); gt_ = wave_sum(gt_);
  if (lane == 0) { s_row[wave][0] = as; s_row[wave][1] = at; s_row[wave][2] = gs_; s_row[wave][3] = gt_; }
  __syncthreads();
  if (wave != 0) return;
  const int r = (lane < wpb && grp * wpb + lane < N) ? grp * wpb + lane : N;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  if (r < N) {
    const double ras = s_row[lane][0], rat = s_row[lane][1];
    const int t = type[r];
    const double2 pa = hst[r], pb = gst[r];
    const DevAtomP ap = ff.atom[t];
    rs_all[r] = make_double2(ras, rat); rs_gh[r] = make_double2(s_row[lane][2], s_row[lane][3]);
    const double ts = ap.eta * pa.x + ras, tt = ap.eta * pa.y + rat;
    a0 = ts * pa.x; a1 = tt * pa.y; a2 = pb.x * pa.x; a3 = pb.y * pa.y;
  }
  a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2); a3 = wave_sum(a3);
  if (lane < 4) {
    const double val = lane == 0 ? a0 : (lane == 1 ? a1 : (lane == 2 ? a2 : a3));
    __hip_atomic_store(partials + static_cast<size_t>(blockIdx.x) * 4 + lane, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// the same with a 16-bit window slot per entry (bit 15: ghost column) next to the value: the row kernel's streams, two entries per lane and
// request (values 16 bytes, slots 4 bytes per lane), its gathers replaced by LDS reads.  10 instead of 12 bytes per entry.
__global__ void k_win16_setup(int N, int S10, unsigned short *__restrict__ sl) {
  const size_t t = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= static_cast<size_t>(N) * S10) return;
  const int e = static_cast<int>(t % S10);
  unsigned long long z = t + 0x9e3779b97f4a7c15ULL; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z ^= z >> 27;
  sl[t] = static_cast<unsigned short>(((e * 3 + static_cast<int>(z & 1)) % (WIN_NW * 64)) | ((z & 0xf0) == 0 ? 0x8000 : 0));
}
template <int STEPS, int IND>        // IND bit 0: rows through rows_sorted; bit 1: the real windows through win_k / win_cnt
__global__ void __launch_bounds__(1024) k_spmv_win16probe(int N, int G, int S10, const double *__restrict__ hess, const unsigned short *__restrict__ sl, const int *__restrict__ n10,
                                                          const int *__restrict__ rows_sorted, const int *__restrict__ win_k, const int *__restrict__ win_cnt,
                                                          const double2 *__restrict__ xv, const double2 *__restrict__ hst, const double2 *__restrict__ gst,
                                                          const int *__restrict__ type, DevFF ff, double *__restrict__ partials, double2 *__restrict__ rs_all, double2 *__restrict__ rs_gh) {
  __shared__ double2 s_x[(IND & 2) ? WIN_MAXUNITS * WIN_UNIT : WIN_NW * 64 + 64];
  __shared__ double s_row[16][4];
  const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int grp = xcd_swizzle(blockIdx.x, gridDim.x);
  const int ridx = grp * wpb + wave;
  const int row = (IND & 1) ? (ridx < N ? rows_sorted[ridx] : N) : ridx;
  const bool live = row < N;
  const size_t base = static_cast<size_t>(live ? row : 0) * S10;
  typedef double d2v __attribute__((ext_vector_type(2)));
  const d2v *hv2 = reinterpret_cast<const d2v *>(hess + base);
  const unsigned *sl2 = reinterpret_cast<const unsigned *>(sl + base);
  double2 v[STEPS]; unsigned ss[STEPS];
  auto request = [&](int kb, int bound) {          // entries kb + 128 u + 2 lane, + 1
#pragma unroll
    for (int u = 0; u < STEPS; ++u) {
      const int k = kb + 128 * u + 2 * lane;
      const bool ok = k < bound;
      if (ok) { const d2v t2 = __builtin_nontemporal_load(hv2 + (k >> 1)); v[u] = make_double2(t2.x, t2.y); } else v[u] = make_double2(0.0, 0.0);
      ss[u] = ok ? __builtin_nontemporal_load(sl2 + (k >> 1)) : 0u;
    }
  };
  request(0, live ? S10 : 0);
  const int n = live ? (n10[row] & N10_COUNT) : 0;
  if (IND & 2) {
    const int nslots = WIN_UNIT * win_cnt[grp];
    const int *wk = win_k + static_cast<size_t>(grp) * WIN_MAXUNITS;
    for (int t = threadIdx.x; t < nslots; t += blockDim.x) s_x[t] = xv[min(wk[t / WIN_UNIT] + (t & (WIN_UNIT - 1)), G - 1)];
  } else
  for (int t = wave; t < WIN_RUNS; t += wpb) {
    const int k0 = static_cast<int>((static_cast<long long>(grp) * 16 + static_cast<long long>(t) * 509) % (G - 64));
    if (lane < WIN_RUNLEN && t * WIN_RUNLEN + lane < WIN_NW * 64 + 64) s_x[t * WIN_RUNLEN + lane] = xv[k0 + lane];
  }
#pragma unroll
  for (int u = 0; u < STEPS; ++u) {                // entries behind the row's end get weight 0
    const int k = 128 * u + 2 * lane;
    v[u].x = k < n ? v[u].x : 0.0; v[u].y = k + 1 < n ? v[u].y : 0.0;
  }
  __syncthreads();
  double as = 0.0, at = 0.0, gs_ = 0.0, gt_ = 0.0;
  for (int kb = 0; kb < n; kb += 128 * STEPS) {
    if (kb > 0) request(kb, n);
#pragma unroll
    for (int u = 0; u < STEPS; ++u) {
      const double2 x0 = s_x[ss[u] & 0x7fffu], x1 = s_x[(ss[u] >> 16) & 0x7fffu];
      as += v[u].x * x0.x; at += v[u].x * x0.y; as += v[u].y * x1.x; at += v[u].y * x1.y;
      const double g0 = (ss[u] & 0x8000u) ? v[u].x : 0.0, g1 = (ss[u] & 0x80000000u) ? v[u].y : 0.0;
      gs_ += g0 * x0.x; gt_ += g0 * x0.y; gs_ += g1 * x1.x; gt_ += g1 * x1.y;
    }
  }
  as = wave_sum(as); at = wave_sum(at); gs_ = wave_sum(gs_); gt_ = wave_sum(gt_);
  if (lane == 0) { s_row[wave][0] = as; s_row[wave][1] = at; s_row[wave][2] = gs_; s_row[wave][3] = gt_; }
  __syncthreads();
  if (wave != 0) return;
  const int r = (lane < wpb && grp * wpb + lane < N) ? grp * wpb + lane : N;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  if (r < N) {
    const double ras = s_row[lane][0], rat = s_row[lane][1];
    const int t = type[r];
    const double2 pa = hst[r], pb = gst[r];
    const DevAtomP ap = ff.atom[t];
    rs_all[r] = make_double2(ras, rat); rs_gh[r] = make_double2(s_row[lane][2], s_row[lane][3]);
    const double ts = ap.eta * pa.x + ras, tt = ap.eta * pa.y + rat;
    a0 = ts * pa.x; a1 = tt * pa.y; a2 = pb.x * pa.x; a3 = pb.y * pa.y;
  }
  a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2); a3 = wave_sum(a3);
  if (lane < 4) {
    const double val = lane == 0 ? a0 : (lane == 1 ? a1 : (lane == 2 ? a2 : a3));
    __hip_atomic_store(partials + static_cast<size_t>(blockIdx.x) * 4 + lane, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
void spmv_winprobe_ms(Engine &e, double *out) {
  unsigned long long *bm = nullptr;
  unsigned short *sl = nullptr;
  for (int k = 0; k < 9; ++k) out[k] = -1.0;
  if (e.S10 < 600 || (e.S10 & 1) || hipMalloc(reinterpret_cast<void **>(&bm), sizeof(unsigned long long) * static_cast<size_t>(e.N) * WIN_NWS) != hipSuccess) return;
  if (hipMalloc(reinterpret_cast<void **>(&sl), sizeof(unsigned short) * static_cast<size_t>(e.N) * e.S10) != hipSuccess) { (void)hipFree(bm); return; }
  const size_t nt = static_cast<size_t>(e.N) * WIN_NWS, ns = static_cast<size_t>(e.N) * e.S10;
  k_winprobe_setup<<<static_cast<unsigned>((nt + 255) / 256), 256, 0, e.stream>>>(e.N, bm);
  k_win16_setup<<<static_cast<unsigned>((ns + 255) / 256), 256, 0, e.stream>>>(e.N, e.S10, sl);
  const int rb = nblk(e.N, 16);
  auto timed = [&](auto launch) {
    for (int r = 0; r < 11; ++r) {
      if (r == 1) hipEventRecord(e.ev[2], e.stream);
      launch();
    }
    hipEventRecord(e.ev[3], e.stream); hipEventSynchronize(e.ev[3]);
    float ms = 0; hipEventElapsedTime(&ms, e.ev[2], e.ev[3]);
    return static_cast<double>(ms) / 10.0;
  };
#define RX_WP(V, NT) timed([&] { k_spmv_winprobe<V, NT><<<rb, 1024, 0, e.stream>>>(e.N, e.G, e.S10, e.hess, bm, e.xs, e.hst, e.gst, e.type, e.dff, e.partials, e.wall, e.wgh); })
  out[0] = RX_WP(0, true); out[1] = RX_WP(1, true); out[2] = RX_WP(0, false); out[3] = RX_WP(1, false);
#undef RX_WP
#define RX_W16(I) timed([&] { k_spmv_win16probe<2, I><<<rb, 1024, 0, e.stream>>>(e.N, e.G, e.S10, e.hess, sl, e.n10, e.rows_sorted, e.win_k, e.win_cnt, e.xs, e.hst, e.gst, e.type, e.dff, e.partials, e.wall, e.wgh); })
  out[4] = RX_W16(0);
  if (e.win_valid) { out[5] = RX_W16(1); out[6] = RX_W16(2); out[7] = RX_W16(3); out[8] = e.win_maxunits; }
#undef RX_W16
  (void)hipFree(bm); (void)hipFree(sl);
}

// the real window pass and the real row pass back to back, ten launches each (debug tap 104; experiments only)
void spmv_isolated_ms(Engine &e, double *out) {
  out[0] = out[1] = -1.0;
  const int reps = std::getenv("RXMD_ISO_REPS") ? std::max(1, std::atoi(std::getenv("RXMD_ISO_REPS"))) : 10;
  auto timed = [&](auto launch) {
    for (int r = 0; r < reps + 1; ++r) {
      if (r == 1) hipEventRecord(e.ev[2], e.stream);
      launch();
    }
    hipEventRecord(e.ev[3], e.stream); hipEventSynchronize(e.ev[3]);
    float ms = 0; hipEventElapsedTime(&ms, e.ev[2], e.ev[3]);
    return static_cast<double>(ms) / reps;
  };
  if (e.ff.pqeq) return;
  if (e.win_valid) {
    const size_t lds = static_cast<size_t>(e.win_maxunits) * WIN_UNIT * sizeof(double2);
    out[0] = timed([&] { k_spmv_win<MODE_HSH, true, false><<<e.win_groups, 64 * WIN_ROWS, lds, e.stream>>>(e.N, e.G, e.S10, e.dff, e.sl10, e.hess, e.n10, e.rows_sorted, e.win_k, e.win_cnt, e.xs, e.hst, e.gst, e.qst, e.q, e.type, e.scal, e.partials, e.wall, e.wgh, e.hsc, e.pqrow, nullptr, e.win_groups, 0, nullptr); });
  }
  out[1] = timed([&] { k_spmv<MODE_HSH, true, false, 1><<<nblk(e.N, 16), 1024, 0, e.stream>>>(e.N, e.S10, e.dff, e.nb10, e.hess, e.n10, e.xs, e.hst, e.gst, e.qst, e.q, e.type, e.scal, e.partials, e.wall, e.wgh, e.hsc, e.pqrow, 1, nullptr, e.N, 0, nullptr); });
}

// timing probe of the ring pass in isolation (debug tap 101; experiments only): env RXMD_RING_PROBE / _R / _C / _WG as in Engine::qeq.
// Writes the scratch row sums wall / wgh and the partials only; returns the average launch time.
double ring_probe_ms(Engine &e, int reps) {
  auto geti = [](const char *k, int d) { const char *v = std::getenv(k); return v ? std::atoi(v) : d; };
  const int probe = geti("RXMD_RING_PROBE", 0), R = geti("RXMD_RING_R", 6144) & ~63, C = std::max(1, std::min(RING_MAXC, geti("RXMD_RING_C", RING_MAXC)));
  const int wgs = geti("RXMD_RING_WG", 2 * e.num_cu);
  const size_t lds = static_cast<size_t>(R) * 12 + sizeof(RingCtl);
  if (e.ff.pqeq || e.max_row10 > 512 || e.max_row10 > R || lds > 160 * 1024) return -1.0;
  hipFuncSetAttribute(reinterpret_cast<const void *>(&k_spmv_ring<MODE_HSH, true, false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int nwg = std::max(1, std::min(wgs, e.N / (2 * C)));
  const int cyclic = (geti("RXMD_RING_CYCLIC", 1) && nwg >= 8 && (nwg & 7) == 0) ? 1 : 0;
  if (!e.rsched[0]) hipMalloc(reinterpret_cast<void **>(&e.rsched[0]), sizeof(int2) * (static_cast<size_t>(e.rows10) + 2 * 4096 + 64));
  k_ring_schedule<<<nblk(nwg * ring_rows_per_wg(e.N, nwg), 256), 256, 0, e.stream>>>(e.N, nwg, cyclic, nullptr, e.n10, e.rsched[0]);
  e.rsched_valid[0] = false;                      // the next real pass rebuilds its own
  for (int r = 0; r < reps + 1; ++r) {
    if (r == 1) hipEventRecord(e.ev[2], e.stream);
    k_spmv_ring<MODE_HSH, true, false, 4><<<nwg, 64 * (C + RING_NL), lds, e.stream>>>(e.N, e.S10, e.dff, e.nb10, e.hess, e.n10, e.xs, e.hst, e.gst, e.qst, e.q, e.type, e.scal, e.partials, e.wall, e.wgh,
                                                                           e.hsc, e.pqrow, 1 | (probe << 8) | (geti("RXMD_RING_GROUP", 0) << 16), e.rsched[0], cyclic, e.N, 0, R, C);
  }
  hipEventRecord(e.ev[3], e.stream);
  hipEventSynchronize(e.ev[3]);
  float ms = 0;
  hipEventElapsedTime(&ms, e.ev[2], e.ev[3]);
  return ms / reps;
}
// bandwidth probe (debug tap 100): plain 16-byte-per-lane grid-stride read of the matrix value array; gives the
// read ceiling of the box the roofline fraction is quoted next to
__global__ void __launch_bounds__(256) k_stream_probe(size_t n16, const f64x2 *__restrict__ a, double *__restrict__ out) {
  double s = 0.0;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n16; i += static_cast<size_t>(gridDim.x) * blockDim.x) {
    const f64x2 v = __builtin_nontemporal_load(a + i);
    s += v.x + v.y;
  }
  if (s == 12345.678) out[0] = s;
}
double stream_probe_ms(Engine &e, int blocks) {
  const size_t n16 = static_cast<size_t>(e.rows10) * e.S10 / 2;
  hipEventRecord(e.ev[2], e.stream);
  for (int r = 0; r < 5; ++r) k_stream_probe<<<blocks, 256, 0, e.stream>>>(n16, reinterpret_cast<const f64x2 *>(e.hess), e.partials);
  hipEventRecord(e.ev[3], e.stream);
  hipEventSynchronize(e.ev[3]);
  float ms = 0;
  hipEventElapsedTime(&ms, e.ev[2], e.ev[3]);
  return ms / 5.0;
}

// single-block deterministic reduction of the per-block partials + the scalar algebra between passes
__device__ inline double block_sum_256(double v, double *sm) {
  sm[threadIdx.x] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
    __syncthreads();
  }
  const double r = sm[0];
  __syncthreads();
  return r;
}

// the scalar algebra between the passes, on the (all-reduced, MPI_ALLREDUCE qeq.F90:107,129,144,357) sums:
// stage 1: after the HSH pass -> REAL(4) line-minimisation factors (qeq.F90:133)
// stage 2: after the q update  -> mu = ssum/tsum (qeq.F90:147)
// stage 3: after the GRAD pass -> Gold<-Gnew, Gnew, Est, beta (qeq.F90:156-161)
// stage 4 (qeq_mode 1, fused loop): sums (qs, qt, gs.gs, gt.gt) -> mu, Gold<-Gnew, Gnew, beta;  stage 5: Est
// stage 6 (qeq_mode 1, multi-rank): stage 4 and Est in one -- Est is a quadratic in mu whose three coefficients are sums the
//          update kernel can form before mu exists (k_cg_update<true>), so the iteration needs two all-reduces instead of three
__device__ inline void scalar_algebra(int stage, double *__restrict__ scal) {
  const int par = stage >> 4;                     // run-ahead CG loop: which of the two stop flags this iteration's decision goes to (the NEXT iteration's parity)
  stage &= 15;
  const double r[4] = {scal[S_RAW0], scal[S_RAW1], scal[S_RAW2], scal[S_RAW3]};
  if (stage == 1) {
    scal[S_HSH_S] = r[0]; scal[S_HSH_T] = r[1]; scal[S_GH_S] = r[2]; scal[S_GH_T] = r[3];
    const float l1 = static_cast<float>(r[2] / r[0]), l2 = static_cast<float>(r[3] / r[1]);   // real(4) :: lmin(2)
    scal[S_LMIN_S] = static_cast<double>(l1); scal[S_LMIN_T] = static_cast<double>(l2);
  } else if (stage == 2) {
    scal[S_SSUM] = r[0]; scal[S_TSUM] = r[1];
    scal[S_MU] = r[0] / r[1];
  } else if (stage == 3) {
    const double go_s = scal[S_GNEW_S], go_t = scal[S_GNEW_T];
    scal[S_GOLD_S] = go_s; scal[S_GOLD_T] = go_t;
    scal[S_GNEW_S] = r[0]; scal[S_GNEW_T] = r[1]; scal[S_EST] = r[2];
    scal[S_BETA_S] = r[0] / go_s; scal[S_BETA_T] = r[1] / go_t;
  } else if (stage == 4 || stage == 6) {
    scal[S_SSUM] = r[0]; scal[S_TSUM] = r[1];
    const double mu = r[0] / r[1];
    scal[S_MU] = mu;
    const double go_s = scal[S_GNEW_S], go_t = scal[S_GNEW_T];
    scal[S_GOLD_S] = go_s; scal[S_GOLD_T] = go_t;
    scal[S_GNEW_S] = r[2]; scal[S_GNEW_T] = r[3];
    scal[S_BETA_S] = r[2] / go_s; scal[S_BETA_T] = r[3] / go_t;
    if (stage == 6) {
      const double prev = scal[S_EST], est = scal[S_RAW4] - mu * scal[S_RAW5] + mu * mu * scal[S_RAW6];
      scal[S_EST] = est;
      // the exit test the NEXT iteration starts with (qeq.F90:114-115), decided here where Est becomes final: the run-ahead CG loop has
      // that iteration's kernels queued already, they return at once when the flag is set; the host reads the same flag, it does not re-evaluate
      const double tol = scal[S_TOL];
      const bool stop = (0.5 * (fabs(prev) + fabs(est)) < tol) || (fabs(prev) > 0.0 && fabs(est / prev - 1.0) < tol);
      scal[S_STOP + par] = stop ? 1.0 : 0.0;      // two flags, by iteration parity: the kernels of iteration k read flag k & 1, which only update(k - 1) writes
    }
  } else {
    scal[S_EST] = r[0];
  }
}
__global__ void k_scalar_algebra(int stage, double *__restrict__ scal) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  scalar_algebra(stage, scal);
}

// In-kernel tail of a deterministic reduction: every workgroup has stored its four partial sums; the LAST one to arrive
// (arrival counter) adds all of them in a fixed order (thread t takes partials t, t+256, ...; then the fixed LDS tree), writes
// scal[S_RAW0..3] and, single rank only, runs the scalar algebra of `stage` -- so a reduction costs no extra launch.
// The result does not depend on which workgroup happens to be last.
__device__ inline void block_finish(int nblocks, double *partials, unsigned *ticket, int stage, double *scal, int nsets = 1) {
  __shared__ double smf[256];
  __shared__ bool last;
  // The partials were stored with device-scope atomics (write-through); waiting for those stores to complete is all the
  // release this needs.  A full __threadfence() here would write back the L2's dirty lines of the whole kernel from every
  // workgroup (measured: +0.25 ms per launch on the 160 MB vector kernels).
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  if (threadIdx.x == 0) last = (__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == static_cast<unsigned>(nblocks - 1));
  __syncthreads();
  if (!last) return;
  for (int set = 0; set < nsets; ++set) {            // set k: nblocks x 4 partials behind those of set k-1 -> scal[S_RAW0 + 4k ..]
    const double *ps = partials + static_cast<size_t>(set) * nblocks * 4;
    double a[4] = {0, 0, 0, 0};
    for (int b = threadIdx.x; b < nblocks; b += 256)
      for (int c = 0; c < 4; ++c) a[c] += __hip_atomic_load(ps + static_cast<size_t>(b) * 4 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int c = 0; c < 4; ++c) {
      const double r = block_sum_256(a[c], smf);
      if (threadIdx.x == 0) scal[S_RAW0 + 4 * set + c] = r;
    }
  }
  if (threadIdx.x == 0) {
    *ticket = 0u;                                    // ready for the next launch (stream order)
    if (stage > 0) scalar_algebra(stage, scal);
  }
}

// the matrix pass leaves one partial per workgroup (245k at 979,776 rows): 128 workgroups sum contiguous chunks,
// the last of them finishes (one launch for level-1 sums, final sum and scalar algebra)
__global__ void __launch_bounds__(256) k_reduce_fused(int nblocks, const double *__restrict__ partials, double *__restrict__ lvl1, unsigned *ticket, int stage, double *__restrict__ scal, const double *__restrict__ stopflag = nullptr) {
  if (stopflag && *stopflag != 0.0) return;
  __shared__ double sm[256];
  const int per = (nblocks + gridDim.x - 1) / gridDim.x;
  const int b0 = blockIdx.x * per, b1 = min(nblocks, b0 + per);
  double a[4] = {0, 0, 0, 0};
  for (int b = b0 + threadIdx.x; b < b1; b += 256)
    for (int c = 0; c < 4; ++c) a[c] += partials[static_cast<size_t>(b) * 4 + c];
  for (int c = 0; c < 4; ++c) {
    const double r = block_sum_256(a[c], sm);
    if (threadIdx.x == 0) __hip_atomic_store(lvl1 + blockIdx.x * 4 + c, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  block_finish(gridDim.x, lvl1, ticket, stage, scal);
}

// qs += lmin1*hs ; qt += lmin2*ht ; partial sums of qs, qt (qeq.F90:136-141)
__global__ void __launch_bounds__(256) k_update_qst(int N, const double *__restrict__ scal, const double2 *__restrict__ hst, double2 *__restrict__ qst, double *__restrict__ partials) {
  const double l1 = scal[S_LMIN_S], l2 = scal[S_LMIN_T];
  double acc[4] = {0, 0, 0, 0};
  double s = 0.0, t = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
    double2 qv = qst[i];
    const double2 hv = hst[i];
    qv.x = qv.x + l1 * hv.x; qv.y = qv.y + l2 * hv.y;
    qst[i] = qv;
    s += qv.x; t += qv.y;
  }
  s = wave_sum(s); t = wave_sum(t);
  acc[0] = s; acc[1] = t;
  block_store_partials<4>(acc, partials, 4);
}
// ---- qeq_mode 1: the whole vector algebra of one CG iteration in two launches -----------------------------------
// A: qs,qt += lmin (hs,ht); stored row sums += lmin H.(hs,ht); new gradient (no mu needed); sums (qs, qt, gs.gs, gt.gt)
//    -> tail: mu, Gnew, beta                                                    (qeq.F90:136-147,349-356,160-161)
//    EST3: also the three coefficients of Est(mu) = E0 - mu E1 + mu^2 E2 (the per-row term of k_cg_direction expanded in mu,
//    qeq.F90:297-306 with q = qs - mu qt) as a second set of partials -> scal[S_RAW4..6]
template <bool EST3>
__global__ void __launch_bounds__(256) k_cg_update(int N, DevFF ff, double *__restrict__ scal, const int *__restrict__ type, const double2 *__restrict__ hst, double2 *__restrict__ qst,
                                                    const double2 *__restrict__ wall, const double2 *__restrict__ wgh, double2 *__restrict__ sall, double2 *__restrict__ sgh,
                                                    double2 *__restrict__ gst, double *__restrict__ partials, unsigned *ticket, const double4 *__restrict__ pqrow, int stage, const double *__restrict__ stopflag) {
  if (stopflag && *stopflag != 0.0) return;
  const double l1 = scal[S_LMIN_S], l2 = scal[S_LMIN_T];
  double s = 0.0, t = 0.0, g1s = 0.0, g2s = 0.0, e0 = 0.0, e1 = 0.0, e2 = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
    double2 qv = qst[i];
    const double2 hv = hst[i];
    qv.x = qv.x + l1 * hv.x; qv.y = qv.y + l2 * hv.y;
    qst[i] = qv;
    double2 a = sall[i], g = sgh[i];
    const double2 wa = wall[i], wg = wgh[i];
    a.x += l1 * wa.x; a.y += l2 * wa.y; g.x += l1 * wg.x; g.y += l2 * wg.y;
    sall[i] = a; sgh[i] = g;
    const DevAtomP ap = ff.atom[type[i]];
    const double g1 = -ap.chi - ap.eta * qv.x - a.x - (pqrow ? pqrow[i].x : 0.0), g2 = -1.0 - ap.eta * qv.y - a.y;
    gst[i] = make_double2(g1, g2);
    s += qv.x; t += qv.y; g1s += g1 * g1; g2s += g2 * g2;
    if (EST3) {        // chi q + eta q^2/2 + q (Hq_all + Hq_res)/2 with q = s - mu t, Hq_all + Hq_res = A - mu B
      const double A = 2.0 * a.x - g.x, B = 2.0 * a.y - g.y;
      e0 += ap.chi * qv.x + 0.5 * ap.eta * qv.x * qv.x + 0.5 * qv.x * A;
      e1 += ap.chi * qv.y + ap.eta * qv.x * qv.y + 0.5 * (qv.x * B + qv.y * A);
      e2 += 0.5 * ap.eta * qv.y * qv.y + 0.5 * qv.y * B;
    }
  }
  double acc[4] = {wave_sum(s), wave_sum(t), wave_sum(g1s), wave_sum(g2s)};
  block_store_partials<4>(acc, partials, 4);
  if (EST3) {
    __syncthreads();                                           // block_store_partials stages through one LDS array
    double acc2[4] = {wave_sum(e0), wave_sum(e1), wave_sum(e2), 0.0};
    block_store_partials<4>(acc2, partials + static_cast<size_t>(gridDim.x) * 4, 4);
  }
  block_finish(gridDim.x, partials, ticket, stage, scal, EST3 ? 2 : 1);   // stage 4, or 0 = sums only (the all-reduce of a multi-rank run comes first)
}
// B: new direction h = g + beta h written to the other (hs,ht) buffer; q = qs - mu qt and the Est term (qeq.F90:150,160-164,297-306)
//    -> tail: Est (stage 5), sums only (0) or no reduction at all (-1: Est came with the update kernel's sums)
__global__ void __launch_bounds__(256) k_cg_direction(int N, DevFF ff, double *__restrict__ scal, const int *__restrict__ type,
                                                       const double2 *__restrict__ gst, const double2 *__restrict__ hst, double2 *__restrict__ hst_new,
                                                       const double2 *__restrict__ qst, const double2 *__restrict__ sall, const double2 *__restrict__ sgh, double *__restrict__ q,
                                                       double *__restrict__ partials, unsigned *ticket, const double4 *__restrict__ pqrow, int stage,
                                                       int G, const int *__restrict__ invpos, const int *__restrict__ groot, double2 *__restrict__ xs, const double *__restrict__ stopflag) {
  if (stopflag && *stopflag != 0.0) return;
  const double mu = scal[S_MU], b1 = scal[S_BETA_S], b2 = scal[S_BETA_T];
  double es = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
    const double2 g = gst[i], h = hst[i];
    const double2 hn = make_double2(g.x + b1 * h.x, g.y + b2 * h.y);
    hst_new[i] = hn;
    if (xs) xs[invpos[i]] = hn;                      // single rank: the cell-sorted gather copy of the next matrix pass (QCOPY2, qeq.F90:164) is written here ...
    const DevAtomP ap = ff.atom[type[i]];
    const double2 qv = qst[i], a = sall[i], gh = sgh[i];
    const double qi = qv.x - mu * qv.y;
    q[i] = qi;
    const double hq_all = a.x - mu * a.y, hq_res = (a.x - gh.x) - mu * (a.y - gh.y);
    if (pqrow) es += pq_est_row(ap, ff.Zpq[type[i]], pqrow[i], qi, hq_all, gh.x - mu * gh.y);
    else es += ap.chi * qi + 0.5 * ap.eta * qi * qi + 0.5 * qi * (hq_all + hq_res);
  }
  if (xs)                                            // ... including the periodic images: a ghost recomputes the value of its owner (two coalesced-by-owner reads)
    for (int t = N + blockIdx.x * blockDim.x + threadIdx.x; t < G; t += gridDim.x * blockDim.x) {
      const int r = groot[t];
      const double2 g = gst[r], h = hst[r];
      xs[invpos[t]] = make_double2(g.x + b1 * h.x, g.y + b2 * h.y);
    }
  if (stage < 0) return;
  double acc[4] = {wave_sum(es), 0.0, 0.0, 0.0};
  block_store_partials<4>(acc, partials, 4);
  block_finish(gridDim.x, partials, ticket, stage, scal);     // stage 5 or 0
}

// q = qs - mu*qt (qeq.F90:150)
__global__ void k_apply_q(int N, const double *__restrict__ scal, const double2 *__restrict__ qst, double *__restrict__ q) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const double2 v = qst[i];
  q[i] = v.x - scal[S_MU] * v.y;
}
// hs = gs + (Gnew1/Gold1)*hs ; ht likewise (qeq.F90:160-161); first = 1: hs = gs (qeq.F90:90-91)
__global__ void k_direction(int N, int first, const double *__restrict__ scal, const double2 *__restrict__ gst, double2 *__restrict__ hst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const double2 g = gst[i];
  if (first) { hst[i] = g; return; }
  const double2 h = hst[i];
  hst[i] = make_double2(g.x + scal[S_BETA_S] * h.x, g.y + scal[S_BETA_T] * h.y);
}
// initial vectors (qeq.F90:36-63)
__global__ void k_qeq_init(int N, int isQEq, double fqs, double *__restrict__ q, double *__restrict__ qsfp, double *__restrict__ qsfv, double2 *__restrict__ qst, double2 *__restrict__ hst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  hst[i] = make_double2(0.0, 0.0);
  if (isQEq == 1) { qsfp[i] = q[i]; qsfv[i] = 0.0; qst[i] = make_double2(q[i], 0.0); }
  else { qst[i] = make_double2(fqs * qsfp[i] + (1.0 - fqs) * q[i], 0.0); }
}

// gradient, Gnew and Est of the CG start vector from the row sums the list sweep left behind (qt = 0, mu = 0): the
// arithmetic of the MODE_GRAD epilogue without the matrix pass (qeq.F90:87, 349-356, 297-306)
__global__ void __launch_bounds__(256) k_grad_start(int N, DevFF ff, const int *__restrict__ type, const double2 *__restrict__ qst, const double *__restrict__ q,
                                                     const double2 *__restrict__ sall, const double2 *__restrict__ sgh, double2 *__restrict__ gst,
                                                     double *__restrict__ partials, const double4 *__restrict__ pqrow) {
  double g1s = 0.0, g2s = 0.0, es = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
    const int ti = type[i];
    const DevAtomP ap = ff.atom[ti];
    const double2 qv = qst[i], a = sall[i], g = sgh[i];
    const double g1 = -ap.chi - ap.eta * qv.x - a.x - (pqrow ? pqrow[i].x : 0.0), g2 = -1.0 - ap.eta * qv.y - a.y;
    gst[i] = make_double2(g1, g2);
    const double qi = q[i];
    g1s += g1 * g1; g2s += g2 * g2;
    if (pqrow) es += pq_est_row(ap, ff.Zpq[ti], pqrow[i], qi, a.x, g.x);
    else es += ap.chi * qi + 0.5 * ap.eta * qi * qi + 0.5 * qi * (a.x + (a.x - g.x));
  }
  double acc[4] = {wave_sum(g1s), wave_sum(g2s), wave_sum(es), 0.0};
  block_store_partials<4>(acc, partials, 4);
}

// the cell-sorted copy of a vector in two parts (multi-rank overlap): resident positions as soon as the vector exists, ghost
// positions when the halo has delivered them
__global__ void k_sorted_part(int G, int N, const int *__restrict__ perm, const double2 *__restrict__ v, double2 *__restrict__ xs, int ghosts) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= G) return;
  const int i = perm[k];
  if ((i >= N) == (ghosts != 0)) xs[k] = v[i];
}

void Engine::qeq_start_vectors() {
  k_qeq_init<<<nblk(N, 256), 256, 0, stream>>>(N, cfg.isQEq, cfg.Lex_fqs, q, qsfp, qsfv, qst, hst);
  sorted_copy(qst);                                                                             // QCOPY1, qeq.F90:86
}

void Engine::allreduce_scal4(int n) {
  if (nprocs == 1 && !nccl) return;
  const bool kt = kt_begin(&st.ms_allreduce, nullptr, &st.allreduce_calls);
  struct End { Engine *e; bool kt; ~End() { e->kt_end(kt); } } end_{this, kt};              // forced staged mode of a single rank without a communicator: nothing to add
  if (nccl) { rccl_allreduce_dev(scal + S_RAW0, n); return; }      // in stream order, no host round trip
  if (!has_comm || !comm.allreduce_sum) throw EngineError(RXMD_E_COMM, "vprocs > 1 needs a transport: call rxmd_hip_set_comm or rxmd_hip_comm_init_rccl first");
  RX_HIP(hipMemcpyAsync(h_scal + 48, scal + S_RAW0, sizeof(double) * n, hipMemcpyDeviceToHost, stream));
  sync_stream();
  if (comm.allreduce_sum(comm.ctx, h_scal + 48, n)) throw EngineError(RXMD_E_COMM, "allreduce callback failed");
  RX_HIP(hipMemcpyAsync(scal + S_RAW0, h_scal + 48, sizeof(double) * n, hipMemcpyHostToDevice, stream));
}

void Engine::qeq() {
  if (!atoms_set) throw EngineError(RXMD_E_STATE, "atoms were never set");
  if (cfg.isQEq != 1 && cfg.isQEq != 2) { nstep_qeq = 0; return; }   // qeq.F90:60-61
  tic(6);
  // the list sweep of this step can form the row sums of the start vector on the way (saves the matrix pass of qeq.F90:87)
  static const bool prepass_on = (std::getenv("RXMD_QEQ_NO_PREPASS") == nullptr);
  sums_from_list = false;
  if (!lists_valid) build_ghosts_and_lists(prepass_on);
  const int nmax = (cfg.isQEq == 1) ? cfg.NMAXQEq : 1;
  // one wavefront per row, sixteen rows per workgroup: measured faster than a persistent grid-stride launch (1.10 vs 1.28 ms
  // per pass at 979,776 rows) -- many short waves overlap each other's load / gather / reduce phases
  constexpr int SPMV_WPB = 16;                   // wavefronts (= rows) per workgroup of the matrix pass
  const int rb = nblk(N, SPMV_WPB);
  const int vb = std::min(nblk(N, 256), 2048);
  // the update kernel finishes its own reduction (last workgroup: one set of partials per workgroup, then the scalar algebra): with one
  // workgroup per CU that tail is short -- 45.8 / 39.3 / 33.8 / 32.6 us per launch at 2048 / 1024 / 512 / 256 workgroups, 68 at 4096
  const int vb_upd = std::min(nblk(N, 256), 256);
  double *lvl1 = partials + partials_cap;      // 128 x 4 first-level sums live behind the per-workgroup partials (fixed offset: independent of the cell count of a sparse box)
  static const int swz = std::getenv("RXMD_NO_XCD_SWIZZLE") ? 0 : 1;
  const bool pipe = (std::getenv("RXMD_SPMV_NO_PIPE") == nullptr);        // read per call: the tests switch it
  // the ring kernel (k_spmv_ring): one persistent workgroup per CU.  RXMD_SPMV_RING=0 keeps the wavefront-per-row kernel (which also serves
  // small systems -- a persistent launch has nothing to stream there -- and rows that would not fit the ring)
  static const int ring_env = std::getenv("RXMD_SPMV_RING") ? std::atoi(std::getenv("RXMD_SPMV_RING")) : 0;
  const int spmv2_env = std::getenv("RXMD_SPMV2") ? std::atoi(std::getenv("RXMD_SPMV2")) : 0;    // read per call
  const int dma_env = std::getenv("RXMD_SPMV_DMA") ? std::atoi(std::getenv("RXMD_SPMV_DMA")) : 0;       // read per call: the tests switch it
  const int dma_wpb_env = std::getenv("RXMD_DMA_WPB") ? std::atoi(std::getenv("RXMD_DMA_WPB")) : 0;
  const int dma_lds_env = std::getenv("RXMD_DMA_LDS") ? std::atoi(std::getenv("RXMD_DMA_LDS")) : 0;
  const int dma_spec_env = std::getenv("RXMD_DMA_SPEC") ? std::atoi(std::getenv("RXMD_DMA_SPEC")) : 1;
  const bool win_env = std::getenv("RXMD_SPMV_WIN") == nullptr || std::atoi(std::getenv("RXMD_SPMV_WIN")) != 0;   // read per call: the tests switch it
  static const int ring_R_env = std::getenv("RXMD_RING_R") ? std::atoi(std::getenv("RXMD_RING_R")) : 0;
  static const int ring_C = std::getenv("RXMD_RING_C") ? std::max(1, std::min(RING_MAXC, std::atoi(std::getenv("RXMD_RING_C")))) : RING_MAXC;
  static const int ring_min_rows = std::getenv("RXMD_RING_MIN_ROWS") ? std::atoi(std::getenv("RXMD_RING_MIN_ROWS")) : 16384;
  static const int ring_wg_env = std::getenv("RXMD_RING_WG") ? std::atoi(std::getenv("RXMD_RING_WG")) : 0;
  static const int ring_cyclic = std::getenv("RXMD_RING_CYCLIC") ? std::atoi(std::getenv("RXMD_RING_CYCLIC")) : 1;
  static const int ring_group = std::getenv("RXMD_RING_GROUP") ? std::atoi(std::getenv("RXMD_RING_GROUP")) : 0;
  static const int ring_probe = std::getenv("RXMD_RING_PROBE") ? std::atoi(std::getenv("RXMD_RING_PROBE")) : 0;
  const int ring_R = ring_R_env > 0 ? (ring_R_env & ~63) : (ff.pqeq ? 3584 : 6144);        // ring entries of 12 (PQEq: 20) bytes: with the control block < 80 KB, two workgroups per CU
  const size_t ring_lds = static_cast<size_t>(ring_R) * (ff.pqeq ? 20 : 12) + sizeof(RingCtl);
  const int ring_wgs = ring_wg_env > 0 ? ring_wg_env : 2 * num_cu;
  // returns the number of partial-sum sets (of four) the launch leaves behind partials[pbase * 4]
  const double *stopflag = nullptr;            // run-ahead CG loop only: kernels of an iteration return at once when scal[S_STOP] is set
  auto pass = [&](int mode, bool store, double2 *ra, double2 *rg, const int *rowlist = nullptr, int nrows = 0, int pbase = 0) -> int {
    const int nr = rowlist ? nrows : N;
    win_used = false;
    if (win_valid && win_env && (!rowlist || rowlist == rows_int || rowlist == rows_bnd)) {   // window pass: the group's partners in LDS, 16-bit slots (k_spmv_win)
      win_used = true;
      const int *glist = !rowlist ? nullptr : (rowlist == rows_int ? win_gint : win_gbnd);    // multi-rank overlap: interior groups while the halo is in flight, then the rest
      const int ng = !rowlist ? win_groups : (rowlist == rows_int ? win_groups - win_nbnd : win_nbnd);
      if (ng == 0) return 0;
      const size_t lds = static_cast<size_t>(win_maxunits) * WIN_UNIT * sizeof(double2);
#define RX_WIN3(M, S, P) k_spmv_win<M, S, P><<<ng, 64 * WIN_ROWS, lds, stream>>>(N, G, S10, dff, sl10, hess, n10, rows_sorted, win_k, win_cnt, xs, hst, gst, qst, q, type, scal, partials, ra, rg, hsc, pqrow, glist, ng, pbase, stopflag)
#define RX_WIN(M, S) do { if (ff.pqeq) RX_WIN3(M, S, true); else RX_WIN3(M, S, false); } while (0)
      if (mode == MODE_HSH) { if (store) RX_WIN(MODE_HSH, true); else RX_WIN(MODE_HSH, false); }
      else { if (store) RX_WIN(MODE_GRAD, true); else RX_WIN(MODE_GRAD, false); }
#undef RX_WIN
#undef RX_WIN3
      return ng;
    }
    // (not with PQEq: at the 64 registers two workgroups per CU allow, its instances spill, and a scratch access in the loader wave would break
    // the loader's own count of its outstanding DMA instructions)
    if (ring_env && !ff.pqeq && max_row10 <= std::min(ring_R, 1024) && nr >= ring_min_rows && ring_lds <= 160 * 1024 && ff.nso <= 15) {
      const int nwg = std::max(1, std::min(ring_wgs, nr / (2 * ring_C)));
      const int cyclic = (ring_cyclic && nwg >= 8 && (nwg & 7) == 0 && swz) ? 1 : 0;
      const int which = rowlist == nullptr ? 0 : (rowlist == rows_int ? 1 : 2);
      if (!rsched[which]) RX_HIP(hipMalloc(reinterpret_cast<void **>(&rsched[which]), sizeof(int2) * (static_cast<size_t>(rows10) + 2 * 4096 + 64)));
      if (!rsched_valid[which]) {                  // once per list build: (row, length) pairs in each workgroup's streaming order
        const int tot = nwg * ring_rows_per_wg(nr, nwg);
        k_ring_schedule<<<nblk(tot, 256), 256, 0, stream>>>(nr, nwg, cyclic, rowlist, n10, rsched[which]);
        rsched_valid[which] = true;
      }
      const int2 *sched = rsched[which];
#define RX_RING4(M, S, P, K)                                                                                                               \
  do {                                                                                                                                     \
    static bool attr_set = false;                                                                                                          \
    if (!attr_set) { RX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_spmv_ring<M, S, P, K>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr_set = true; } \
    k_spmv_ring<M, S, P, K><<<nwg, 64 * (ring_C + RING_NL), ring_lds, stream>>>(N, S10, dff, nb10, hess, n10, xs, hst, gst, qst, q, type, scal, partials, ra, rg, hsc, pqrow, swz | (ring_probe << 8) | (ring_group << 16), sched, cyclic, nr, pbase, ring_R, ring_C); \
  } while (0)
#define RX_RING3(M, S, P) do { if (max_row10 <= 512) RX_RING4(M, S, P, 4); else RX_RING4(M, S, P, 8); } while (0)
#define RX_RING(M, S) do { if (ff.pqeq) RX_RING3(M, S, true); else RX_RING3(M, S, false); } while (0)
      if (mode == MODE_HSH) { if (store) RX_RING(MODE_HSH, true); else RX_RING(MODE_HSH, false); }
      else { if (store) RX_RING(MODE_GRAD, true); else RX_RING(MODE_GRAD, false); }
#undef RX_RING
#undef RX_RING4
#undef RX_RING3
      return nwg;
    }
    if (dma_env && max_row10 <= 1024 && ff.nso <= 15) {
      const int kh = max_row10 <= 512 ? 4 : 8;
      const int slot = 128 * kh * (ff.pqeq ? 20 : 12);
      const int wpb = std::max(1, std::min(dma_wpb_env > 0 ? dma_wpb_env : 16, (dma_lds_env > 0 ? dma_lds_env : 80 * 1024) / slot));
      const int nbl = nblk(nr, wpb);
      if (nbl == 0) return 0;
      // rows nearly equally long (the longest within 12 % of the shortest): request every row's streams for the longest row's length, no dependent row-length load first
      const int dma_n4 = (dma_spec_env && min_row10 > 0 && max_row10 * 100 <= min_row10 * 112) ? ((max_row10 + 3) & ~3) : 0;
#define RX_DMA4(M, S, P, K)                                                                                                                \
  do {                                                                                                                                     \
    static bool attr_set = false;                                                                                                          \
    if (!attr_set) { RX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_spmv_dma<M, S, P, K>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024)); attr_set = true; }   /* + the static staging of block_store_partials */ \
    k_spmv_dma<M, S, P, K><<<nbl, 64 * wpb, static_cast<size_t>(wpb) * slot, stream>>>(N, S10, dff, nb10, hess, n10, xs, hst, gst, qst, q, type, scal, partials, ra, rg, hsc, pqrow, swz, rowlist, nr, pbase, dma_n4); \
  } while (0)
#define RX_DMA3(M, S, P) do { if (kh == 4) RX_DMA4(M, S, P, 4); else RX_DMA4(M, S, P, 8); } while (0)
#define RX_DMA(M, S) do { if (ff.pqeq) RX_DMA3(M, S, true); else RX_DMA3(M, S, false); } while (0)
      if (mode == MODE_HSH) { if (store) RX_DMA(MODE_HSH, true); else RX_DMA(MODE_HSH, false); }
      else { if (store) RX_DMA(MODE_GRAD, true); else RX_DMA(MODE_GRAD, false); }
#undef RX_DMA
#undef RX_DMA3
#undef RX_DMA4
      return nbl;
    }
    if (spmv2_env) {                              // two rows per wavefront, side by side
      const int rb2 = nblk((nr + 1) / 2, SPMV_WPB);
#define RX_P2(M, S) do { if (ff.pqeq) k_spmv2<M, S, true><<<rb2, 64 * SPMV_WPB, 0, stream>>>(N, S10, dff, nb10, hess, n10, xs, hst, gst, qst, q, type, scal, partials, ra, rg, hsc, pqrow, swz, rowlist, nr, pbase, stopflag); \
                         else k_spmv2<M, S, false><<<rb2, 64 * SPMV_WPB, 0, stream>>>(N, S10, dff, nb10, hess, n10, xs, hst, gst, qst, q, type, scal, partials, ra, rg, hsc, pqrow, swz, rowlist, nr, pbase, stopflag); } while (0)
      if (mode == MODE_HSH) { if (store) RX_P2(MODE_HSH, true); else RX_P2(MODE_HSH, false); }
      else { if (store) RX_P2(MODE_GRAD, true); else RX_P2(MODE_GRAD, false); }
#undef RX_P2
      return rb2;
    }
    const int rbl = rowlist ? nblk(nrows, SPMV_WPB) : rb;
    if (rbl == 0) return 0;
#define RX_PASS3(M, S, P, PI) k_spmv<M, S, P, PI><<<rbl, 64 * SPMV_WPB, 0, stream>>>(N, S10, dff, nb10, hess, n10, xs, hst, gst, qst, q, type, scal, partials, ra, rg, hsc, pqrow, swz, rowlist, nrows, pbase, stopflag)
#define RX_PASS(M, S)                                                                                                  \
  do {                                                                                                                 \
    if (ff.pqeq) { if (pipe) RX_PASS3(M, S, true, 1); else RX_PASS3(M, S, true, 0); }                                  \
    else { if (pipe) RX_PASS3(M, S, false, 1); else RX_PASS3(M, S, false, 0); }                                        \
  } while (0)
    if (mode == MODE_HSH) { if (store) RX_PASS(MODE_HSH, true); else RX_PASS(MODE_HSH, false); }
    else { if (store) RX_PASS(MODE_GRAD, true); else RX_PASS(MODE_GRAD, false); }
#undef RX_PASS
#undef RX_PASS3
    return rbl;
  };
  auto reduce = [&](int stage, int nb_) {
    if (!multi()) {                              // single rank: level-1 sums, final sum and scalar algebra in one launch
      if (nb_ > 1024) k_reduce_fused<<<128, 256, 0, stream>>>(nb_, partials, lvl1, tickets, stage, scal, stopflag);
      else k_reduce_fused<<<1, 256, 0, stream>>>(nb_, partials, lvl1, tickets, stage, scal, stopflag);
      return;
    }
    if (nb_ > 1024) k_reduce_fused<<<128, 256, 0, stream>>>(nb_, partials, lvl1, tickets, 0, scal);   // stage 0: rank-local sums only
    else k_reduce_fused<<<1, 256, 0, stream>>>(nb_, partials, lvl1, tickets, 0, scal);
    allreduce_scal4();                           // MPI_ALLREDUCE of the rank-local sums
    k_scalar_algebra<<<1, 64, 0, stream>>>(stage, scal);
  };
  RX_HIP(hipMemsetAsync(scal, 0, sizeof(double) * 32, stream));
  h_scal[60] = cfg.QEq_tol;
  RX_HIP(hipMemcpyAsync(scal + S_TOL, h_scal + 60, sizeof(double), hipMemcpyHostToDevice, stream));
  const bool onepass = (cfg.qeq_mode == 1);
  if (sums_from_list) {
    k_grad_start<<<vb, 256, 0, stream>>>(N, dff, type, qst, q, sall, sgh, gst, partials, pqrow);
    reduce(3, vb);
  } else {
    qeq_start_vectors();
    const int np0 = pass(MODE_GRAD, onepass, onepass ? sall : nullptr, onepass ? sgh : nullptr);
    reduce(3, np0);
  }
  k_direction<<<nblk(N, 256), 256, 0, stream>>>(N, 1, scal, gst, hst);
  RX_HIP(hipMemcpyAsync(h_scal, scal, sizeof(double) * S_COUNT, hipMemcpyDeviceToHost, stream));
  sync_stream();
  double GEst2 = 1e99, Est = h_scal[S_EST];
  int it = 0;
  float ms = 0;
  bool xs_current = false;       // the fused direction kernel leaves the sorted copy of the new (hs,ht) in xs
  const bool overlap_on = (std::getenv("RXMD_NO_HALO_OVERLAP") == nullptr);     // read per call: the tests switch it
  const bool est_with_update = (std::getenv("RXMD_EST_SEPARATE") == nullptr);
  const bool cg_scatter = (std::getenv("RXMD_CG_NO_SCATTER") == nullptr);
  const bool overlap = overlap_on && multi() && onepass && !rows_split_pending_invalid();
  bool halo_in_flight = false;
  // ---- run-ahead loop (single rank, qeq_mode 1, plain QEq; RXMD_CG_NO_RUNAHEAD=1 switches it off) ------------------------------------
  // The host is one iteration BEHIND the device: iteration it is queued in full before the host has seen the Est that decides whether it
  // happens.  The decision (qeq.F90:114-115) is made on the device where Est becomes final (scalar_algebra stage 6 -> scal[S_STOP]); the
  // kernels of an iteration that is not to happen return at once; the host reads the same flag.  Iteration counts, charges and every
  // sum are those of the loop below -- what changes is that no host round trip lies between two iterations (on a host that shares its
  // cores with other jobs the blocking loop lost 90 us per iteration: 172 against 80 us of everything that is not the matrix pass).
  const bool runahead = !multi() && onepass && !ff.pqeq && est_with_update && std::getenv("RXMD_CG_NO_RUNAHEAD") == nullptr;
  if (runahead) {
    auto exit_test = [&](double prev, double est) {
      return (0.5 * (std::fabs(prev) + std::fabs(est)) < cfg.QEq_tol) || (std::fabs(prev) > 0.0 && std::fabs(est / prev - 1.0) < cfg.QEq_tol);
    };
    auto enqueue = [&](int k) {                    // everything of iteration k; on the device a no-op when its stop flag is set
      stopflag = (k == 0) ? nullptr : scal + S_STOP + (k & 1);    // iteration 0 is decided by the host (Est of the start vector is here already)
      if (!xs_current) sorted_copy(hst);           // first iteration only: afterwards the direction kernel leaves the sorted copy behind
      xs_current = false;
      hipEventRecord(ev_pass[k & 1][0], stream);   // read once the host has confirmed that the iteration happened (a pass that returned at once is not timed)
      const int np1 = pass(MODE_HSH, true, wall, wgh);
      hipEventRecord(ev_pass[k & 1][1], stream);
      reduce(1, np1);
      k_cg_update<true><<<vb_upd, 256, 0, stream>>>(N, dff, scal, type, hst, qst, wall, wgh, sall, sgh, gst, partials, tickets + 1, pqrow, 6 | (((k + 1) & 1) << 4), stopflag);
      RX_HIP(hipMemcpyAsync(h_scal + 64 + 64 * (k & 1), scal, sizeof(double) * S_COUNT, hipMemcpyDeviceToHost, stream));
      RX_HIP(hipEventRecord(ev_spec[k & 1], stream));
      const bool scatter = cg_scatter && k + 1 <= nmax - 1;
      k_cg_direction<<<vb, 256, 0, stream>>>(N, dff, scal, type, gst, hst, hst2, qst, sall, sgh, q, partials, tickets + 2, pqrow, -1, G, invpos, groot, scatter ? xs : nullptr, stopflag);
      std::swap(hst, hst2);
      if (k + 1 <= nmax - 1) { if (!scatter) sorted_copy(hst); xs_current = true; }
    };
    it = 0;
    if (nmax >= 1 && !exit_test(GEst2, Est)) {
      GEst2 = Est;
      enqueue(0);
      for (it = 1;; ++it) {
        const bool queued = it <= nmax - 1;
        if (queued) enqueue(it);                   // ahead of the decision
        sync_event(ev_spec[(it - 1) & 1]);         // iteration it - 1 has produced its Est and the decision about iteration it
        collect_timers();
        const double *hs = h_scal + 64 + 64 * ((it - 1) & 1);
        Est = hs[S_EST];
        { float pms = 0; if (hipEventElapsedTime(&pms, ev_pass[(it - 1) & 1][0], ev_pass[(it - 1) & 1][1]) == hipSuccess) st.ms_qeq_spmv += pms; }
        st.spmv_launches += 1;
        if (!queued) break;                        // NMAXQEq iterations done
        if (hs[S_STOP + (it & 1)] != 0.0) { std::swap(hst, hst2); st.spmv_noop_launches += 1; break; }       // iteration it did not happen: its direction kernel wrote nothing, undo the swap
        GEst2 = Est;
      }
      stopflag = nullptr;
    }
    if (ff.pqeq) pqeq_update_shells();
    nstep_qeq = it; last_est = Est;
    st.qeq_iters_last = it; st.qeq_iters_total += it; st.qeq_calls += 1;
    sync_stream();                                 // the kernels of an iteration that did not happen are still in the queue: cheap, but they read scal
    collect_timers();
    st.ms_qeq += toc(6, 7);
    return;
  }
  for (it = 0; it <= nmax - 1; ++it) {
    if (0.5 * (std::fabs(GEst2) + std::fabs(Est)) < cfg.QEq_tol) break;                          // qeq.F90:114
    if (std::fabs(GEst2) > 0.0 && std::fabs(Est / GEst2 - 1.0) < cfg.QEq_tol) break;            // qeq.F90:115
    GEst2 = Est;
    if (halo_in_flight) {
      // multi-rank overlap: the (hs,ht) halo of this iteration was started on the second stream right after the direction update;
      // rows without a ghost partner do not need it
      const int n_int = N - n_bnd;
      hipEventRecord(ev[2], stream);
      const int np_int = pass(MODE_HSH, onepass, wall, wgh, rows_int, n_int, 0);
      join_comm_stream();
      const int np_bnd = pass(MODE_HSH, onepass, wall, wgh, rows_bnd, n_bnd, np_int);
      hipEventRecord(ev[3], stream);
      reduce(1, np_int + np_bnd);
      halo_in_flight = false;
    } else {
      if (!xs_current) sorted_copy(hst);                                                         // QCOPY2, qeq.F90:93,164
      xs_current = false;
      hipEventRecord(ev[2], stream);
      const int np1 = pass(MODE_HSH, onepass, onepass ? wall : nullptr, onepass ? wgh : nullptr);
      hipEventRecord(ev[3], stream);
      reduce(1, np1);
    }
    if (onepass) {       // qeq_mode 1: one matrix pass per iteration; gradient and Est by recurrence on the stored row sums
      const bool fuse = !multi();                  // single rank: every reduction finishes in-kernel; multi: sums, all-reduce, algebra
      // plain QEq: Est rides on the update kernel's sums (quadratic in mu, scalar_algebra stage 6) -- multi-rank: two all-reduces per
      // iteration instead of three; any rank count: Est is final BEFORE the direction kernel, so its copy to the host, the host's exit
      // test and the launch of the next matrix pass all run underneath the direction update and the sorted copy / halo
      const bool est3 = !ff.pqeq && est_with_update;
      if (est3) k_cg_update<true><<<vb_upd, 256, 0, stream>>>(N, dff, scal, type, hst, qst, wall, wgh, sall, sgh, gst, partials, tickets + 1, pqrow, fuse ? 6 : 0, nullptr);
      else k_cg_update<false><<<vb_upd, 256, 0, stream>>>(N, dff, scal, type, hst, qst, wall, wgh, sall, sgh, gst, partials, tickets + 1, pqrow, fuse ? 4 : 0, nullptr);
      if (!fuse) { allreduce_scal4(est3 ? 8 : 4); k_scalar_algebra<<<1, 64, 0, stream>>>(est3 ? 6 : 4, scal); }
      if (est3) { RX_HIP(hipMemcpyAsync(h_scal, scal, sizeof(double) * S_COUNT, hipMemcpyDeviceToHost, stream)); RX_HIP(hipEventRecord(ev_est, stream)); }
      // the direction update runs over the residents in atom order (every access coalesced); the cell-sorted copy with the images
      // is one gather pass queued behind it (sorted_copy).  Doing both in one kernel over the sorted positions
      // (five random 16-byte accesses per atom) was 0.4 ms per step slower.
      // single rank: the direction kernel also scatters the new (hs,ht) to its cell-sorted positions, residents and their periodic images -- the
      // separate gather pass (k_sorted_vec, 12 us per iteration) is gone (RXMD_CG_NO_SCATTER=1 restores it)
      const bool scatter = fuse && cg_scatter && it + 1 <= nmax - 1;
      k_cg_direction<<<est3 ? vb : vb_upd, 256, 0, stream>>>(N, dff, scal, type, gst, hst, hst2, qst, sall, sgh, q, partials, tickets + 2, pqrow, est3 ? -1 : (fuse ? 5 : 0),
                                                                 G, invpos, groot, scatter ? xs : nullptr, nullptr);
      if (!fuse && !est3) { allreduce_scal4(); k_scalar_algebra<<<1, 64, 0, stream>>>(5, scal); }
      if (!est3) {       // PQEq: Est comes out of the direction kernel; the host still waits for this copy only, not for the sorted copy behind it
        RX_HIP(hipMemcpyAsync(h_scal, scal, sizeof(double) * S_COUNT, hipMemcpyDeviceToHost, stream));
        RX_HIP(hipEventRecord(ev_est, stream));
      }
      std::swap(hst, hst2);
      xs_current = false;
      if (!overlap && it + 1 <= nmax - 1) {        // sorted copy (multi-rank: after the (hs,ht) halo) queued before the host waits for Est
        if (!scatter) sorted_copy(hst);
        xs_current = true;
      }
      if (overlap && it + 1 <= nmax - 1) {         // overlapped: residents' part of the sorted copy now, halo + ghosts' part on the second stream
        k_sorted_part<<<nblk(G, 256), 256, 0, stream>>>(G, N, perm, hst, xs, 0);
        on_comm_stream([&] {
          halo_staged(reinterpret_cast<double *>(hst), 2);
          k_sorted_part<<<nblk(G, 256), 256, 0, stream>>>(G, N, perm, hst, xs, 1);
        });
        halo_in_flight = true;
      }
      sync_event(ev_est);
      collect_timers();
      Est = h_scal[S_EST];
      hipEventElapsedTime(&ms, ev[2], ev[3]); st.ms_qeq_spmv += ms;
      st.spmv_launches += 1;
      continue;
    }
    k_update_qst<<<vb, 256, 0, stream>>>(N, scal, hst, qst, partials);
    reduce(2, vb);
    k_apply_q<<<nblk(N, 256), 256, 0, stream>>>(N, scal, qst, q);
    sorted_copy(qst);                                                                            // QCOPY1, qeq.F90:153
    hipEventRecord(ev[4], stream);
    const int np2 = pass(MODE_GRAD, false, nullptr, nullptr);
    hipEventRecord(ev[5], stream);
    reduce(3, np2);
    k_direction<<<nblk(N, 256), 256, 0, stream>>>(N, 0, scal, gst, hst);
    RX_HIP(hipMemcpyAsync(h_scal, scal, sizeof(double) * S_COUNT, hipMemcpyDeviceToHost, stream));
    sync_stream();
    Est = h_scal[S_EST];
    hipEventElapsedTime(&ms, ev[2], ev[3]); st.ms_qeq_spmv += ms;
    hipEventElapsedTime(&ms, ev[4], ev[5]); st.ms_qeq_spmv += ms;
    st.spmv_launches += 2;
  }
  if (halo_in_flight) join_comm_stream();             // the loop ended while a halo it will not use was still in flight
  if (ff.pqeq) pqeq_update_shells();                  // pqeq.F90:169
  nstep_qeq = it; last_est = Est;
  st.qeq_iters_last = it; st.qeq_iters_total += it; st.qeq_calls += 1;
  st.ms_qeq += toc(6, 7);
}

}  // namespace rxmd
