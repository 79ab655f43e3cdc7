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
#include "spmv_kernels.h"

#include <chrono>
#include <map>
#include <mutex>
#include <tuple>
#include <cstdio>

#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <memory>
#include <utility>
#include <vector>

namespace rxmd {


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
__device__ inline void scalar_algebra(int stage, double *__restrict__ scal, double *hsnap = nullptr, double hseq = 0.0) {
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
      // a snapshot of the scalars of THIS iteration (parity par ^ 1) for the host: copied by a second stream while the main stream goes on with the
      // direction kernel; the slot is rewritten two iterations later, after the host has read it (run-ahead loop of Engine::qeq)
      double *snap = scal + S_SNAP + 32 * (par ^ 1);
      for (int c = 0; c < S_COUNT; ++c) snap[c] = scal[c];
      // run-ahead loop (round 6): the same snapshot straight into PINNED HOST memory, then its sequence number -- the host polls that word.  Until round 6
      // an event on the main stream, a wait + a 200-byte copy + an event on the second stream carried it: the event record alone cost the dependent
      // chain update -> direction ~7 us per iteration.
      if (hsnap != nullptr) {
        double *hp = hsnap + 64 * (par ^ 1);
        for (int c = 0; c < S_COUNT; ++c) __hip_atomic_store(hp + c, scal[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");                  // system scope: the data before the sequence number
        __hip_atomic_store(hp + 63, hseq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
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
__device__ inline void block_finish(int nblocks, double *partials, unsigned *ticket, int stage, double *scal, int nsets = 1, double *hsnap = nullptr, double hseq = 0.0) {
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
    if (stage > 0) scalar_algebra(stage, scal, hsnap, hseq);
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
                                                    double2 *__restrict__ gst, double *__restrict__ partials, unsigned *ticket, const double4 *__restrict__ pqrow, int stage, const double *__restrict__ stopflag,
                                                    double *hsnap = nullptr, double hseq = 0.0, const int *__restrict__ rows = nullptr, int nres = 0) {
  // rows != nullptr (row order, run-ahead loop): the arrays are indexed by the rows' places in rows_sorted, N = the number of places, a place whose entry is
  // >= nres holds no row
  if (stopflag && *stopflag != 0.0) return;
  const double l1 = scal[S_LMIN_S], l2 = scal[S_LMIN_T];
  double s = 0.0, t = 0.0, g1s = 0.0, g2s = 0.0, e0 = 0.0, e1 = 0.0, e2 = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
    if (rows != nullptr && rows[i] >= nres) continue;
    double2 qv = qst[i];
    const double2 hv = hst[i];
    qv.x = qv.x + l1 * hv.x; qv.y = qv.y + l2 * hv.y;
    qst[i] = qv;
    // (skipping the ghost-column sums of rows without a ghost partner -- identically zero, 48 of the 164 bytes per row -- behind a per-row flag was
    //  measured: 38 against 33 us per launch, the predicated accesses cost more than they save)
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
  block_finish(gridDim.x, partials, ticket, stage, scal, EST3 ? 2 : 1, hsnap, hseq);   // stage 4, or 0 = sums only (the all-reduce of a multi-rank run comes first)
}
// B: new direction h = g + beta h written to the other (hs,ht) buffer; q = qs - mu qt and the Est term (qeq.F90:150,160-164,297-306)
//    -> tail: Est (stage 5), sums only (0) or no reduction at all (-1: Est came with the update kernel's sums)
//    QEST = false (Est came with the update kernel's sums, stage -1): only the direction -- the charges q = qs - mu qt are formed once, behind the
//    last iteration (k_apply_q), instead of in every iteration: three 16-byte reads and an 8-byte write per atom and iteration less
template <bool QEST>
__global__ void __launch_bounds__(256) k_cg_direction(int N, DevFF ff, double *__restrict__ scal, const int *__restrict__ type,
                                                       const double2 *__restrict__ gst, const double2 *__restrict__ hst, double2 *__restrict__ hst_new,
                                                       const double2 *__restrict__ qst, const double2 *__restrict__ sall, const double2 *__restrict__ sgh, double *__restrict__ q,
                                                       double *__restrict__ partials, unsigned *ticket, const double4 *__restrict__ pqrow, int stage,
                                                       int G, const int *__restrict__ invpos, const int *__restrict__ groot, double2 *__restrict__ xs, const double *__restrict__ stopflag,
                                                       const int *__restrict__ rows = nullptr, int nres = -1, const int *__restrict__ xpos = nullptr) {
  // rows != nullptr (row order): N places, xpos[place] = the row's cell-sorted position, groot[ghost] = the PLACE of the ghost's owner, nres = residents
  if (stopflag && *stopflag != 0.0) return;
  const double mu = scal[S_MU], b1 = scal[S_BETA_S], b2 = scal[S_BETA_T];
  double es = 0.0;
  if (nres < 0) nres = N;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
    if (rows != nullptr && rows[i] >= nres) continue;
    const double2 g = gst[i], h = hst[i];
    const double2 hn = make_double2(g.x + b1 * h.x, g.y + b2 * h.y);
    hst_new[i] = hn;
    if (xs) xs[xpos != nullptr ? xpos[i] : invpos[i]] = hn;                      // single rank: the cell-sorted gather copy of the next matrix pass (QCOPY2, qeq.F90:164) is written here ...
    if (!QEST) continue;
    const DevAtomP ap = ff.atom[type[i]];
    const double2 qv = qst[i], a = sall[i], gh = sgh[i];
    const double qi = qv.x - mu * qv.y;
    q[i] = qi;
    const double hq_all = a.x - mu * a.y, hq_res = (a.x - gh.x) - mu * (a.y - gh.y);
    if (pqrow) es += pq_est_row(ap, ff.Zpq[type[i]], pqrow[i], qi, hq_all, gh.x - mu * gh.y);
    else es += ap.chi * qi + 0.5 * ap.eta * qi * qi + 0.5 * qi * (hq_all + hq_res);
  }
  if (xs)                                            // ... including the periodic images: a ghost recomputes the value of its owner (two coalesced-by-owner reads)
    for (int t = nres + blockIdx.x * blockDim.x + threadIdx.x; t < G; t += gridDim.x * blockDim.x) {
      const int r = groot[t];
      const double2 g = gst[r], h = hst[r];
      xs[invpos[t]] = make_double2(g.x + b1 * h.x, g.y + b2 * h.y);
    }
  if (!QEST || stage < 0) return;
  double acc[4] = {wave_sum(es), 0.0, 0.0, 0.0};
  block_store_partials<4>(acc, partials, 4);
  block_finish(gridDim.x, partials, ticket, stage, scal);     // stage 5 or 0
}

// ---- row order of the run-ahead CG loop (round 6) ----------------------------------------------------------------------------------------------
// For the iterations of ONE QEq call the CG vectors of the residents live in the order of rows_sorted (place = group * WIN_ROWS + wavefront of the window
// pass): what the pass reads and writes per row -- (hs,ht), (gs,gt), type, length; the two row sums -- is then one contiguous run per workgroup, and the
// direction kernel's scatter into the cell-sorted gather copy is nearly a copy (rows_sorted IS cell-sorted order).  Entered behind the start gradient, left
// with the charges: two gathers per call.  The reductions of the vector kernels run over places instead of atoms: the same numbers in another order.
__global__ void k_rows_enter(int R, int N, const int *__restrict__ rows, const int *__restrict__ type, const int *__restrict__ n10, const int *__restrict__ invpos,
                             const double2 *__restrict__ qst, const double2 *__restrict__ hst, const double2 *__restrict__ gst, const double2 *__restrict__ sall, const double2 *__restrict__ sgh,
                             double2 *__restrict__ r_qst, double2 *__restrict__ r_hst, double2 *__restrict__ r_gst, double2 *__restrict__ r_sall, double2 *__restrict__ r_sgh,
                             int *__restrict__ r_type, int *__restrict__ r_n10, int *__restrict__ r_xpos, int *__restrict__ rpos) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const int a = rows[r];
  if (a >= N || a < 0) { r_type[r] = 1; r_n10[r] = 0; r_xpos[r] = 0; return; }      // not a row: every kernel skips the place by rows[r] >= N
  r_qst[r] = qst[a]; r_hst[r] = hst[a]; r_gst[r] = gst[a]; r_sall[r] = sall[a]; r_sgh[r] = sgh[a];
  r_type[r] = type[a]; r_n10[r] = n10[a]; r_xpos[r] = invpos[a]; rpos[a] = r;
}
__global__ void k_ghost_rows(int N, int G, const int *__restrict__ groot, const int *__restrict__ rpos, int *__restrict__ g_rrow) {
  const int t = N + blockIdx.x * blockDim.x + threadIdx.x;
  if (t < G) g_rrow[t] = rpos[groot[t]];
}
// back to atom order: (qs,qt) and q = qs - mu qt (qeq.F90:150) with the mu of the last iteration that happened
__global__ void k_rows_exit(int R, int N, const int *__restrict__ rows, const double *__restrict__ scal, const double2 *__restrict__ r_qst, double2 *__restrict__ qst, double *__restrict__ q) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const int a = rows[r];
  if (a >= N || a < 0) return;
  const double2 v = r_qst[r];
  qst[a] = v;
  q[a] = v.x - scal[S_MU] * v.y;
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
  const bool kt = kt_begin(&st.ms_allreduce, nullptr, &st.allreduce_calls, 0);
  struct End { Engine *e; bool kt; ~End() { e->kt_end(kt); } } end_{this, kt};              // forced staged mode of a single rank without a communicator: nothing to add
  if (nccl) { rccl_allreduce_dev(scal + S_RAW0, n); return; }      // in stream order, no host round trip
  if (!has_comm || !comm.allreduce_sum) throw EngineError(RXMD_E_COMM, "vprocs > 1 needs a transport: call rxmd_hip_set_comm or rxmd_hip_comm_init_rccl first");
  RX_HIP(hipMemcpyAsync(h_scal + 48, scal + S_RAW0, sizeof(double) * n, hipMemcpyDeviceToHost, stream));
  sync_stream();
  if (comm.allreduce_sum(comm.ctx, h_scal + 48, n)) throw EngineError(RXMD_E_COMM, "allreduce callback failed");
  RX_HIP(hipMemcpyAsync(scal + S_RAW0, h_scal + 48, sizeof(double) * n, hipMemcpyHostToDevice, stream));
}

// Where the streams of the window pass lie in physical memory is worth up to 15 % of its time (NOTES.md 3: the same pass on copies of the same
// arrays runs 0.77 ... 0.90 ms, a property of the buffer, repeatable to 0.1-0.5 %, drawn anew by every hipMalloc).  So, once per engine, after the
// first QEq call that used the window pass: a few more placements of the value / slot (/ shell-core) arrays are tried, each timed with 30 launches
// of the real pass, and the fastest is kept.  RXMD_PLACE_TRIES=<n> (default: up to 10 placements including the first, stopping once one is 8 % faster than the slowest seen; 1 switches the search off).
// One-time cost ~35 ms per placement, +6.3 GB of memory while a candidate is alive (979,776 atoms).  Measured, fresh processes alternating on one
// box, default bench: 52.4-54.1 ms/step with the search against 53.4-55.6 without (the pass in the loop follows the kept placement + 0.03-0.04 ms).
void Engine::tune_window_placement() {
  place_tuned = true;
  const int tries = static_cast<int>(opt.place_tries);       // (round 4: the draws of one box lie between 0.80 and 0.93 ms, a third of them fast: ten draws miss the fast kind in 3 % of the processes, six in 12 %)
  if (tries <= 1 || !win_valid || N < 65536) return;              // (small systems: nothing to gain)
  const size_t ne = static_cast<size_t>(rows10) * S10;
  {   // a candidate needs a second copy of the streams while it is timed (6.3 GB at 979,776 atoms): not on a device that is nearly full
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess || fr < 2 * ne * (ff.pqeq ? 18 : 10)) { (void)hipGetLastError(); return; }
  }
  struct Cand {                                                    // freed on every way out, an exception of a copy included
    double *h = nullptr, *c = nullptr; unsigned short *s = nullptr;
    ~Cand() { if (h) (void)hipFree(h); if (s) (void)hipFree(s); if (c) (void)hipFree(c); }
  };
  const size_t lds = static_cast<size_t>(win_maxunits) * WIN_UNIT * sizeof(double2);
  auto time_pass = [&](const double *h, const unsigned short *sl, const double *hc) {
    float ms = 0;
    for (int r = 0; r < 35; ++r) {
      if (r == 5) hipEventRecord(ev[2], stream);
      if (ff.pqeq && opt.pq_prefetch != 0) k_spmv_win<MODE_HSH, true, true, 1, WIN_PREFETCH><<<win_groups, 64 * WIN_ROWS, lds, stream>>>(N, G, S10, dff, sl, h, n10, rows_sorted, win_k, win_cnt, xs, hst, gst, qst, q, type, scal, partials, wall, wgh, hc, pqrow, nullptr, win_groups, 0, nullptr);
      else if (ff.pqeq) k_spmv_win<MODE_HSH, true, true><<<win_groups, 64 * WIN_ROWS, lds, stream>>>(N, G, S10, dff, sl, h, n10, rows_sorted, win_k, win_cnt, xs, hst, gst, qst, q, type, scal, partials, wall, wgh, hc, pqrow, nullptr, win_groups, 0, nullptr);
      else if (opt.spmv_one_trip != 0 && max_row10 > 256 && max_row10 <= 384 && rows_live) k_spmv_win<MODE_HSH, true, false, 3, WIN_LEAN><<<win_groups, 64 * WIN_ROWS, lds, stream>>>(N, G, S10, dff, sl, h, r_n10, rows_sorted, win_k, win_cnt, xs, r_hst, r_gst, qst, q, r_type, scal, partials, r_wall, r_wgh, hc, pqrow, nullptr, win_groups, 0, nullptr, win_flag, 1);
      else if (opt.spmv_one_trip != 0 && max_row10 > 256 && max_row10 <= 384) k_spmv_win<MODE_HSH, true, false, 3, WIN_LEAN><<<win_groups, 64 * WIN_ROWS, lds, stream>>>(N, G, S10, dff, sl, h, n10, rows_sorted, win_k, win_cnt, xs, hst, gst, qst, q, type, scal, partials, wall, wgh, hc, pqrow, nullptr, win_groups, 0, nullptr, win_flag);
      else if (rows_live) k_spmv_win<MODE_HSH, true, false, 2, WIN_PREFETCH | WIN_LEAN><<<win_groups, 64 * WIN_ROWS, lds, stream>>>(N, G, S10, dff, sl, h, r_n10, rows_sorted, win_k, win_cnt, xs, r_hst, r_gst, qst, q, r_type, scal, partials, r_wall, r_wgh, hc, pqrow, nullptr, win_groups, 0, nullptr, win_flag, 1);   // (the form the run-ahead loop launches: per-row operands in row order)
      else k_spmv_win<MODE_HSH, true, false, 2, WIN_PREFETCH | WIN_LEAN><<<win_groups, 64 * WIN_ROWS, lds, stream>>>(N, G, S10, dff, sl, h, n10, rows_sorted, win_k, win_cnt, xs, hst, gst, qst, q, type, scal, partials, wall, wgh, hc, pqrow, nullptr, win_groups, 0, nullptr, win_flag);
    }
    hipEventRecord(ev[3], stream); hipEventSynchronize(ev[3]);
    hipEventElapsedTime(&ms, ev[2], ev[3]);
    return static_cast<double>(ms) / 30.0;
  };
#ifdef RXMD_EXPERIMENTS
  // RXMD_PLACE_SCAN=<slabs> (experiments build): is the placement lottery a RULE?  The two streams of the pass are copied into ONE allocation at
  // controlled offsets -- the value stream at slab + oh, the slot stream behind it at a 2 MB boundary + os -- and the real pass is timed at each
  // (oh, os); repeated for <slabs> fresh allocations.  If the time followed the relative position of the two streams (channel / stack aliasing),
  // it would show as a pattern in os that repeats from slab to slab; a time that changes from slab to slab but not with the offsets inside one
  // says the draw is the physical pages the driver hands out.  Output: one line per measurement on stderr (scripts/gpu_place_scan.sh).
  if (opt.place_scan > 0) {
    if (!ff.pqeq) {
      const int nslab = std::max(1, static_cast<int>(opt.place_scan));
      const size_t MB2 = size_t(2) << 20, hb = ne * sizeof(double), sb = ne * sizeof(unsigned short);
      const size_t pad = size_t(64) << 20, total = hb + sb + 3 * pad;
      const size_t offs[] = {0, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288, size_t(1) << 20, (size_t(1) << 20) + 4096, size_t(2) << 20, size_t(4) << 20, size_t(8) << 20, size_t(16) << 20, size_t(32) << 20};
      std::fprintf(stderr, "[place_scan] engine's own placement: %.4f ms  hess %p sl10 %p  (ne %zu)\n", time_pass(hess, sl10, hsc), static_cast<void *>(hess), static_cast<void *>(sl10), ne);
      std::vector<char *> held;
      for (int sidx = 0; sidx < nslab; ++sidx) {
        char *slab = nullptr;
        if (hipMalloc(reinterpret_cast<void **>(&slab), total) != hipSuccess) { (void)hipGetLastError(); break; }
        held.push_back(slab);                        // kept until the end: a freed slab would come straight back
        auto at = [&](size_t oh, size_t os) {
          double *h2 = reinterpret_cast<double *>(slab + oh);
          unsigned short *s2 = reinterpret_cast<unsigned short *>(slab + ((pad + hb + MB2 - 1) / MB2) * MB2 + os);
          RX_HIP(hipMemcpyAsync(h2, hess, hb, hipMemcpyDeviceToDevice, stream));
          RX_HIP(hipMemcpyAsync(s2, sl10, sb, hipMemcpyDeviceToDevice, stream));
          const double t = time_pass(h2, s2, hsc);
          std::fprintf(stderr, "[place_scan] slab %d %p  oh %9zu  os %9zu  %.4f ms\n", sidx, static_cast<void *>(slab), oh, os, t);
        };
        for (size_t os : offs) at(0, os);
        for (size_t oh : offs) if (oh) at(oh, 0);
        at(0, 0);                                    // repeatability inside the slab
      }
      // (round 6, second series) WHERE does the draw come from?  The same scan point (0, 0) on slabs obtained in other ways: physically contiguous
      // memory (hipExtMallocWithFlags(hipDeviceMallocContiguous)) and virtual-memory-management slabs built from physical chunks of a chosen size
      // (hipMemCreate / hipMemMap: 2 MB, 64 MB, 1 GB) -- if the slow draws are allocations the driver had to back with small page fragments, large
      // chunks are always of the fast kind.  Next to each pass time the plain sequential read of the value stream from the same slab.
      auto seq_ms = [&](const void *ptr) {
        hipEventRecord(ev[2], stream);
        for (int r = 0; r < 5; ++r) k_stream_probe<<<num_cu * 8, 256, 0, stream>>>(hb / 16, reinterpret_cast<const f64x2 *>(ptr), partials);
        hipEventRecord(ev[3], stream); hipEventSynchronize(ev[3]);
        float ms = 0; hipEventElapsedTime(&ms, ev[2], ev[3]); return static_cast<double>(ms) / 5.0;
      };
      auto on_slab = [&](const char *how, char *slab) {
        double *h2 = reinterpret_cast<double *>(slab);
        unsigned short *s2 = reinterpret_cast<unsigned short *>(slab + ((pad + hb + MB2 - 1) / MB2) * MB2);
        RX_HIP(hipMemcpyAsync(h2, hess, hb, hipMemcpyDeviceToDevice, stream));
        RX_HIP(hipMemcpyAsync(s2, sl10, sb, hipMemcpyDeviceToDevice, stream));
        const double t = time_pass(h2, s2, hsc), t_hv = time_pass(h2, sl10, hsc), t_sv = time_pass(hess, s2, hsc);
        std::fprintf(stderr, "[place_scan] %-28s %p  pass %.4f ms (its values + engine slots %.4f, engine values + its slots %.4f)  sequential read of its value stream %.4f ms = %.0f GB/s\n", how, static_cast<void *>(slab), t, t_hv, t_sv, seq_ms(h2), hb / (seq_ms(h2) * 1e-3) / 1e9);
      };
      std::fprintf(stderr, "[place_scan] engine's own value stream: sequential read %.4f ms\n", seq_ms(hess));
      for (char *p : held) on_slab("hipMalloc slab (above)", p);
      for (int c = 0; c < 3; ++c) {
        char *slab = nullptr;
        if (hipExtMallocWithFlags(reinterpret_cast<void **>(&slab), total, hipDeviceMallocContiguous) != hipSuccess) { (void)hipGetLastError(); std::fprintf(stderr, "[place_scan] contiguous allocation refused\n"); break; }
        on_slab("hipDeviceMallocContiguous", slab);
        held.push_back(slab);
      }
      for (size_t chunk : {size_t(2) << 20, size_t(64) << 20, size_t(1) << 30}) {
        for (int rep = 0; rep < 2; ++rep) {
          hipMemAllocationProp prop{};
          prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = cfg.device;
          size_t gran = 0;
          if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess) { (void)hipGetLastError(); std::fprintf(stderr, "[place_scan] no VMM on this device\n"); break; }
          const size_t ck = ((chunk + gran - 1) / gran) * gran, nck = (total + ck - 1) / ck, vsz = nck * ck;
          void *va = nullptr;
          if (hipMemAddressReserve(&va, vsz, ck, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); std::fprintf(stderr, "[place_scan] VMM reserve failed\n"); break; }
          std::vector<hipMemGenericAllocationHandle_t> hs;
          bool ok = true;
          for (size_t k = 0; k < nck && ok; ++k) {
            hipMemGenericAllocationHandle_t hnd;
            ok = hipMemCreate(&hnd, ck, &prop, 0) == hipSuccess;
            if (ok) { hs.push_back(hnd); ok = hipMemMap(static_cast<char *>(va) + k * ck, ck, 0, hnd, 0) == hipSuccess; }
          }
          hipMemAccessDesc ad{}; ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
          if (ok) ok = hipMemSetAccess(va, vsz, &ad, 1) == hipSuccess;
          if (ok) {
            char how[64]; std::snprintf(how, sizeof how, "VMM chunks of %zu MB (gran %zu KB)", ck >> 20, gran >> 10);
            on_slab(how, static_cast<char *>(va));
          } else { (void)hipGetLastError(); std::fprintf(stderr, "[place_scan] VMM slab with %zu MB chunks failed\n", ck >> 20); }
          RX_HIP(hipStreamSynchronize(stream));
          (void)hipMemUnmap(va, vsz);
          for (auto &hnd : hs) (void)hipMemRelease(hnd);
          (void)hipMemAddressFree(va, vsz);
        }
      }
      // and the lottery itself next to it: separate allocations, as the search draws them
      std::vector<std::unique_ptr<Cand>> dr;
      for (int c = 0; c < 6; ++c) {
        dr.push_back(std::make_unique<Cand>());
        Cand &cd_ = *dr.back();
        if (hipMalloc(reinterpret_cast<void **>(&cd_.h), hb) != hipSuccess || hipMalloc(reinterpret_cast<void **>(&cd_.s), sb) != hipSuccess) { (void)hipGetLastError(); break; }
        RX_HIP(hipMemcpyAsync(cd_.h, hess, hb, hipMemcpyDeviceToDevice, stream));
        RX_HIP(hipMemcpyAsync(cd_.s, sl10, sb, hipMemcpyDeviceToDevice, stream));
        std::fprintf(stderr, "[place_scan] separate draw %d: %.4f ms  hess %p sl10 %p\n", c, time_pass(cd_.h, cd_.s, hsc), static_cast<void *>(cd_.h), static_cast<void *>(cd_.s));
        // cross pairs: this draw's value stream with the engine's slot stream and the other way round -- which of the two streams carries the draw?
        std::fprintf(stderr, "[place_scan]   its value stream + the engine's slots: %.4f ms ; the engine's values + its slot stream: %.4f ms\n", time_pass(cd_.h, sl10, hsc), time_pass(hess, cd_.s, hsc));
      }
      dr.clear();
      for (char *p : held) (void)hipFree(p);
    }
  }
#endif
  const auto t_search0 = std::chrono::steady_clock::now();
  double best = time_pass(hess, sl10, hsc);
  st.place_ms_first = best;
  double worst = best;
  st.place_draws = 1; st.place_bytes_held = 0.0;
  // Bounds of the search (round 5).  (1) What a fast placement looks like is a property of the matrix shape on this device: the best time any
  // engine of this process has kept for the same (rows, stride, PQEq) is remembered, and an engine whose FIRST placement is within 3 % of it
  // does not search at all.  (2) At most three losers stay allocated (each a copy of the streams: 4.2 GB at 979,776 atoms, 7.6 GB with PQEq);
  // the oldest is freed only AFTER the next candidate has been allocated, so a freed block cannot come straight back as the "new" draw.
  static std::mutex seen_mx; static std::map<std::tuple<int, int, int>, double> seen_best;
  const auto shape = std::make_tuple(N, S10, ff.pqeq ? 1 : 0);
  {
    std::lock_guard<std::mutex> lk(seen_mx);
    auto it = seen_best.find(shape);
    if (it != seen_best.end() && best <= 1.03 * it->second && !opt.place_all) {
      st.place_ms_kept = best;
      st.place_total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_search0).count();
      return;
    }
  }
  constexpr size_t PLACE_MAX_HELD = 3;
  const bool verbose = opt.place_verbose;
  if (verbose) std::fprintf(stderr, "[rxmd_hip] placement draw 0: %.4f ms  hess %p sl10 %p\n", best, static_cast<void *>(hess), static_cast<void *>(sl10));
  const bool draw_all = opt.place_all;      // diagnosis: no early stop
  // The losers stay allocated until the search is over: a block that is freed comes straight back from the next hipMalloc of its size, and the
  // "next draw" would be the same physical placement again (round 4: a process that kept its first placement at 0.90 ms after ten such draws,
  // while a second process on the same box found 0.82 ms).  Each draw costs a copy of the streams while the search runs, bounded by free memory.
  std::vector<std::unique_ptr<Cand>> drawn;
  for (int c = 1; c < tries; ++c) {
    {
      size_t fr = 0, tot = 0;
      if (hipMemGetInfo(&fr, &tot) != hipSuccess || fr < 2 * ne * (ff.pqeq ? 18 : 10)) { (void)hipGetLastError(); break; }
    }
    drawn.push_back(std::make_unique<Cand>());
    Cand &cd_ = *drawn.back();
    st.place_draws = c + 1;
    double *&h2 = cd_.h, *&c2 = cd_.c; unsigned short *&s2 = cd_.s;
    // (plain hipMalloc: memory from hipExtMallocWithFlags(hipDeviceMallocContiguous) was the fast kind more often in the copies experiment, but
    // an engine that allocated its streams that way failed 15 unrelated tests of the suite with corrupted results -- not used anywhere)
    bool ok = hipMalloc(reinterpret_cast<void **>(&h2), ne * sizeof(double)) == hipSuccess && hipMalloc(reinterpret_cast<void **>(&s2), ne * sizeof(unsigned short)) == hipSuccess;
    if (ok && ff.pqeq) ok = hipMalloc(reinterpret_cast<void **>(&c2), ne * sizeof(double)) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); break; }
    st.place_bytes_held = std::max(st.place_bytes_held, static_cast<double>(drawn.size()) * static_cast<double>(ne) * (ff.pqeq ? 18.0 : 10.0));
    if (drawn.size() > PLACE_MAX_HELD + 1) drawn.erase(drawn.begin());      // the oldest loser goes, now that the new block exists (cd_ stays valid: unique_ptr elements)
    RX_HIP(hipMemcpyAsync(h2, hess, ne * sizeof(double), hipMemcpyDeviceToDevice, stream));
    RX_HIP(hipMemcpyAsync(s2, sl10, ne * sizeof(unsigned short), hipMemcpyDeviceToDevice, stream));
    if (ff.pqeq) RX_HIP(hipMemcpyAsync(c2, hsc, ne * sizeof(double), hipMemcpyDeviceToDevice, stream));
    // Round 6 (profiles/r06_place_scan*.txt): the draw is carried by EACH stream on its own -- a value stream of the slow kind costs 0.08-0.10 ms
    // whatever slot stream runs with it, a slot stream of the slow kind 0.02-0.04 ms -- and not by their relative position (no pattern in 32 offsets
    // inside one allocation), nor by physical contiguity (hipDeviceMallocContiguous draws both kinds), nor by the size of the physical chunks (VMM
    // slabs of 2 MB / 64 MB / 1 GB chunks: all mediocre), and a plain sequential read of a slow value stream is as fast as that of a fast one.  So
    // every candidate stream is timed next to the best partner in hand and adopted on its own: the kept pair is the best value stream and the best
    // slot stream of all draws, not the best PAIR.
    const double t_h = time_pass(h2, sl10, hsc);
    if (verbose) std::fprintf(stderr, "[rxmd_hip] placement draw %d: value stream %p with the slots in hand %.4f ms (best so far %.4f)\n", c, static_cast<void *>(h2), t_h, best);
    worst = std::max(worst, t_h);
    if (t_h < 0.99 * best) { best = t_h; std::swap(hess, h2); }                 // (cd_ now holds the previous best of this stream)
    const double t_s = time_pass(hess, s2, hsc);
    if (verbose) std::fprintf(stderr, "[rxmd_hip] placement draw %d: slot stream %p with the values in hand %.4f ms (best so far %.4f)\n", c, static_cast<void *>(s2), t_s, best);
    worst = std::max(worst, t_s);
    if (t_s < 0.99 * best) { best = t_s; std::swap(sl10, s2); }
    if (ff.pqeq) {
      const double t_c = time_pass(hess, sl10, c2);
      worst = std::max(worst, t_c);
      if (t_c < 0.99 * best) { best = t_c; std::swap(hsc, c2); }
    }
    if (best < 0.92 * worst && c >= 3 && !draw_all) break;          // a value stream of the fast kind is in hand (and a slow one was seen): stop drawing
  }
  drawn.clear();                                        // frees every loser; the stream is idle: time_pass waited for its last launch
  st.place_ms_kept = best;
  st.place_total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_search0).count();
  { std::lock_guard<std::mutex> lk(seen_mx); auto it = seen_best.find(shape); if (it == seen_best.end() || best < it->second) seen_best[shape] = best; }
}

void Engine::qeq() {
  if (!atoms_set) throw EngineError(RXMD_E_STATE, "atoms were never set");
  if (cfg.isQEq != 1 && cfg.isQEq != 2) { nstep_qeq = 0; return; }   // qeq.F90:60-61
  rows_live = false;
  const KtPair t_qeq = outer_begin(&st.ms_qeq);
  // the list sweep of this step can form the row sums of the start vector on the way (saves the matrix pass of qeq.F90:87)
  const bool prepass_on = !opt.qeq_no_prepass;      // (experiments build only)
  sums_from_list = false;
  if (!lists_valid) build_ghosts_and_lists(prepass_on);
  const int nmax = (cfg.isQEq == 1) ? cfg.NMAXQEq : 1;
  // one wavefront per row, sixteen rows per workgroup: measured faster than a persistent grid-stride launch (1.10 vs 1.28 ms
  // per pass at 979,776 rows) -- many short waves overlap each other's load / gather / reduce phases (NOTES.md 3, K4/K5)
  constexpr int SPMV_WPB = 16;                   // wavefronts (= rows) per workgroup of the matrix pass
  const int rb = nblk(N, SPMV_WPB);
  const int vb = std::min(nblk(N, 256), 2048);
  // the update kernel finishes its own reduction (last workgroup: one set of partials per workgroup, then the scalar algebra): with one
  // workgroup per CU that tail is short -- 45.8 / 39.3 / 33.8 / 32.6 us per launch at 2048 / 1024 / 512 / 256 workgroups, 68 at 4096
  const int vb_upd = std::min(nblk(N, 256), 256);
  double *lvl1 = partials + partials_cap;      // 128 x 4 first-level sums live behind the per-workgroup partials (fixed offset: independent of the cell count of a sparse box)
  const int swz = opt.no_xcd_swizzle ? 0 : 1;       // (experiments build only)
  const bool pipe = !opt.spmv_no_pipe;
  const bool win_env = opt.spmv_win != 0;
  // returns the number of partial-sum sets (of four) the launch leaves behind partials[pbase * 4]
  const double *stopflag = nullptr;            // run-ahead CG loop only: kernels of an iteration return at once when scal[S_STOP] is set
  bool pass_roword = false;                    // set by the run-ahead loop while its vectors are in row order
  auto pass = [&](int mode, bool store, double2 *ra, double2 *rg, const int *rowlist = nullptr, int nrows = 0, int pbase = 0) -> int {
    win_used = false;
    if (win_valid && win_env && (!rowlist || rowlist == rows_int || rowlist == rows_bnd)) {   // window pass: the group's partners in LDS, 16-bit slots (k_spmv_win)
      win_used = true;
      const int *glist = !rowlist ? nullptr : (rowlist == rows_int ? win_gint : win_gbnd);    // multi-rank overlap: interior groups while the halo is in flight, then the rest
      const int ng = !rowlist ? win_groups : (rowlist == rows_int ? win_groups - win_nbnd : win_nbnd);
      if (ng == 0) return 0;
      const size_t lds = static_cast<size_t>(win_maxunits) * WIN_UNIT * sizeof(double2);
      const bool one_trip = !ff.pqeq && opt.spmv_one_trip != 0 && max_row10 > 256 && max_row10 <= 384;      // (PQEq: the third stream of 384 entries does not fit the 64 registers of two workgroups per CU)
      const bool pq_pre = ff.pqeq && opt.pq_prefetch != 0;
      st.spmv_nstep = pq_pre ? 1 : (one_trip ? 3 : 2); st.spmv_var = pq_pre ? WIN_PREFETCH : (one_trip ? WIN_LEAN : (WIN_PREFETCH | WIN_LEAN));   // what the line below dispatches (bench.py names the instance whose counters it quotes)
      // per-row operands of the tail: by atom, or (run-ahead loop in row order, MODE_HSH only) by the row's place in rows_sorted
      const int *p_n10 = pass_roword ? r_n10 : n10, *p_type = pass_roword ? r_type : type;
      const double2 *p_hst = pass_roword ? r_hst : hst; double2 *p_gst = pass_roword ? r_gst : gst;
#define RX_WIN3(M, S, P) do { if (P && pq_pre) k_spmv_win<M, S, P, 1, WIN_PREFETCH><<<ng, 64 * WIN_ROWS, lds, stream>>>(N, G, S10, dff, sl10, hess, p_n10, rows_sorted, win_k, win_cnt, xs, p_hst, p_gst, qst, q, p_type, scal, partials, ra, rg, hsc, pqrow, glist, ng, pbase, stopflag, win_flag, pass_roword ? 1 : 0); \
                              else if (one_trip && !P) k_spmv_win<M, S, false, 3, WIN_LEAN><<<ng, 64 * WIN_ROWS, lds, stream>>>(N, G, S10, dff, sl10, hess, p_n10, rows_sorted, win_k, win_cnt, xs, p_hst, p_gst, qst, q, p_type, scal, partials, ra, rg, hsc, pqrow, glist, ng, pbase, stopflag, win_flag, pass_roword ? 1 : 0); \
                              else k_spmv_win<M, S, P, 2, WIN_PREFETCH | WIN_LEAN><<<ng, 64 * WIN_ROWS, lds, stream>>>(N, G, S10, dff, sl10, hess, p_n10, rows_sorted, win_k, win_cnt, xs, p_hst, p_gst, qst, q, p_type, scal, partials, ra, rg, hsc, pqrow, glist, ng, pbase, stopflag, win_flag, pass_roword ? 1 : 0); } while (0)
#define RX_WIN(M, S) do { if (ff.pqeq) RX_WIN3(M, S, true); else RX_WIN3(M, S, false); } while (0)
      if (mode == MODE_HSH) { if (store) RX_WIN(MODE_HSH, true); else RX_WIN(MODE_HSH, false); }
      else { if (store) RX_WIN(MODE_GRAD, true); else RX_WIN(MODE_GRAD, false); }
#undef RX_WIN
#undef RX_WIN3
      return ng;
    }
    const int rbl = rowlist ? nblk(nrows, SPMV_WPB) : rb;
    if (rbl == 0) return 0;
    st.spmv_nstep = 0; st.spmv_var = 0;
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
  const bool est_with_update_ = !opt.est_separate;  // (experiments build only)
  // run-ahead loop (below): iteration 0 always happens -- the exit tests of qeq.F90:114-115 compare Est with GEst2 = 1e99 -- so the host does not
  // wait for Est of the start vector; it reads it (for the trace) when the first iteration's Est arrives, which is later in stream order
  const bool start_async = !multi() && cfg.qeq_mode == 1 && !ff.pqeq && est_with_update_ && !opt.cg_no_runahead && nmax >= 1;
  if (!start_async) sync_stream();
  double GEst2 = 1e99, Est = start_async ? 0.0 : h_scal[S_EST];
  est_trace.clear(); est_trace.push_back(Est);      // Est of the start vector, then of every iteration (debug tap 13: the reference's QEQDUMP trace)
  int it = 0;
  float ms = 0;
  bool xs_current = false;       // the fused direction kernel leaves the sorted copy of the new (hs,ht) in xs
  const bool overlap_on = !opt.no_halo_overlap;
  const bool est_with_update = !opt.est_separate, cg_scatter = !opt.cg_no_scatter;     // (experiments build only)
  const bool overlap = overlap_on && multi() && onepass && !rows_split_pending_invalid();
  bool halo_in_flight = false, q_pending = false;
  // ---- run-ahead loop (single rank, qeq_mode 1, plain QEq; RXMD_CG_NO_RUNAHEAD=1 switches it off) ------------------------------------
  // The host is one iteration BEHIND the device: iteration it is queued in full before the host has seen the Est that decides whether it
  // happens.  The decision (qeq.F90:114-115) is made on the device where Est becomes final (scalar_algebra stage 6 -> scal[S_STOP]); the
  // kernels of an iteration that is not to happen return at once; the host reads the same flag.  Iteration counts, charges and every
  // sum are those of the loop below -- what changes is that no host round trip lies between two iterations (on a host that shares its
  // cores with other jobs the blocking loop lost 90 us per iteration: 172 against 80 us of everything that is not the matrix pass).
  const bool runahead = !multi() && onepass && !ff.pqeq && est_with_update && !opt.cg_no_runahead;
  if (runahead) {
    auto exit_test = [&](double prev, double est) {
      return (0.5 * (std::fabs(prev) + std::fabs(est)) < cfg.QEq_tol) || (std::fabs(prev) > 0.0 && std::fabs(est / prev - 1.0) < cfg.QEq_tol);
    };
    // row order (above: k_rows_enter): the window pass must be the one that runs, and its rows are the places
    const bool roword = opt.cg_row_order != 0 && win_valid && win_env;
    const int R = win_groups * WIN_ROWS;
    auto enqueue = [&](int k) {                    // everything of iteration k; on the device a no-op when its stop flag is set
      stopflag = (k == 0) ? nullptr : scal + S_STOP + (k & 1);    // iteration 0 is decided by the host (Est of the start vector is here already)
      if (!xs_current) sorted_copy(hst);           // first iteration only: afterwards the direction kernel leaves the sorted copy behind
      xs_current = false;
      // the HIP event pair that times the pass rides on every n-th launch only (opt.pass_timing_every, default 8): an event between two dependent
      // kernels costs the chain ~7 us, two per iteration were 15-20 us of every CG iteration (profiles/r06_ab_pass_events.txt).  Read once the host has
      // confirmed that the iteration happened (a pass that returned at once is not timed).
      const bool timed = !opt.no_pass_events && (pass_counter++ % static_cast<unsigned long long>(std::max<long long>(opt.pass_timing_every, 1))) == 0;
      pass_timed_k[k & 1] = timed;
      if (timed) hipEventRecord(ev_pass[k & 1][0], stream);
      pass_roword = roword;
      const int np1 = pass(MODE_HSH, true, roword ? r_wall : wall, roword ? r_wgh : wgh);
      pass_roword = false;
      if (timed) hipEventRecord(ev_pass[k & 1][1], stream);
      // (round 6, measured and dropped: the pass finishing its own reduction -- chunks of 256 workgroups, the last to arrive adds its chunk, the last chunk
      //  adds the chunk sums and runs the stage-1 algebra.  The launch of k_reduce_fused goes away, 81 -> 71-74 us per iteration outside the pass, but
      //  every workgroup's last wavefront then waits for its four stores and a ticket before it frees the workgroup's LDS: the pass in the loop went from
      //  0.877-0.880 to 0.919-0.924 ms, +36 us per iteration net; profiles/r06_ab_reduce_in_pass.txt)
      reduce(1, np1);
      // Est and the stop flags of this iteration reach the host as a snapshot the update kernel's tail stores into pinned host memory (slot k & 1), its
      // sequence number last: nothing sits between the update and the direction kernel, and the host polls a word of its own memory
      snap_expect[k & 1] = static_cast<double>(++snap_seq);
      if (roword) k_cg_update<true><<<vb_upd, 256, 0, stream>>>(R, dff, scal, r_type, r_hst, r_qst, r_wall, r_wgh, r_sall, r_sgh, r_gst, partials, tickets + 1, pqrow, 6 | (((k + 1) & 1) << 4), stopflag,
                                                                h_scal + 64, snap_expect[k & 1], rows_sorted, N);
      else k_cg_update<true><<<vb_upd, 256, 0, stream>>>(N, dff, scal, type, hst, qst, wall, wgh, sall, sgh, gst, partials, tickets + 1, pqrow, 6 | (((k + 1) & 1) << 4), stopflag,
                                                         h_scal + 64, snap_expect[k & 1]);
      const bool scatter = cg_scatter && k + 1 <= nmax - 1;
      if (roword) { k_cg_direction<false><<<vb, 256, 0, stream>>>(R, dff, scal, r_type, r_gst, r_hst, r_hst2, r_qst, r_sall, r_sgh, q, partials, tickets + 2, pqrow, -1, G, invpos, g_rrow, scatter ? xs : nullptr, stopflag,
                                                                  rows_sorted, N, r_xpos); std::swap(r_hst, r_hst2); }
      else { k_cg_direction<false><<<vb, 256, 0, stream>>>(N, dff, scal, type, gst, hst, hst2, qst, sall, sgh, q, partials, tickets + 2, pqrow, -1, G, invpos, groot, scatter ? xs : nullptr, stopflag); std::swap(hst, hst2); }
      if (k + 1 <= nmax - 1) { if (!scatter) { if (roword) throw EngineError(RXMD_E_STATE, "row order needs the scatter form of the direction kernel"); sorted_copy(hst); } xs_current = true; }
    };
    it = 0;
    if (nmax >= 1 && (start_async || !exit_test(GEst2, Est))) {
      GEst2 = Est;
      if (roword) {                                // the start vectors (qs,qt), (hs,ht) = (gs,gt), the row sums of the start vector: into row order
        if (!cg_scatter) throw EngineError(RXMD_E_STATE, "row order needs the scatter form of the direction kernel");
        sorted_copy(hst); xs_current = true;       // (the first iteration's gather copy, from the atom-ordered start direction)
        k_rows_enter<<<nblk(R, 256), 256, 0, stream>>>(R, N, rows_sorted, type, n10, invpos, qst, hst, gst, sall, sgh, r_qst, r_hst, r_gst, r_sall, r_sgh, r_type, r_n10, r_xpos, rpos);
        if (G > N) k_ghost_rows<<<nblk(G - N, 256), 256, 0, stream>>>(N, G, groot, rpos, g_rrow);
        rows_live = true;
      }
      enqueue(0);
      for (it = 1;; ++it) {
        const bool queued = it <= nmax - 1;
        if (queued) enqueue(it);                   // ahead of the decision
        wait_snapshot((it - 1) & 1, snap_expect[(it - 1) & 1]);     // iteration it - 1 has produced its Est and the decision about iteration it
        if (it == 1 && start_async) est_trace[0] = h_scal[S_EST];      // (the start vector's Est: copied before anything of iteration 0 ran)
        collect_timers();
        const double *hs = h_scal + 64 + 64 * ((it - 1) & 1);
        Est = hs[S_EST];
        est_trace.push_back(Est);
        if (pass_timed_k[(it - 1) & 1]) { float pms = 0; if (hipEventElapsedTime(&pms, ev_pass[(it - 1) & 1][0], ev_pass[(it - 1) & 1][1]) == hipSuccess) { pass_timed_ms += pms; pass_timed_n += 1; } }
        st.spmv_launches += 1;
        if (!queued) break;                        // NMAXQEq iterations done
        if (hs[S_STOP + (it & 1)] != 0.0) { if (roword) std::swap(r_hst, r_hst2); else std::swap(hst, hst2); st.spmv_noop_launches += 1; break; }       // iteration it did not happen: its direction kernel wrote nothing, undo the swap
        GEst2 = Est;
      }
      stopflag = nullptr;
      if (roword) k_rows_exit<<<nblk(R, 256), 256, 0, stream>>>(R, N, rows_sorted, scal, r_qst, qst, q);      // back to atom order: (qs,qt) and q = qs - mu qt
      else k_apply_q<<<nblk(N, 256), 256, 0, stream>>>(N, scal, qst, q);     // q = qs - mu qt with the mu of the last iteration that happened (qeq.F90:150)
    }
    if (ff.pqeq) pqeq_update_shells();
    nstep_qeq = it; last_est = Est;
    st.qeq_iters_last = it; st.qeq_iters_total += it; st.qeq_calls += 1; qeq_iters_smooth = qeq_iters_smooth < 0.0 ? it : 0.75 * qeq_iters_smooth + 0.25 * it;
    // No host wait here (round 5): the kernels of the iteration that did not happen are still in the queue (they return at once and store no snapshot)
    // and FORCE queues behind them in stream order.  (Until round 5 a sync here cost ~50 us of idle GPU per step.)
    outer_end(t_qeq);
    if (!place_tuned && win_used && it >= 1) { sync_stream(); collect_timers(); tune_window_placement(); }
    return;
  }
  for (it = 0; it <= nmax - 1; ++it) {
    if (0.5 * (std::fabs(GEst2) + std::fabs(Est)) < cfg.QEq_tol) break;                          // qeq.F90:114
    if (std::fabs(GEst2) > 0.0 && std::fabs(Est / GEst2 - 1.0) < cfg.QEq_tol) break;            // qeq.F90:115
    GEst2 = Est;
    // (the event pair around the pass on every n-th iteration only, as in the run-ahead loop: an event costs the dependent chain ~7 us)
    const bool timed_it = onepass ? (pass_counter++ % static_cast<unsigned long long>(std::max<long long>(opt.pass_timing_every, 1))) == 0 : true;
    if (halo_in_flight) {
      // multi-rank overlap: the (hs,ht) halo of this iteration was started on the second stream right after the direction update;
      // rows without a ghost partner do not need it
      const int n_int = N - n_bnd;
      if (timed_it) hipEventRecord(ev[2], stream);
      const int np_int = pass(MODE_HSH, onepass, wall, wgh, rows_int, n_int, 0);
      join_comm_stream();
      const int np_bnd = pass(MODE_HSH, onepass, wall, wgh, rows_bnd, n_bnd, np_int);
      if (timed_it) hipEventRecord(ev[3], stream);
      reduce(1, np_int + np_bnd);
      halo_in_flight = false;
    } else {
      if (!xs_current) sorted_copy(hst);                                                         // QCOPY2, qeq.F90:93,164
      xs_current = false;
      if (timed_it) hipEventRecord(ev[2], stream);
      const int np1 = pass(MODE_HSH, onepass, onepass ? wall : nullptr, onepass ? wgh : nullptr);
      if (timed_it) hipEventRecord(ev[3], stream);
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
      if (est3) k_cg_direction<false><<<vb, 256, 0, stream>>>(N, dff, scal, type, gst, hst, hst2, qst, sall, sgh, q, partials, tickets + 2, pqrow, -1, G, invpos, groot, scatter ? xs : nullptr, nullptr);
      else k_cg_direction<true><<<vb_upd, 256, 0, stream>>>(N, dff, scal, type, gst, hst, hst2, qst, sall, sgh, q, partials, tickets + 2, pqrow, fuse ? 5 : 0, G, invpos, groot, scatter ? xs : nullptr, nullptr);
      q_pending = est3;
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
      est_trace.push_back(Est);
      if (timed_it) { hipEventElapsedTime(&ms, ev[2], ev[3]); pass_timed_ms += ms; pass_timed_n += 1; }
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
    est_trace.push_back(Est);
    hipEventElapsedTime(&ms, ev[2], ev[3]); pass_timed_ms += ms;
    hipEventElapsedTime(&ms, ev[4], ev[5]); pass_timed_ms += ms;
    pass_timed_n += 2;
    st.spmv_launches += 2;
  }
  if (halo_in_flight) join_comm_stream();             // the loop ended while a halo it will not use was still in flight
  if (q_pending) k_apply_q<<<nblk(N, 256), 256, 0, stream>>>(N, scal, qst, q);     // (the direction kernel without Est leaves q to the end)
  if (ff.pqeq) pqeq_update_shells();                  // pqeq.F90:169
  nstep_qeq = it; last_est = Est;
  st.qeq_iters_last = it; st.qeq_iters_total += it; st.qeq_calls += 1; qeq_iters_smooth = qeq_iters_smooth < 0.0 ? it : 0.75 * qeq_iters_smooth + 0.25 * it;
  outer_end(t_qeq);
  if (!place_tuned && win_used && it >= 1) { sync_stream(); collect_timers(); tune_window_placement(); }
}

}  // namespace rxmd
