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

namespace rxmd {

static inline int nblk(long long n, int b) { return n > 0 ? static_cast<int>((n + b - 1) / b) : 1; }   // an empty rank still launches (kernels guard their range)

enum { S_MU = 0, S_LMIN_S, S_LMIN_T, S_GOLD_S, S_GOLD_T, S_GNEW_S, S_GNEW_T, S_EST, S_GH_S, S_GH_T, S_HSH_S, S_HSH_T, S_SSUM, S_TSUM, S_BETA_S, S_BETA_T, S_RAW0, S_RAW1, S_RAW2, S_RAW3, S_RAW4, S_RAW5, S_RAW6, S_RAW7, S_COUNT };
enum { MODE_HSH = 0, MODE_GRAD = 1 };
typedef double f64x2 __attribute__((ext_vector_type(2)));
#ifndef SPMV_UNR
#define SPMV_UNR 4
#endif
constexpr int UNR = SPMV_UNR;   // 4 x 64 = 256 entries in flight per wavefront and pass of the row loop (see k_spmv)

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
                                               const int *__restrict__ rowlist, int nrows, int pbase) {
  // rowlist != nullptr: this launch covers nrows rows named by the list (interior or boundary rows of a multi-rank domain);
  // its workgroups write their partial sums behind the pbase workgroups of the other launch
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  // the row is the same for the 64 lanes: say so (readfirstlane), and the row's base addresses, its length and the loop bounds live in
  // scalar registers -- 44 instead of 57 VGPRs and scalar address arithmetic: 1.03-1.09 -> 0.95 ms per pass on the same box
  const int widx = (swz ? xcd_swizzle(blockIdx.x, gridDim.x) : static_cast<int>(blockIdx.x)) * wpb + __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int row = rowlist ? (widx < nrows ? rowlist[widx] : N) : widx;
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
    if (PIPE) request(0, S10, e, h, c);
    const int n = n10[row];
    // operands of the row tail, requested before the streams so that they are not a further dependent round trip after the reduction
    const int pf_t = type[row];
    const double2 pf_a = (MODE == MODE_HSH) ? hst[row] : qst[row];
    const double2 pf_b = (MODE == MODE_HSH) ? gst[row] : make_double2(q[row], 0.0);
    const double mu = (MODE == MODE_GRAD) ? scal[S_MU] : 0.0;
    double as = 0.0, at = 0.0, gs_ = 0.0, gt_ = 0.0;
    if (PIPE) {
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
    if (lane == 0) {
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
  block_store_partials<4>(acc, partials + static_cast<size_t>(pbase) * 4, 4);
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
    if (stage == 6) scal[S_EST] = scal[S_RAW4] - mu * scal[S_RAW5] + mu * mu * scal[S_RAW6];
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
__global__ void __launch_bounds__(256) k_reduce_fused(int nblocks, const double *__restrict__ partials, double *__restrict__ lvl1, unsigned *ticket, int stage, double *__restrict__ scal) {
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
                                                    double2 *__restrict__ gst, double *__restrict__ partials, unsigned *ticket, const double4 *__restrict__ pqrow, int stage) {
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
                                                       double *__restrict__ partials, unsigned *ticket, const double4 *__restrict__ pqrow, int stage) {
  const double mu = scal[S_MU], b1 = scal[S_BETA_S], b2 = scal[S_BETA_T];
  double es = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
    const double2 g = gst[i], h = hst[i];
    hst_new[i] = make_double2(g.x + b1 * h.x, g.y + b2 * h.y);
    const DevAtomP ap = ff.atom[type[i]];
    const double2 qv = qst[i], a = sall[i], gh = sgh[i];
    const double qi = qv.x - mu * qv.y;
    q[i] = qi;
    const double hq_all = a.x - mu * a.y, hq_res = (a.x - gh.x) - mu * (a.y - gh.y);
    if (pqrow) es += pq_est_row(ap, ff.Zpq[type[i]], pqrow[i], qi, hq_all, gh.x - mu * gh.y);
    else es += ap.chi * qi + 0.5 * ap.eta * qi * qi + 0.5 * qi * (hq_all + hq_res);
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
  if (nprocs == 1 && !nccl) return;              // forced staged mode of a single rank without a communicator: nothing to add
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
  const int nred = rb;                                                             // partials one matrix pass leaves
  double *lvl1 = partials + partials_cap;      // 128 x 4 first-level sums live behind the per-workgroup partials (fixed offset: independent of the cell count of a sparse box)
  static const int swz = std::getenv("RXMD_NO_XCD_SWIZZLE") ? 0 : 1;
  const bool pipe = (std::getenv("RXMD_SPMV_NO_PIPE") == nullptr);        // read per call: the tests switch it
  auto pass = [&](int mode, bool store, double2 *ra, double2 *rg, const int *rowlist = nullptr, int nrows = 0, int pbase = 0) {
    const int rbl = rowlist ? nblk(nrows, SPMV_WPB) : rb;
    if (rbl == 0) return;
#define RX_PASS3(M, S, P, PI) k_spmv<M, S, P, PI><<<rbl, 64 * SPMV_WPB, 0, stream>>>(N, S10, dff, nb10, hess, n10, xs, hst, gst, qst, q, type, scal, partials, ra, rg, hsc, pqrow, swz, rowlist, nrows, pbase)
#define RX_PASS(M, S)                                                                                                  \
  do {                                                                                                                 \
    if (ff.pqeq) { if (pipe) RX_PASS3(M, S, true, 1); else RX_PASS3(M, S, true, 0); }                                  \
    else { if (pipe) RX_PASS3(M, S, false, 1); else RX_PASS3(M, S, false, 0); }                                        \
  } while (0)
    if (mode == MODE_HSH) { if (store) RX_PASS(MODE_HSH, true); else RX_PASS(MODE_HSH, false); }
    else { if (store) RX_PASS(MODE_GRAD, true); else RX_PASS(MODE_GRAD, false); }
#undef RX_PASS
#undef RX_PASS3
  };
  auto reduce = [&](int stage, int nb_) {
    if (!multi()) {                              // single rank: level-1 sums, final sum and scalar algebra in one launch
      if (nb_ > 1024) k_reduce_fused<<<128, 256, 0, stream>>>(nb_, partials, lvl1, tickets, stage, scal);
      else k_reduce_fused<<<1, 256, 0, stream>>>(nb_, partials, lvl1, tickets, stage, scal);
      return;
    }
    if (nb_ > 1024) k_reduce_fused<<<128, 256, 0, stream>>>(nb_, partials, lvl1, tickets, 0, scal);   // stage 0: rank-local sums only
    else k_reduce_fused<<<1, 256, 0, stream>>>(nb_, partials, lvl1, tickets, 0, scal);
    allreduce_scal4();                           // MPI_ALLREDUCE of the rank-local sums
    k_scalar_algebra<<<1, 64, 0, stream>>>(stage, scal);
  };
  RX_HIP(hipMemsetAsync(scal, 0, sizeof(double) * 32, stream));
  const bool onepass = (cfg.qeq_mode == 1);
  if (sums_from_list) {
    k_grad_start<<<vb, 256, 0, stream>>>(N, dff, type, qst, q, sall, sgh, gst, partials, pqrow);
    reduce(3, vb);
  } else {
    qeq_start_vectors();
    pass(MODE_GRAD, onepass, onepass ? sall : nullptr, onepass ? sgh : nullptr);
    reduce(3, nred);
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
  const bool overlap = overlap_on && multi() && onepass && !rows_split_pending_invalid();
  bool halo_in_flight = false;
  for (it = 0; it <= nmax - 1; ++it) {
    if (0.5 * (std::fabs(GEst2) + std::fabs(Est)) < cfg.QEq_tol) break;                          // qeq.F90:114
    if (std::fabs(GEst2) > 0.0 && std::fabs(Est / GEst2 - 1.0) < cfg.QEq_tol) break;            // qeq.F90:115
    GEst2 = Est;
    if (halo_in_flight) {
      // multi-rank overlap: the (hs,ht) halo of this iteration was started on the second stream right after the direction update;
      // rows without a ghost partner do not need it
      const int n_int = N - n_bnd, nb_int = nblk(n_int, SPMV_WPB);
      hipEventRecord(ev[2], stream);
      pass(MODE_HSH, onepass, wall, wgh, rows_int, n_int, 0);
      join_comm_stream();
      pass(MODE_HSH, onepass, wall, wgh, rows_bnd, n_bnd, nb_int);
      hipEventRecord(ev[3], stream);
      reduce(1, nb_int + nblk(n_bnd, SPMV_WPB));
      halo_in_flight = false;
    } else {
      if (!xs_current) sorted_copy(hst);                                                         // QCOPY2, qeq.F90:93,164
      xs_current = false;
      hipEventRecord(ev[2], stream);
      pass(MODE_HSH, onepass, onepass ? wall : nullptr, onepass ? wgh : nullptr);
      hipEventRecord(ev[3], stream);
      reduce(1, nred);
    }
    if (onepass) {       // qeq_mode 1: one matrix pass per iteration; gradient and Est by recurrence on the stored row sums
      const bool fuse = !multi();                  // single rank: every reduction finishes in-kernel; multi: sums, all-reduce, algebra
      // plain QEq: Est rides on the update kernel's sums (quadratic in mu, scalar_algebra stage 6) -- multi-rank: two all-reduces per
      // iteration instead of three; any rank count: Est is final BEFORE the direction kernel, so its copy to the host, the host's exit
      // test and the launch of the next matrix pass all run underneath the direction update and the sorted copy / halo
      const bool est3 = !ff.pqeq && est_with_update;
      if (est3) k_cg_update<true><<<vb_upd, 256, 0, stream>>>(N, dff, scal, type, hst, qst, wall, wgh, sall, sgh, gst, partials, tickets + 1, pqrow, fuse ? 6 : 0);
      else k_cg_update<false><<<vb_upd, 256, 0, stream>>>(N, dff, scal, type, hst, qst, wall, wgh, sall, sgh, gst, partials, tickets + 1, pqrow, fuse ? 4 : 0);
      if (!fuse) { allreduce_scal4(est3 ? 8 : 4); k_scalar_algebra<<<1, 64, 0, stream>>>(est3 ? 6 : 4, scal); }
      if (est3) { RX_HIP(hipMemcpyAsync(h_scal, scal, sizeof(double) * S_COUNT, hipMemcpyDeviceToHost, stream)); RX_HIP(hipEventRecord(ev_est, stream)); }
      // the direction update runs over the residents in atom order (every access coalesced); the cell-sorted copy with the images
      // is one gather pass queued behind it (sorted_copy).  Doing both in one kernel over the sorted positions
      // (five random 16-byte accesses per atom) was 0.4 ms per step slower.
      k_cg_direction<<<est3 ? vb : vb_upd, 256, 0, stream>>>(N, dff, scal, type, gst, hst, hst2, qst, sall, sgh, q, partials, tickets + 2, pqrow, est3 ? -1 : (fuse ? 5 : 0));
      if (!fuse && !est3) { allreduce_scal4(); k_scalar_algebra<<<1, 64, 0, stream>>>(5, scal); }
      if (!est3) {       // PQEq: Est comes out of the direction kernel; the host still waits for this copy only, not for the sorted copy behind it
        RX_HIP(hipMemcpyAsync(h_scal, scal, sizeof(double) * S_COUNT, hipMemcpyDeviceToHost, stream));
        RX_HIP(hipEventRecord(ev_est, stream));
      }
      std::swap(hst, hst2);
      xs_current = false;
      if (!overlap && it + 1 <= nmax - 1) {        // sorted copy (multi-rank: after the (hs,ht) halo) queued before the host waits for Est
        sorted_copy(hst);
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
    pass(MODE_GRAD, false, nullptr, nullptr);
    hipEventRecord(ev[5], stream);
    reduce(3, nred);
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
