// lists.hip -- bonded (3 A) neighbour list with reverse index, and the 10 A pair list with the QEq
// "hessian" (shielded-Coulomb matrix) built in the same sweep.
//   NEIGHBORLIST            reference src/main.F90:321-417   -> k_bonded_list + k_reverse_index
//   qeq_initialize          reference src/qeq.F90:183-268    \  one sweep: k_list10
//   GetNonbondingPairList   reference src/main.F90:420-477   /  (the reference walks the stencil twice)
// Candidates come from the engine's own cell grid (cell edge >= max(5 A, maxrc)), cell-sorted with z
// fastest, so a (dx,dy) column of the stencil is ONE contiguous range of the sorted array.
#include "engine.h"

#include <hipcub/hipcub.hpp>

namespace rxmd {

static inline int nblk(long long n, int b) { return n > 0 ? static_cast<int>((n + b - 1) / b) : 1; }   // an empty rank still launches (kernels guard their range)

// w component of sorted_xyzi: low 32 bits atom index, bits 32.. type
__global__ void k_pack_type(int G, const int *perm, const int *type, double4 *s) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= G) return;
  const int i = perm[k];
  const long long w = (static_cast<long long>(type[i]) << 32) | static_cast<unsigned int>(i);
  s[k].w = __longlong_as_double(w);
}

__global__ void __launch_bounds__(256) k_bonded_list(int G, int NB, int MAXNB, Grid g, DevFF ff, const int *__restrict__ cellid, const int *__restrict__ cellstart,
                                                      const double4 *__restrict__ sorted, const double *__restrict__ x, const double *__restrict__ y,
                                                      const double *__restrict__ z, const int *__restrict__ type, int *__restrict__ nbr, int *__restrict__ nbrcnt, int *err) {
  // squared bond cut-off of every type pair in LDS (0 = the pair has no bond row): one LDS read per candidate instead of two
  // dependent global look-ups (inxn2, then bond[inxn].rc2)
  __shared__ double s_rc2[256];
  for (int t = threadIdx.x; t < ff.n1 * ff.n1 && t < 256; t += blockDim.x) { const int ix = ff.inxn2[t]; s_rc2[t] = ix ? ff.bond[ix].rc2 : 0.0; }
  __syncthreads();
  const int i = xcd_swizzle(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  if (i >= G) return;
  const int c = cellid[i];
  const int cz = c % g.n[2], cy = (c / g.n[2]) % g.n[1], cx = c / (g.n[2] * g.n[1]);
  const double xi = x[i], yi = y[i], zi = z[i];
  const int ti = type[i];
  const double *rc2row = s_rc2 + ti * ff.n1;
  const int z0 = max(cz - 1, 0), z1 = min(cz + 1, g.n[2] - 1);
  int cnt = 0;
  for (int dx = -1; dx <= 1; ++dx) {
    const int x2 = cx + dx;
    if (x2 < 0 || x2 >= g.n[0]) continue;
    for (int dy = -1; dy <= 1; ++dy) {
      const int y2 = cy + dy;
      if (y2 < 0 || y2 >= g.n[1]) continue;
      const int cb = (x2 * g.n[1] + y2) * g.n[2];
      const int k0 = cellstart[cb + z0], k1 = cellstart[cb + z1 + 1];
      for (int k = k0; k < k1; ++k) {
        const double4 p = sorted[k];
        const long long w = __double_as_longlong(p.w);
        const int j = static_cast<int>(w & 0xffffffffLL);
        if (j == i) continue;
        const int tj = static_cast<int>(w >> 32);
        const double d0 = p.x - xi, d1 = p.y - yi, d2 = p.z - zi;
        const double r2 = d0 * d0 + d1 * d1 + d2 * d2;
        if (r2 < rc2row[tj]) {                  // dr2 < rc2(inxn), main.F90:366 (no bond row: cut-off 0)
          if (cnt < MAXNB) nbr[static_cast<size_t>(cnt) * NB + i] = j;
          ++cnt;
        }
      }
    }
  }
  if (cnt > MAXNB) { atomicMax(&err[1], cnt); atomicCAS(&err[0], DERR_NONE, DERR_MAXNB); cnt = MAXNB; }  // main.F90:402-407
  nbrcnt[i] = cnt;
}

// nbrindx(i,i1) = j1 such that nbrlist(j,j1) == i   (main.F90:383-399)
__global__ void k_reverse_index(int G, int NB, const int *__restrict__ nbr, const int *__restrict__ nbrcnt, unsigned char *__restrict__ nbrindx, int *err) {
  const int i = xcd_swizzle(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  if (i >= G) return;
  const int ni = nbrcnt[i];
  for (int s = 0; s < ni; ++s) {
    const int j = nbr[static_cast<size_t>(s) * NB + i];
    const int nj = nbrcnt[j];
    int found = -1;
    for (int t = 0; t < nj; ++t)
      if (nbr[static_cast<size_t>(t) * NB + j] == i) found = t;
    if (found < 0) { atomicCAS(&err[0], DERR_NONE, DERR_NBRINDX); found = 0; }
    nbrindx[static_cast<size_t>(s) * NB + i] = static_cast<unsigned char>(found);
  }
}

// get_coulomb_and_dcoulomb_pqeq (reference src/module.F90:401-418): energy kernel and (1/r) dE/dr at squared distance r2.
// Beyond the taper cutoff the reference returns without touching its outputs (callers then see the previous pair's
// values); here such a lookup contributes nothing -- see DESIGN.md "PQEq beyond-cutoff lookups".
__device__ inline bool pq_lookup(const DevFF &ff, const double4 *__restrict__ tab, int row, double r2, double &E, double &F) {
  if (r2 > ff.rctap2) { E = 0.0; F = 0.0; return false; }
  const int itb = static_cast<int>(r2 * ff.UDRi);
  double t = r2 - itb * ff.UDR;
  t = t * ff.UDRi;
  const double4 nd = tab[static_cast<size_t>(row) * (NTABLE + 2) + itb];
  E = nd.x + t * nd.y; F = nd.z + t * nd.w;
  return true;
}
__device__ inline double wave_sum_l(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// One wavefront per resident row.  Lanes sweep the candidates of a stencil column 64 at a time,
// the accepted ones are compacted with a ballot so that a row is written as contiguous runs
// (coalesced 8-byte + 4-byte streams) in a deterministic order.
// A list entry names the partner by its CELL-SORTED position (the loop variable of this sweep), not by atom index:
// the consumers (QEq matrix passes, ENbond, Ehb) gather from cell-sorted copies, so the 64 lanes of a wavefront hit a
// handful of cache lines instead of 64 scattered ones.  Bits: see NB10_* in engine.h.
// PQ: PQEq variant of qeq_initialize (pqeq.F90:262-353): core-core hessian from the pcc table, the shell-core matrix hsc of
// get_hsh's Csicj term, and per row (fpqeq Eq. 30, sum_j H Z_j, sum_j hsc Z_j, shell-shell energy) -> pqrow
template <bool SELFCHECK, bool PQ>
__global__ void __launch_bounds__(256) k_list10(int N, int S10, Grid g, DevFF ff, const int *__restrict__ cellid, const int *__restrict__ cellstart,
                                                 const double4 *__restrict__ sorted, const double *__restrict__ x, const double *__restrict__ y,
                                                 const double *__restrict__ z, const int *__restrict__ type, const long long *__restrict__ gid,
                                                 int *__restrict__ nb10, unsigned short *__restrict__ nb10s, double *__restrict__ hess, int *__restrict__ n10, int *err,
                                                 const double4 *__restrict__ sorted_shl, const double *__restrict__ shx, const double *__restrict__ shy, const double *__restrict__ shz,
                                                 double *__restrict__ hsc, double4 *__restrict__ pqrow,
                                                 const double2 *__restrict__ xs0, double2 *__restrict__ s_all, double2 *__restrict__ s_gh, int *__restrict__ rowflag) {
  // dynamic LDS: [4][128] queue of accepted candidates (sorted position, neighbourhood position), [4][128] chunk table, then [4][S10] 16-bit rows
  extern __shared__ int lds_all[];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));   // wave-uniform -> the row's constants live in scalar registers
  const int i = xcd_swizzle(blockIdx.x, gridDim.x) * (blockDim.x >> 6) + w;
  if (i >= N) return;
  constexpr int MAXCH = 128;                     // chunks of one row (RDX: 26-50)
  int2 *sq = reinterpret_cast<int2 *>(lds_all) + w * 128;
  int4 *ck = reinterpret_cast<int4 *>(lds_all + 4 * 128 * 2) + w * MAXCH;
  unsigned short *srow = reinterpret_cast<unsigned short *>(lds_all + 4 * 128 * 2 + 4 * MAXCH * 4) + static_cast<size_t>(w) * S10;
  int nchunk = 0;
  const int c = cellid[i];
  const int cz = c % g.n[2], cy = (c / g.n[2]) % g.n[1], cx = c / (g.n[2] * g.n[1]);
  const double xi = x[i], yi = y[i], zi = z[i];
  const int ti = type[i];
  const int z0 = max(cz - 2, 0), z1 = min(cz + 2, g.n[2] - 1);
  const size_t row = static_cast<size_t>(i) * S10;
  double sxi = 0.0, syi = 0.0, szi = 0.0, Zi = 0.0, p_f = 0.0, p_hz = 0.0, p_bz = 0.0, p_ss = 0.0;
  if (PQ) { sxi = shx[i]; syi = shy[i]; szi = shz[i]; Zi = ff.Zpq[ti]; }
  // xs0 != nullptr: the sweep also forms the row sums H.(qs,qt) of the CG start vector (qt = 0) -- the matrix pass that
  // get_gradient would need before the first iteration (qeq.F90:87) comes for free while the entries are in registers
  double ra = 0.0, rg = 0.0;
  bool anyghost = false;                          // does the row have a ghost partner (boundary row of the domain)?
  int cnt = 0;      // entries written so far
  int qn = 0;       // accepted candidates waiting in the queue
  int loff = 0;     // candidates in the stencil columns already swept = position of this column inside the neighbourhood

  // Phase 2, dense: one queued candidate per lane -> table interpolation, list entry, hessian value.  Only ~27 % of the
  // candidates of a chunk pass the distance test, so doing this work on compacted batches keeps every lane busy.
  auto emit = [&](int nproc) {
    if (lane < nproc) {
      const int2 qe = sq[lane];
      const int k = qe.x, slot = cnt + lane;
      if (slot < S10) {
        const double4 p = sorted[k];
        const long long wv = __double_as_longlong(p.w);
        const int j = static_cast<int>(wv & 0xffffffffLL), tj = static_cast<int>(wv >> 32);
        const double d0 = xi - p.x, d1 = yi - p.y, d2 = zi - p.z;
        const double r2 = d0 * d0 + d1 * d1 + d2 * d2;
        // hessian entry as qeq_initialize computes it: r^2 rounded to REAL(4) first (qeq.F90:191,222-240)
        const float r2f = static_cast<float>(r2);
        double h = 0.0, hc = 0.0;
        const int inxn = ff.inxn2[ti * ff.n1 + tj];
        if (PQ) {
          if (static_cast<double>(r2f) < ff.rctap2) {                  // the pair is in PQEq's own list (real(4) test, pqeq.F90:305)
            const double C0q = 14.4;                                   // Cclmb0_qeq, module.F90:682
            const double4 sj = sorted_shl[k];
            const double Zj = ff.Zpq[tj];
            const int prow = ff.inxnpq[ti * ff.npq1 + tj];
            double E, F;
            pq_lookup(ff, ff.tabPcc, prow, r2, E, F);                  // core(i)-core(j)
            h = C0q * E;
            p_hz += h * Zj;
            // Eq. 30: field of core(j) minus field of shell(j) at core(i); table row (jty,ity), pqeq.F90:328-334
            double e0 = d0 - sj.x, e1 = d1 - sj.y, e2 = d2 - sj.z;
            pq_lookup(ff, ff.tabPsc, prow, e0 * e0 + e1 * e1 + e2 * e2, E, F);
            p_f += h * Zj - C0q * E * Zj;
            // shell(i)-core(j): Csicj = -hsc * (q_j + Z_j), pqeq.F90:392-395
            e0 = d0 + sxi; e1 = d1 + syi; e2 = d2 + szi;
            pq_lookup(ff, ff.tabPsc, prow, e0 * e0 + e1 * e1 + e2 * e2, E, F);
            hc = C0q * E * Zi;
            p_bz += hc * Zj;
            // shell(i)-shell(j): Csisj, pqeq.F90:397-401 (half of it per row, :409)
            e0 -= sj.x; e1 -= sj.y; e2 -= sj.z;
            pq_lookup(ff, ff.tabPss, prow, e0 * e0 + e1 * e1 + e2 * e2, E, F);
            p_ss += 0.5 * C0q * E * Zi * Zj;
          }
          hsc[row + slot] = hc;
        } else if (static_cast<double>(r2f) < ff.rctap2 && inxn != 0) {
          const int itb = static_cast<int>(static_cast<double>(r2f) * ff.UDRi);
          double drtb = static_cast<double>(r2f) - itb * ff.UDR;
          drtb = drtb * ff.UDRi;
          const double *T = ff.tabQEq + static_cast<size_t>(inxn) * (NTABLE + 2);
          h = (1.0 - drtb) * T[itb] + drtb * T[itb + 1];
        }
        unsigned ent = static_cast<unsigned>(k) | (static_cast<unsigned>(tj) << NB10_IDX_BITS) | (j >= N ? NB10_GHOST : 0u);
        if (SELFCHECK && gid[j] == gid[i]) ent |= NB10_SELF;           // an atom and its own periodic image (small boxes only)
        if (xs0) {
          const double qsj = xs0[k].x;
          ra += h * qsj;
          if (PQ) rg += hc * qsj; else if (j >= N) rg += h * qsj;
        }
        anyghost |= (j >= N);
        nb10[row + slot] = static_cast<int>(ent);
        if (nb10s) srow[slot] = static_cast<unsigned short>((qe.y & 0x7fff) | (j >= N ? 0x8000 : 0));
        hess[row + slot] = h;
      }
    }
    cnt += nproc;
  };

  // chunk table of the row: the 25 stencil columns (contiguous runs of the sorted array) cut into 64-candidate chunks, in sweep
  // order.  Lane t < 25 owns column t; an exclusive scan over the lanes places its chunks.  The sweep is bound by the latency of
  // the candidate loads (26 dependent round trips per row when done one chunk at a time), so four chunks are loaded at once.
  {
    int k0 = 0, len = 0;
    if (lane < 25) {
      const int x2 = cx + lane / 5 - 2, y2 = cy + lane % 5 - 2;
      if (x2 >= 0 && x2 < g.n[0] && y2 >= 0 && y2 < g.n[1]) {
        const int cb = (x2 * g.n[1] + y2) * g.n[2];
        k0 = cellstart[cb + z0];
        len = cellstart[cb + z1 + 1] - k0;
      }
    }
    int nch = (len + 63) >> 6, cpre = nch, lpre = len;      // inclusive scans over lanes 0..24 (chunks, candidates)
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
      const int c2 = __shfl_up(cpre, o, 64), l2 = __shfl_up(lpre, o, 64);
      if (lane >= o) { cpre += c2; lpre += l2; }
    }
    const int cfirst = cpre - nch, lfirst = lpre - len;
    for (int t = 0; t < nch && cfirst + t < MAXCH; ++t) ck[cfirst + t] = make_int4(k0 + 64 * t, k0 + len, lfirst - k0, 0);
    loff = __shfl(lpre, 24, 64);
    nchunk = __shfl(cpre, 24, 64);
    if (nchunk > MAXCH) { if (lane == 0) atomicCAS(&err[0], DERR_NONE, DERR_GRID); nchunk = MAXCH; }   // > 8192 candidates around one atom
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  for (int c0 = 0; c0 < nchunk; c0 += 4) {
    // Phase 1, sparse: distance test of 4 x 64 candidates (all loads first), survivors appended to the queue in candidate order
    double4 p[4];
    int kk[4], lb[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      ok[u] = false; kk[u] = 0; lb[u] = 0;
      if (c0 + u < nchunk) {
        const int4 ce = ck[c0 + u];
        kk[u] = ce.x + lane; lb[u] = ce.z;
        ok[u] = kk[u] < ce.y;
      }
      p[u] = ok[u] ? sorted[kk[u]] : make_double4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      bool in = false;
      if (ok[u]) {
        const int j = static_cast<int>(__double_as_longlong(p[u].w) & 0xffffffffLL);
        const double d0 = xi - p[u].x, d1 = yi - p[u].y, d2 = zi - p[u].z;
        const double r2 = d0 * d0 + d1 * d1 + d2 * d2;
        in = (j != i) && (r2 <= ff.rctap2);       // dr2 <= rctap2, main.F90:458
      }
      const unsigned long long m = __ballot(in);
      if (in) sq[qn + __popcll(m & ((1ULL << lane) - 1ULL))] = make_int2(kk[u], lb[u] + kk[u]);
      qn += __popcll(m);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      if (qn >= 64) {
        emit(64);
        const int rest = qn - 64;
        int2 v = make_int2(0, 0);
        if (lane < rest) v = sq[64 + lane];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (lane < rest) sq[lane] = v;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        qn = rest;
      }
    }
  }
  if (qn > 0) emit(qn);
  if (cnt > S10) { if (lane == 0) { atomicMax(&err[1], cnt); atomicCAS(&err[0], DERR_NONE, DERR_MAXN10); } cnt = S10; }  // qeq.F90:248-252
  if (lane < 4 && cnt + lane < ((cnt + 3) & ~3)) { nb10[row + cnt + lane] = 0; if (nb10s) srow[cnt + lane] = 0; hess[row + cnt + lane] = 0.0; }   // zero-pad the row to a multiple of 4
  if (nb10s) {
    const int nw = ((cnt + 3) & ~3) >> 1;            // the wavefront's own LDS row: no barrier needed beyond the wave's program order
    const unsigned *sw = reinterpret_cast<const unsigned *>(srow);
    unsigned *dw = reinterpret_cast<unsigned *>(nb10s + row);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    for (int t = lane; t < nw; t += 64) dw[t] = sw[t];
  }
  if (xs0) {
    ra = wave_sum_l(ra); rg = wave_sum_l(rg);
    if (lane == 0) { s_all[i] = make_double2(ra, 0.0); s_gh[i] = make_double2(rg, 0.0); }
  }
  if (PQ) {
    p_f = wave_sum_l(p_f); p_hz = wave_sum_l(p_hz); p_bz = wave_sum_l(p_bz); p_ss = wave_sum_l(p_ss);
    if (lane < 4 && cnt + lane < ((cnt + 3) & ~3)) hsc[row + cnt + lane] = 0.0;
    if (lane == 0) pqrow[i] = make_double4(p_f, p_hz, p_bz, p_ss);
  }
  { const unsigned long long mg = __ballot(anyghost); if (lane == 0 && rowflag) rowflag[i] = (mg != 0ULL) ? 1 : 0; }
  if (lane == 0) { n10[i] = cnt; if (loff > __hip_atomic_load(&err[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&err[2], loff); }
}

// boundary rows keep their order, interior rows too: index lists for the two launches of the matrix pass
__global__ void k_split_rows(int N, const int *__restrict__ flag, const int *__restrict__ scan, int *__restrict__ rows_int, int *__restrict__ rows_bnd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  if (flag[i]) rows_bnd[scan[i]] = i; else rows_int[i - scan[i]] = i;
}

void Engine::build_bonded_list() {
  k_pack_type<<<nblk(G, 256), 256, 0, stream>>>(G, perm, type, sorted_xyzi);
  k_bonded_list<<<nblk(G, 256), 256, 0, stream>>>(G, NB, MAXNB, grid, dff, cellid, cellstart, sorted_xyzi, pos[0], pos[1], pos[2], type, nbr, nbrcnt, d_err);
  k_reverse_index<<<nblk(G, 256), 256, 0, stream>>>(G, NB, nbr, nbrcnt, nbrindx, d_err);
}

void Engine::build_list10() {
  // an atom can meet its own image within rctap only if some box edge is shorter than 2*rctap
  RX_HIP(hipMemsetAsync(d_err + 2, 0, sizeof(int), stream));
  const bool selfcheck = (box.lat[0] < 2.0 * ff.rctap + 1.0) || (box.lat[1] < 2.0 * ff.rctap + 1.0) || (box.lat[2] < 2.0 * ff.rctap + 1.0);
  const size_t lds = 4 * 128 * sizeof(int2) + 4 * 128 * sizeof(int4) + static_cast<size_t>(S10) * 4 * sizeof(unsigned short);
#define RX_LIST10(SC, PQF)                                                                                                                     \
  k_list10<SC, PQF><<<nblk(N, 4), 256, lds, stream>>>(N, S10, grid, dff, cellid, cellstart, sorted_xyzi, pos[0], pos[1], pos[2], type, gid, nb10, nb10s, \
                                                      hess, n10, d_err, sorted_shl, shl[0], shl[1], shl[2], hsc, pqrow, \
                                                      sums_from_list ? xs : nullptr, sall, sgh, multi() ? flags : nullptr)
  if (ff.pqeq) { if (selfcheck) RX_LIST10(true, true); else RX_LIST10(false, true); }
  else { if (selfcheck) RX_LIST10(true, false); else RX_LIST10(false, false); }
#undef RX_LIST10
  if (multi()) {     // interior rows (no ghost partner) and boundary rows: the matrix pass does the former while the vector halo is in flight
    RX_HIP(hipMemsetAsync(flags + N, 0, sizeof(int), stream));
    size_t tb = cubtmp_bytes;
    RX_HIP(hipcub::DeviceScan::ExclusiveSum(cubtmp, tb, flags, scanout, N + 1, stream));
    k_split_rows<<<nblk(N, 256), 256, 0, stream>>>(N, flags, scanout, rows_int, rows_bnd);
    RX_HIP(hipMemcpyAsync(&n_bnd, scanout + N, sizeof(int), hipMemcpyDeviceToHost, stream));
    rows_split_pending = true;          // n_bnd is valid after the next stream synchronisation (check_device_error of the list build)
  }
}

}  // namespace rxmd
