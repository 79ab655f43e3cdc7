// nonbonded.hip -- ENbond (reference src/pot.F90:676-781): tabulated van der Waals + shielded Coulomb
// over the 10 A list, plus the charge self-energy.
// The reference visits each pair once (gid(j) < gid(i)) and scatters +-ff to both atoms with atomics.
// Here one wavefront owns row i and gathers the force on i from EVERY partner in the row (the pair is
// seen again from the partner's own row), so there is no scatter; each side books half of the pair
// energy.  Partner position+charge is ONE 32-byte gather from the cell-sorted copy; the partner's type
// rides in the list entry; a table lookup reads ONE 64-byte node (value + difference to the next).
#include "engine.h"

#include <cstdlib>

namespace rxmd {

static inline int nblk(long long n, int b) { return n > 0 ? static_cast<int>((n + b - 1) / b) : 1; }   // an empty rank still launches (kernels guard their range)
static constexpr double CEchrge = 23.02;   // module.F90:683

// w of the cell-sorted position array becomes the charge of the atom's owner (ghost charges = MODE_COPY payload, comm.F90:135)
__global__ void k_sorted_charge(int G, const int *__restrict__ rootperm, const double *__restrict__ q, double4 *__restrict__ pk) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < G) pk[k].w = q[rootperm[k]];
}

__device__ inline double wave_sum_n(double v) { return wave_sum64(v); }   // DPP reduction, engine.h

// one pair of a row: r^2, the table node of the type pair, energies (half of the pair from each side), force on the row's atom, pair virial
struct NbAcc { double f0, f1, f2, v0, v1, v2, v3, v4, v5, e11, e12; };
__device__ inline void nb_pair(const DevFF &ff, const int *__restrict__ ix2, double xi, double yi, double zi, double qi, const double4 &pj, int tj, NbAcc &a) {
  const double d0 = xi - pj.x, d1 = yi - pj.y, d2 = zi - pj.z;
  const double r2 = d0 * d0 + d1 * d1 + d2 * d2;
  if (r2 > ff.rctap2) return;                                     // pot.F90:720
  const int inxn = ix2[tj];
  const int itb = static_cast<int>(r2 * ff.UDRi);                 // pot.F90:729-733
  double t = r2 - itb * ff.UDR;
  t = t * ff.UDRi;
  const DevNBTab nd = ff.tabNB[static_cast<size_t>(inxn) * (NTABLE + 2) + itb];   // (four non-temporal 16-byte loads instead: 12.5 ms against 3.1 -- the table must STAY in L2)
  const double qij = qi * pj.w;
  const double CEvdw = nd.CEvdw + t * nd.dCEvdw_;
  const double CEclmb = (nd.CEclmb + t * nd.dCEclmb_) * qij;
  // every pair sits in two rows (i's and its partner's): half of the pair energy from each side
  a.e11 += 0.5 * (nd.Evdw + t * nd.dEvdw_);
  a.e12 += 0.5 * (nd.Eclmb + t * nd.dEclmb_) * qij;
  const double c = CEvdw + CEclmb;
  a.f0 -= c * d0; a.f1 -= c * d1; a.f2 -= c * d2;
  const double hc = -0.5 * c;
  a.v0 += hc * d0 * d0; a.v1 += hc * d1 * d1; a.v2 += hc * d2 * d2; a.v3 += hc * d1 * d2; a.v4 += hc * d2 * d0; a.v5 += hc * d0 * d1;
}

#ifndef NB_UNR
#define NB_UNR 8      // entries per lane and pass: a whole RDX row in one pass (measured 4.83 / 4.59 / 4.45 / 4.10 ms at 1 / 2 / 4 / 8)
#endif
#ifndef NB_WPB
#define NB_WPB 8      // rows per workgroup (measured 4.08 / 3.73 / 4.35 ms at 4 / 8 / 16)
#endif
__global__ void __launch_bounds__(64 * NB_WPB) k_nonbond(int N, int S10, DevFF ff, const int *__restrict__ nb10, const int *__restrict__ n10,
                                                  const double4 *__restrict__ pk, const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
                                                  const double *__restrict__ q, const int *__restrict__ type,
                                                  double *__restrict__ fx, double *__restrict__ fy, double *__restrict__ fz, double *__restrict__ pe, int assign) {
  __shared__ double sm[NB_WPB][3], sv[NB_WPB][6];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));   // wave-uniform -> scalar registers
  if (threadIdx.x < 6 * NB_WPB) sv[threadIdx.x / 6][threadIdx.x % 6] = 0.0;
  __syncthreads();
  const int i = xcd_swizzle(blockIdx.x, gridDim.x) * (blockDim.x >> 6) + w;
  double e11 = 0.0, e12 = 0.0, e13 = 0.0;
  if (i < N) {
    const double xi = x[i], yi = y[i], zi = z[i], qi = q[i];
    const int ti = type[i];
    const int n = n10[i] & N10_COUNT;
    const size_t row = static_cast<size_t>(i) * S10;
    const int *ix2 = ff.inxn2 + ti * ff.n1;
    NbAcc acc = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};      // force, pair virial (see the stress note at the end of the row), energies
    for (int k0 = lane; k0 < n; k0 += 64 * NB_UNR) {
      unsigned ee[NB_UNR];
#pragma unroll
      for (int u = 0; u < NB_UNR; ++u) { const int k = k0 + 64 * u; ee[u] = (k < n) ? static_cast<unsigned>(__builtin_nontemporal_load(nb10 + row + k)) : NB10_SELF; }
#pragma unroll
      for (int u = 0; u < NB_UNR; ++u) {
        const unsigned e = ee[u];
        if (e & NB10_SELF) continue;                                    // padding, or the atom's own periodic image (pot.F90:715)
        nb_pair(ff, ix2, xi, yi, zi, qi, pk[e & NB10_IDX_MASK], static_cast<int>((e >> NB10_IDX_BITS) & 15u), acc);
      }
    }
    e11 = acc.e11; e12 = acc.e12;
    double f0 = wave_sum_n(acc.f0), f1 = wave_sum_n(acc.f1), f2 = wave_sum_n(acc.f2);
    double v0 = acc.v0, v1 = acc.v1, v2 = acc.v2, v3 = acc.v3, v4 = acc.v4, v5 = acc.v5;
    // stress: the reference scatters -ff to i and +ff to its partner (possibly a ghost image), so its virial sum_a pos_a f_a
    // (pot.F90:65-72) holds (pos_i - pos_j) f_ij once per pair.  This row gathered all of f_i at pos_i instead: add the
    // difference  1/2 sum_j dr_ij f_ij - pos_i f_i  to the accumulators so that Engine::accumulate_stress sees the reference's sum
    v0 = wave_sum_n(v0); v1 = wave_sum_n(v1); v2 = wave_sum_n(v2); v3 = wave_sum_n(v3); v4 = wave_sum_n(v4); v5 = wave_sum_n(v5);
    if (lane == 0) {
      if (assign) { fx[i] = f0; fy[i] = f1; fz[i] = f2; }                  // (the bonded chain owns the force array meanwhile: Engine::force adds these behind the join)
      else { fx[i] += f0; fy[i] += f1; fz[i] += f2; }
      const DevAtomP ap = ff.atom[ti];
      e13 = CEchrge * (ap.chi * qi + 0.5 * ap.eta * qi * qi);           // pot.F90:708
      sv[w][0] = v0 - xi * f0; sv[w][1] = v1 - yi * f1; sv[w][2] = v2 - zi * f2; sv[w][3] = v3 - yi * f2; sv[w][4] = v4 - zi * f0; sv[w][5] = v5 - xi * f1;
    }
  }
  e11 = wave_sum_n(e11); e12 = wave_sum_n(e12); e13 = wave_sum_n(e13);
  if (lane == 0) { sm[w][0] = e11; sm[w][1] = e12; sm[w][2] = e13; }
  __syncthreads();
  if (threadIdx.x < 3) {
    double s = 0.0;
    for (int k = 0; k < NB_WPB; ++k) s += sm[k][threadIdx.x];
    if (s != 0.0) atomicAdd(pe + 11 + threadIdx.x, s);
  }
  if (threadIdx.x >= 64 && threadIdx.x < 70) {      // pe = scal + 32: the stress accumulators sit at scal + 48
    const int c = threadIdx.x - 64;
    double s = 0.0;
    for (int k = 0; k < NB_WPB; ++k) s += sv[k][c];
    if (s != 0.0) atomicAdd(pe + 16 + c, s);
  }
}

// ---- the same over the WINDOWS of the QEq matrix pass (engine.h WIN_*, lists.hip) ----------------------------------------------------------------
// k_nonbond sends one 32-byte gather per pair through the vector memory path for the partner's position and charge (13.5 GB of L2 -> L1 lines per
// launch) next to the 64-byte table node.  The 16 rows of a window group couple to the same ~1,200 cell-sorted positions: the workgroup copies
// (x, y, z, q) and the type of those positions into LDS once, with coalesced loads, and every pair reads its partner from there through the 16-bit
// window slot of the list entry (2 bytes streamed instead of 4).  Same lane <-> entry assignment, same arithmetic (nb_pair), same reductions as
// k_nonbond: forces and energies are bit for bit the same.  Not for boxes in which an atom meets its own image (the slot has no bit for that).
// LDS holds the first `maxunits` units of a window (33 bytes per slot; 296 units = 78 KB: two workgroups per CU); the few groups with a larger
// window (328 units is the largest of the RDX run, ~250 the mean) fetch the partners beyond it from the cell-sorted arrays as k_nonbond does.
__global__ void __launch_bounds__(64 * WIN_ROWS, 8) k_nonbond_win(int N, int G, int S10, DevFF ff, const unsigned short *__restrict__ sl10, const int *__restrict__ n10,
                                                                const int *__restrict__ rows_sorted, const int *__restrict__ win_k, const int *__restrict__ win_cnt, int maxunits,
                                                                const double4 *__restrict__ pk, const unsigned char *__restrict__ stype,
                                                                const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
                                                                const double *__restrict__ q, const int *__restrict__ type,
                                                                double *__restrict__ fx, double *__restrict__ fy, double *__restrict__ fz, double *__restrict__ pe, int assign) {
  extern __shared__ double4 s_p[];                 // window: slot -> (x, y, z, q) ...
  unsigned char *s_t = reinterpret_cast<unsigned char *>(s_p + static_cast<size_t>(maxunits) * WIN_UNIT);   // ... and the atom type
  __shared__ double sm[WIN_ROWS][3], sv[WIN_ROWS][6];
  constexpr int NT = 64 * WIN_ROWS;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int grp = xcd_swizzle(blockIdx.x, gridDim.x);
  const int ridx = grp * WIN_ROWS + w;
  const int i = min(rows_sorted[ridx], N);          // (sentinel >= N: an unused row of a cell column's last group)
  const int nslots = WIN_UNIT * win_cnt[grp];
  const int *wk = win_k + static_cast<size_t>(grp) * WIN_MAXUNITS;
  const int capslots = WIN_UNIT * maxunits;
  for (int t = threadIdx.x; t < min(nslots, capslots); t += NT) {
    const int pos = min(wk[t / WIN_UNIT] + (t & (WIN_UNIT - 1)), G - 1);
    s_p[t] = pk[pos]; s_t[t] = stype[pos];
  }
  if (threadIdx.x < 6 * WIN_ROWS) sv[threadIdx.x / 6][threadIdx.x % 6] = 0.0;
  __syncthreads();
  double e11 = 0.0, e12 = 0.0, e13 = 0.0;
  if (i < N) {
    const double xi = x[i], yi = y[i], zi = z[i], qi = q[i];
    const int ti = type[i];
    const int n = n10[i] & N10_COUNT;
    const size_t row = static_cast<size_t>(i) * S10;
    const int *ix2 = ff.inxn2 + ti * ff.n1;
    NbAcc acc = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int k0 = lane; k0 < n; k0 += 64 * NB_UNR) {
      unsigned ee[NB_UNR];
#pragma unroll
      for (int u = 0; u < NB_UNR; ++u) { const int k = k0 + 64 * u; ee[u] = (k < n) ? static_cast<unsigned>(__builtin_nontemporal_load(sl10 + row + k)) : 0xffffu; }
#pragma unroll
      for (int u = 0; u < NB_UNR; ++u) {
        if (ee[u] == 0xffffu) continue;                                  // behind the row's end (no slot is that large)
        const int sl = static_cast<int>(ee[u] & 0x7fffu);
        if (sl < capslots) nb_pair(ff, ix2, xi, yi, zi, qi, s_p[sl], static_cast<int>(s_t[sl]), acc);
        else { const int pos = min(wk[sl / WIN_UNIT] + (sl & (WIN_UNIT - 1)), G - 1); nb_pair(ff, ix2, xi, yi, zi, qi, pk[pos], static_cast<int>(stype[pos]), acc); }
      }
    }
    e11 = acc.e11; e12 = acc.e12;
    const double f0 = wave_sum_n(acc.f0), f1 = wave_sum_n(acc.f1), f2 = wave_sum_n(acc.f2);
    const double v0 = wave_sum_n(acc.v0), v1 = wave_sum_n(acc.v1), v2 = wave_sum_n(acc.v2), v3 = wave_sum_n(acc.v3), v4 = wave_sum_n(acc.v4), v5 = wave_sum_n(acc.v5);
    if (lane == 0) {
      if (assign) { fx[i] = f0; fy[i] = f1; fz[i] = f2; }                  // (the bonded chain owns the force array meanwhile: Engine::force adds these behind the join)
      else { fx[i] += f0; fy[i] += f1; fz[i] += f2; }
      const DevAtomP ap = ff.atom[ti];
      e13 = CEchrge * (ap.chi * qi + 0.5 * ap.eta * qi * qi);           // pot.F90:708
      sv[w][0] = v0 - xi * f0; sv[w][1] = v1 - yi * f1; sv[w][2] = v2 - zi * f2; sv[w][3] = v3 - yi * f2; sv[w][4] = v4 - zi * f0; sv[w][5] = v5 - xi * f1;   // the stress note of k_nonbond
    }
  }
  e11 = wave_sum_n(e11); e12 = wave_sum_n(e12); e13 = wave_sum_n(e13);
  if (lane == 0) { sm[w][0] = e11; sm[w][1] = e12; sm[w][2] = e13; }
  __syncthreads();
  if (threadIdx.x < 3) {
    double s = 0.0;
    for (int k = 0; k < WIN_ROWS; ++k) s += sm[k][threadIdx.x];
    if (s != 0.0) atomicAdd(pe + 11 + threadIdx.x, s);
  }
  if (threadIdx.x >= 64 && threadIdx.x < 70) {
    const int c = threadIdx.x - 64;
    double s = 0.0;
    for (int k = 0; k < WIN_ROWS; ++k) s += sv[k][c];
    if (s != 0.0) atomicAdd(pe + 16 + c, s);
  }
}

// ghost charges (MODE_COPY payload, comm.F90:135) and their cell-sorted copy; multi-rank: through the staged exchange
void Engine::charge_halo() {
  if (multi()) {
    halo_staged(q, 1);
    k_sorted_charge<<<nblk(G, 256), 256, 0, stream>>>(G, perm, q, sorted_xyzi);
  } else
    k_sorted_charge<<<nblk(G, 256), 256, 0, stream>>>(G, rootperm, q, sorted_xyzi);
}

void Engine::nonbonded(bool to_fnb) {
  double *f0 = to_fnb ? fnb[0] : frc[0], *f1 = to_fnb ? fnb[1] : frc[1], *f2 = to_fnb ? fnb[2] : frc[2];
  const int assign = to_fnb ? 1 : 0;
  // over the windows of the matrix pass when this list build has them and no atom can meet its own image (RXMD_NONBOND_WIN=0: the row form)
  const bool win_env = opt.nonbond_win != 0;
  const int units = std::min(win_maxunits, 296);                         // 296 units x 8 slots x 33 bytes = 78 KB: two workgroups per CU
  const size_t lds = static_cast<size_t>(units) * WIN_UNIT * (sizeof(double4) + 1) + 16;
  if (win_valid && win_env && !list_selfcheck) {
    k_nonbond_win<<<win_groups, 64 * WIN_ROWS, lds, stream>>>(N, G, S10, dff, sl10, n10, rows_sorted, win_k, win_cnt, units, sorted_xyzi, sorted_type, pos[0], pos[1], pos[2], q, type,
                                                             f0, f1, f2, scal + 32, assign);
    RX_HIP(hipGetLastError());
    return;
  }
  k_nonbond<<<nblk(N, NB_WPB), 64 * NB_WPB, 0, stream>>>(N, S10, dff, nb10, n10, sorted_xyzi, pos[0], pos[1], pos[2], q, type, f0, f1, f2, scal + 32, assign);
}

}  // namespace rxmd
