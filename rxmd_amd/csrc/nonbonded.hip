// nonbonded.hip -- ENbond (reference src/pot.F90:676-781): tabulated van der Waals + shielded Coulomb
// over the 10 A list, plus the charge self-energy.
// The reference visits each pair once (gid(j) < gid(i)) and scatters +-ff to both atoms with atomics.
// Here one wavefront owns row i and gathers the force on i from EVERY partner in the row (the pair is
// seen again from the partner's own row), so there is no scatter; each side books half of the pair
// energy.  Partner position+charge is ONE 32-byte gather from the cell-sorted copy; the partner's type
// rides in the list entry; a table lookup reads ONE 64-byte node (value + difference to the next).
#include "engine.h"

namespace rxmd {

static inline int nblk(long long n, int b) { return n > 0 ? static_cast<int>((n + b - 1) / b) : 1; }   // an empty rank still launches (kernels guard their range)
static constexpr double CEchrge = 23.02;   // module.F90:683

// w of the cell-sorted position array becomes the charge of the atom's owner (ghost charges = MODE_COPY payload, comm.F90:135)
__global__ void k_sorted_charge(int G, const int *__restrict__ rootperm, const double *__restrict__ q, double4 *__restrict__ pk) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < G) pk[k].w = q[rootperm[k]];
}

__device__ inline double wave_sum_n(double v) { return wave_sum64(v); }   // DPP reduction, engine.h

#ifndef NB_UNR
#define NB_UNR 8      // entries per lane and pass: a whole RDX row in one pass (measured 4.83 / 4.59 / 4.45 / 4.10 ms at 1 / 2 / 4 / 8)
#endif
#ifndef NB_WPB
#define NB_WPB 8      // rows per workgroup (measured 4.08 / 3.73 / 4.35 ms at 4 / 8 / 16)
#endif
__global__ void __launch_bounds__(64 * NB_WPB) k_nonbond(int N, int S10, DevFF ff, const int *__restrict__ nb10, const int *__restrict__ n10,
                                                  const double4 *__restrict__ pk, const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
                                                  const double *__restrict__ q, const int *__restrict__ type,
                                                  double *__restrict__ fx, double *__restrict__ fy, double *__restrict__ fz, double *__restrict__ pe) {
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
    double f0 = 0.0, f1 = 0.0, f2 = 0.0;
    double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0, v4 = 0.0, v5 = 0.0;   // pair virial, see the stress note at the end of the row
    for (int k0 = lane; k0 < n; k0 += 64 * NB_UNR) {
      unsigned ee[NB_UNR];
#pragma unroll
      for (int u = 0; u < NB_UNR; ++u) { const int k = k0 + 64 * u; ee[u] = (k < n) ? static_cast<unsigned>(__builtin_nontemporal_load(nb10 + row + k)) : NB10_SELF; }
#pragma unroll
      for (int u = 0; u < NB_UNR; ++u) {
        const unsigned e = ee[u];
        if (e & NB10_SELF) continue;                                    // padding, or the atom's own periodic image (pot.F90:715)
        const double4 pj = pk[e & NB10_IDX_MASK];
        const double d0 = xi - pj.x, d1 = yi - pj.y, d2 = zi - pj.z;
        const double r2 = d0 * d0 + d1 * d1 + d2 * d2;
        if (r2 > ff.rctap2) continue;                                   // pot.F90:720
        const int inxn = ix2[(e >> NB10_IDX_BITS) & 15u];
        const int itb = static_cast<int>(r2 * ff.UDRi);                 // pot.F90:729-733
        double t = r2 - itb * ff.UDR;
        t = t * ff.UDRi;
        const DevNBTab nd = ff.tabNB[static_cast<size_t>(inxn) * (NTABLE + 2) + itb];
        const double qij = qi * pj.w;
        const double CEvdw = nd.CEvdw + t * nd.dCEvdw_;
        const double CEclmb = (nd.CEclmb + t * nd.dCEclmb_) * qij;
        // every pair sits in two rows (i's and its partner's): half of the pair energy from each side
        e11 += 0.5 * (nd.Evdw + t * nd.dEvdw_);
        e12 += 0.5 * (nd.Eclmb + t * nd.dEclmb_) * qij;
        const double c = CEvdw + CEclmb;
        f0 -= c * d0; f1 -= c * d1; f2 -= c * d2;
        const double hc = -0.5 * c;
        v0 += hc * d0 * d0; v1 += hc * d1 * d1; v2 += hc * d2 * d2; v3 += hc * d1 * d2; v4 += hc * d2 * d0; v5 += hc * d0 * d1;
      }
    }
    f0 = wave_sum_n(f0); f1 = wave_sum_n(f1); f2 = wave_sum_n(f2);
    // stress: the reference scatters -ff to i and +ff to its partner (possibly a ghost image), so its virial sum_a pos_a f_a
    // (pot.F90:65-72) holds (pos_i - pos_j) f_ij once per pair.  This row gathered all of f_i at pos_i instead: add the
    // difference  1/2 sum_j dr_ij f_ij - pos_i f_i  to the accumulators so that Engine::accumulate_stress sees the reference's sum
    v0 = wave_sum_n(v0); v1 = wave_sum_n(v1); v2 = wave_sum_n(v2); v3 = wave_sum_n(v3); v4 = wave_sum_n(v4); v5 = wave_sum_n(v5);
    if (lane == 0) {
      fx[i] += f0; fy[i] += f1; fz[i] += f2;
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

// ghost charges (MODE_COPY payload, comm.F90:135) and their cell-sorted copy; multi-rank: through the staged exchange
void Engine::charge_halo() {
  if (multi()) {
    halo_staged(q, 1);
    k_sorted_charge<<<nblk(G, 256), 256, 0, stream>>>(G, perm, q, sorted_xyzi);
  } else
    k_sorted_charge<<<nblk(G, 256), 256, 0, stream>>>(G, rootperm, q, sorted_xyzi);
}

void Engine::nonbonded() {
  k_nonbond<<<nblk(N, NB_WPB), 64 * NB_WPB, 0, stream>>>(N, S10, dff, nb10, n10, sorted_xyzi, pos[0], pos[1], pos[2], q, type, frc[0], frc[1], frc[2], scal + 32);
}

}  // namespace rxmd
