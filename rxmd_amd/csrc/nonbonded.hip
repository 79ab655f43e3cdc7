// nonbonded.hip -- ENbond (reference src/pot.F90:676-781): tabulated van der Waals + shielded Coulomb
// over the 10 A list, plus the charge self-energy.
// The reference visits each pair once (gid(j) < gid(i)) and scatters +-ff to both atoms with atomics.
// Here one wavefront owns row i and gathers the force on i from EVERY partner in the row (the pair is
// seen again from the partner's own row), so there is no scatter; energies are booked only in the
// reference's orientation gid(j) < gid(i).  Positions+charge travel as one 32-byte gather.
#include "engine.h"

namespace rxmd {

static inline int nblk(long long n, int b) { return static_cast<int>((n + b - 1) / b); }
static constexpr double CEchrge = 23.02;   // module.F90:683

__global__ void k_pack_xyzq(int G, const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z, const double *__restrict__ q,
                            const int *__restrict__ type, const long long *__restrict__ gid, double4 *__restrict__ pk, long long *__restrict__ tg) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= G) return;
  pk[i] = make_double4(x[i], y[i], z[i], q[i]);
  tg[i] = (gid[i] << 8) | static_cast<long long>(type[i]);
}

__device__ inline double wave_sum_n(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ void __launch_bounds__(256) k_nonbond(int N, int S10, DevFF ff, const int *__restrict__ nb10, const int *__restrict__ n10,
                                                  const double4 *__restrict__ pk, const long long *__restrict__ tg,
                                                  double *__restrict__ fx, double *__restrict__ fy, double *__restrict__ fz, double *__restrict__ pe) {
  __shared__ double sm[4][3];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = blockIdx.x * (blockDim.x >> 6) + w;
  double e11 = 0.0, e12 = 0.0, e13 = 0.0;
  if (i < N) {
    const double4 pi = pk[i];
    const long long tgi = tg[i];
    const int ti = static_cast<int>(tgi & 0xff);
    const long long iid = tgi >> 8;
    const int n = n10[i];
    const size_t row = static_cast<size_t>(i) * S10;
    double f0 = 0.0, f1 = 0.0, f2 = 0.0;
    for (int k0 = lane; k0 < n; k0 += 128) {
      int jj[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) { const int k = k0 + 64 * u; jj[u] = (k < n) ? __builtin_nontemporal_load(nb10 + row + k) : -1; }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int j = jj[u];
        if (j < 0) continue;
        const double4 pj = pk[j];
        const long long tgj = tg[j];
        const long long jid = tgj >> 8;
        if (jid == iid) continue;                                       // periodic self image: never a pair (pot.F90:715)
        const double d0 = pi.x - pj.x, d1 = pi.y - pj.y, d2 = pi.z - pj.z;
        const double r2 = d0 * d0 + d1 * d1 + d2 * d2;
        if (r2 > ff.rctap2) continue;                                   // pot.F90:720
        const int inxn = ff.inxn2[ti * ff.n1 + static_cast<int>(tgj & 0xff)];
        const int itb = static_cast<int>(r2 * ff.UDRi);                 // pot.F90:729-733
        double drtb = r2 - itb * ff.UDR;
        drtb = drtb * ff.UDRi;
        const double drtb1 = 1.0 - drtb;
        const DevNBTab *T = ff.tabNB + static_cast<size_t>(inxn) * (NTABLE + 2) + itb;
        const DevNBTab ta = T[0], tb = T[1];
        const double qij = pi.w * pj.w;
        const double CEvdw = drtb1 * ta.dEvdw + drtb * tb.dEvdw;
        const double CEclmb = (drtb1 * ta.dEclmb + drtb * tb.dEclmb) * qij;
        if (jid < iid) {
          e11 += drtb1 * ta.Evdw + drtb * tb.Evdw;
          e12 += (drtb1 * ta.Eclmb + drtb * tb.Eclmb) * qij;
        }
        const double c = CEvdw + CEclmb;
        f0 -= c * d0; f1 -= c * d1; f2 -= c * d2;
      }
    }
    f0 = wave_sum_n(f0); f1 = wave_sum_n(f1); f2 = wave_sum_n(f2);
    if (lane == 0) {
      fx[i] += f0; fy[i] += f1; fz[i] += f2;
      const DevAtomP ap = ff.atom[ti];
      e13 = CEchrge * (ap.chi * pi.w + 0.5 * ap.eta * pi.w * pi.w);     // pot.F90:708
    }
  }
  e11 = wave_sum_n(e11); e12 = wave_sum_n(e12); e13 = wave_sum_n(e13);
  if (lane == 0) { sm[w][0] = e11; sm[w][1] = e12; sm[w][2] = e13; }
  __syncthreads();
  if (threadIdx.x < 3) {
    double s = 0.0;
    for (int k = 0; k < 4; ++k) s += sm[k][threadIdx.x];
    if (s != 0.0) atomicAdd(pe + 11 + threadIdx.x, s);
  }
}

void Engine::nonbonded() {
  // packed per-atom data lives in the (qs,qt)/(gs,gt) vectors' neighbours: reuse the list-build scratch
  double4 *pk = sorted_xyzi;                           // free after the list build
  long long *tg = reinterpret_cast<long long *>(cc_);  // 8-byte per-atom scratch, rewritten later by assemble
  k_pack_xyzq<<<nblk(G, 256), 256, 0, stream>>>(G, pos[0], pos[1], pos[2], q, type, gid, pk, tg);
  k_nonbond<<<nblk(N, 4), 256, 0, stream>>>(N, S10, dff, nb10, n10, pk, tg, frc[0], frc[1], frc[2], scal + 32);
}

}  // namespace rxmd
