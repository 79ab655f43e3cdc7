// pqeq.hip -- the PQEq-only kernels (reference src/pqeq.F90, src/pot.F90:784-923, src/module.F90:386-445).
//   update_shell_positions (pqeq.F90:184-259)  -> k_shell_update      one wavefront per resident row of the 10 A list
//   ENbond_PQEq            (pot.F90:784-923)   -> k_nonbond_pqeq      same row-gather formulation as k_nonbond
//   spos halo + sorted copy                    -> pqeq_sorted_shells  (MODE_COPY carries spos, comm.F90:122,129-131)
// The PQEq matrix (core-core hessian, shell-core values, field term) is built by k_list10<.., true> in lists.hip and
// consumed by the QEq kernels in qeq.hip; this file holds what runs after the CG loop and inside FORCE.
//
// Beyond-cutoff lookups: get_coulomb_and_dcoulomb_pqeq returns without touching its outputs when r^2 > rctap^2
// (module.F90:401), so the reference re-uses the previous pair's value in qeq_initialize and update_shell_positions --
// a value that depends on traversal order and, in the OpenMP build, on the thread that ran the previous pair.  The engine
// gives such a lookup zero weight; see DESIGN.md "PQEq beyond-cutoff lookups".
#include "engine.h"

namespace rxmd {

static inline int nblk(long long n, int b) { return n > 0 ? static_cast<int>((n + b - 1) / b) : 1; }   // an empty rank still launches (kernels guard their range)
static constexpr double CEchrge = 23.02;     // module.F90:683
static constexpr double Cclmb0 = 332.0638;   // module.F90:681

__device__ inline bool pq_lookup2(const DevFF &ff, const double4 *__restrict__ tab, int row, double r2, double &E, double &F) {
  if (r2 > ff.rctap2) { E = 0.0; F = 0.0; return false; }
  const int itb = static_cast<int>(r2 * ff.UDRi);
  double t = r2 - itb * ff.UDR;
  t = t * ff.UDRi;
  const double4 nd = tab[static_cast<size_t>(row) * (NTABLE + 2) + itb];
  E = nd.x + t * nd.y; F = nd.z + t * nd.w;
  return true;
}
__device__ inline double wave_sum_p(double v) { return wave_sum64(v); }   // DPP reduction, engine.h

__global__ void k_sorted_shl(int G, const int *__restrict__ perm, const double *__restrict__ sx, const double *__restrict__ sy, const double *__restrict__ sz, double4 *__restrict__ out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= G) return;
  const int i = perm[k];
  out[k] = make_double4(sx[i], sy[i], sz[i], 0.0);
}
__global__ void k_ghost_copy1(int N, int G, const int *__restrict__ groot, double *__restrict__ v) {
  const int g = N + blockIdx.x * blockDim.x + threadIdx.x;
  if (g < G) v[g] = v[groot[g]];
}

void Engine::pqeq_sorted_shells() {
  for (int a = 0; a < 3; ++a) {
    if (multi()) halo_staged(shl[a], 1);
    else if (G > N) k_ghost_copy1<<<nblk(G - N, 256), 256, 0, stream>>>(N, G, groot, shl[a]);
  }
  k_sorted_shl<<<nblk(G, 256), 256, 0, stream>>>(G, perm, shl[0], shl[1], shl[2], sorted_shl);
}

// force on shell i:  -K s_i  + sum_j [ C0 Z_i (q_j+Z_j) F_sc(|shell_i - core_j|) (shell_i - core_j)  -  C0 Z_i Z_j F_ss(|shell_i - shell_j|) (shell_i - shell_j) ]
// (Eqs. 37-38); displacement = force / K, clipped to 1e-3 A per call (Eq. 39, pqeq.F90:190,245-255)
__global__ void __launch_bounds__(256) k_shell_update(int N, int S10, DevFF ff, const int *__restrict__ nb10, const int *__restrict__ n10,
                                                       const double4 *__restrict__ pk, const double4 *__restrict__ sorted_shl,
                                                       const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z, const int *__restrict__ type,
                                                       const double *__restrict__ sx, const double *__restrict__ sy, const double *__restrict__ sz,
                                                       double *__restrict__ nx, double *__restrict__ ny, double *__restrict__ nz, int edir, double efield) {
  const int lane = threadIdx.x & 63;
  const int i = xcd_swizzle(blockIdx.x, gridDim.x) * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  if (i >= N) return;
  const int ti = type[i];
  const double s0 = sx[i], s1 = sy[i], s2 = sz[i];
  const double hx = x[i] + s0, hy = y[i] + s1, hz = z[i] + s2;      // shell position
  const double Zi = ff.Zpq[ti], Ki = ff.Kspq[ti];
  const int n = n10[i] & N10_COUNT;
  const size_t row = static_cast<size_t>(i) * S10;
  double f0 = 0.0, f1 = 0.0, f2 = 0.0;
  constexpr int SU = 8;                                              // entries per lane and pass, all entry words requested first (as k_nonbond)
  for (int k0 = lane; k0 < n; k0 += 64 * SU) {
    unsigned ee[SU];
#pragma unroll
    for (int u = 0; u < SU; ++u) { const int k = k0 + 64 * u; ee[u] = (k < n) ? static_cast<unsigned>(__builtin_nontemporal_load(nb10 + row + k)) : 0xffffffffu; }
#pragma unroll
    for (int u = 0; u < SU; ++u) {
    const unsigned e = ee[u];                                        // an own periodic image stays in (the reference only excludes i == j)
    if (e == 0xffffffffu) continue;
    const int p = e & NB10_IDX_MASK, tj = (e >> NB10_IDX_BITS) & 15u;
    const double4 pj = pk[p], sj = sorted_shl[p];
    const int prow = ff.inxnpq[ti * ff.npq1 + tj];
    const double Zj = ff.Zpq[tj], qjc = pj.w + Zj;
    double d0 = hx - pj.x, d1 = hy - pj.y, d2 = hz - pj.z, E, F;
    pq_lookup2(ff, ff.tabPsc, prow, d0 * d0 + d1 * d1 + d2 * d2, E, F);
    double c = Cclmb0 * F * qjc * Zi;                                // sforce -= ff, ff = -Cclmb0*sf*qjc*Z_i (pqeq.F90:226-227)
    f0 += c * d0; f1 += c * d1; f2 += c * d2;
    d0 -= sj.x; d1 -= sj.y; d2 -= sj.z;
    pq_lookup2(ff, ff.tabPss, prow, d0 * d0 + d1 * d1 + d2 * d2, E, F);
    c = Cclmb0 * F * Zi * Zj;                                        // ff = Cclmb0*sf*Z_i*Z_j (:234-235)
    f0 -= c * d0; f1 -= c * d1; f2 -= c * d2;
    }
  }
  f0 = wave_sum_p(f0); f1 = wave_sum_p(f1); f2 = wave_sum_p(f2);
  if (lane == 0) {
    f0 -= Ki * s0; f1 -= Ki * s1; f2 -= Ki * s2;                     // Eq. 37
    if (edir == 1) f0 -= Zi * efield; else if (edir == 2) f1 -= Zi * efield; else if (edir == 3) f2 -= Zi * efield;   // pqeq.F90:205 (efield = E*Eev_kcal)
    double r0 = f0 / Ki, r1 = f1 / Ki, r2 = f2 / Ki;
    const double ddr = sqrt(r0 * r0 + r1 * r1 + r2 * r2);
    if (ddr > 1e-3) { r0 = r0 / ddr * 1e-3; r1 = r1 / ddr * 1e-3; r2 = r2 / ddr * 1e-3; }
    nx[i] = s0 + r0; ny[i] = s1 + r1; nz[i] = s2 + r2;
  }
}

__global__ void k_sorted_charge_p(int G, const int *__restrict__ rootperm, const double *__restrict__ q, double4 *__restrict__ pk) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < G) pk[k].w = q[rootperm[k]];
}

void Engine::pqeq_update_shells() {
  // partner charges: final q of this PQEq call, ghosts resolved (the reference's q(j) after the last QCOPY)
  if (multi()) { halo_staged(q, 1); k_sorted_charge_p<<<nblk(G, 256), 256, 0, stream>>>(G, perm, q, sorted_xyzi); }
  else k_sorted_charge_p<<<nblk(G, 256), 256, 0, stream>>>(G, rootperm, q, sorted_xyzi);
  // new shells into scratch (every row reads its partners' old shells), then copy back
  k_shell_update<<<nblk(N, 4), 256, 0, stream>>>(N, S10, dff, nb10, n10, sorted_xyzi, sorted_shl, pos[0], pos[1], pos[2], type, shl[0], shl[1], shl[2], cds, cd, cc_,
                                                cfg.efield_dir, cfg.efield_strength * 23.060538);
  double *tmp[3] = {cds, cd, cc_};
  for (int a = 0; a < 3; ++a) RX_HIP(hipMemcpyAsync(shl[a], tmp[a], sizeof(double) * N, hipMemcpyDeviceToDevice, stream));
  pqeq_sorted_shells();            // FORCE's MODE_COPY carries the moved shells to the ghosts (pot.F90:28)
}

// ENbond_PQEq: van der Waals from the ReaxFF table; Coulomb between cores and shells of both atoms from the three PQEq
// tables.  The reference visits a pair once (gid(i) < gid(j)) and scatters; row i here gathers the force on core i from every
// partner and books half of the pair energy; the self term (incl. the shell spring energy) goes to PE(13).
__global__ void __launch_bounds__(256) k_nonbond_pqeq(int N, int S10, DevFF ff, const int *__restrict__ nb10, const int *__restrict__ n10,
                                                       const double4 *__restrict__ pk, const double4 *__restrict__ sorted_shl,
                                                       const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
                                                       const double *__restrict__ q, const int *__restrict__ type,
                                                       const double *__restrict__ sx, const double *__restrict__ sy, const double *__restrict__ sz,
                                                       double *__restrict__ fx, double *__restrict__ fy, double *__restrict__ fz, double *__restrict__ pe) {
  __shared__ double sm[4][3], sv[4][6];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  if (threadIdx.x < 24) sv[threadIdx.x / 6][threadIdx.x % 6] = 0.0;
  __syncthreads();
  const int i = xcd_swizzle(blockIdx.x, gridDim.x) * (blockDim.x >> 6) + w;
  double e11 = 0.0, e12 = 0.0, e13 = 0.0;
  if (i < N) {
    const double xi = x[i], yi = y[i], zi = z[i], qi = q[i];
    const double s0 = sx[i], s1 = sy[i], s2 = sz[i];
    const int ti = type[i];
    const double Zi = ff.Zpq[ti], qic = qi + Zi;
    const int n = n10[i] & N10_COUNT;
    const size_t row = static_cast<size_t>(i) * S10;
    const int *ix2 = ff.inxn2 + ti * ff.n1;
    double f0 = 0.0, f1 = 0.0, f2 = 0.0;
    double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0, v4 = 0.0, v5 = 0.0;   // pair virial (stress note in nonbonded.hip)
    constexpr int PU = 4;                                           // entries per lane and pass, their entry words requested first (as k_nonbond)
    for (int k0 = lane; k0 < n; k0 += 64 * PU) {
      unsigned ee[PU];
#pragma unroll
      for (int u = 0; u < PU; ++u) { const int k = k0 + 64 * u; ee[u] = (k < n) ? static_cast<unsigned>(__builtin_nontemporal_load(nb10 + row + k)) : NB10_SELF; }
#pragma unroll
      for (int u = 0; u < PU; ++u) {
      const unsigned e = ee[u];
      if (e & NB10_SELF) continue;
      const int p = e & NB10_IDX_MASK, tj = (e >> NB10_IDX_BITS) & 15u;
      const double4 pj = pk[p], sj = sorted_shl[p];
      const double d0 = xi - pj.x, d1 = yi - pj.y, d2 = zi - pj.z;
      const double r2 = d0 * d0 + d1 * d1 + d2 * d2;
      if (r2 > ff.rctap2) continue;
      const int inxn = ix2[tj];
      const int itb = static_cast<int>(r2 * ff.UDRi);
      double t = r2 - itb * ff.UDR;
      t = t * ff.UDRi;
      const DevNBTab nd = ff.tabNB[static_cast<size_t>(inxn) * (NTABLE + 2) + itb];
      e11 += 0.5 * (nd.Evdw + t * nd.dEvdw_);
      const double CEvdw = nd.CEvdw + t * nd.dCEvdw_;
      const double Zj = ff.Zpq[tj], qjc = pj.w + Zj;
      const int prow = ff.inxnpq[ti * ff.npq1 + tj];
      double E, F, a0, a1, a2, c;
      double g0 = CEvdw * d0, g1 = CEvdw * d1, g2 = CEvdw * d2, ec;
      pq_lookup2(ff, ff.tabPcc, prow, r2, E, F);                       // core-core (pot.F90:864-867)
      c = Cclmb0 * qic * qjc; ec = c * E; g0 += c * F * d0; g1 += c * F * d1; g2 += c * F * d2;
      a0 = d0 + s0; a1 = d1 + s1; a2 = d2 + s2;                       // shell(i)-core(j) (:869-874)
      pq_lookup2(ff, ff.tabPsc, prow, a0 * a0 + a1 * a1 + a2 * a2, E, F);
      c = -Cclmb0 * Zi * qjc; ec += c * E; g0 += c * F * a0; g1 += c * F * a1; g2 += c * F * a2;
      a0 = d0 - sj.x; a1 = d1 - sj.y; a2 = d2 - sj.z;                 // core(i)-shell(j) (:876-881)
      pq_lookup2(ff, ff.tabPsc, prow, a0 * a0 + a1 * a1 + a2 * a2, E, F);
      c = -Cclmb0 * Zj * qic; ec += c * E; g0 += c * F * a0; g1 += c * F * a1; g2 += c * F * a2;
      a0 += s0; a1 += s1; a2 += s2;                                   // shell-shell (:883-888)
      pq_lookup2(ff, ff.tabPss, prow, a0 * a0 + a1 * a1 + a2 * a2, E, F);
      c = Cclmb0 * Zi * Zj; ec += c * E; g0 += c * F * a0; g1 += c * F * a1; g2 += c * F * a2;
      e12 += 0.5 * ec;
      f0 -= g0; f1 -= g1; f2 -= g2;
      v0 -= 0.5 * d0 * g0; v1 -= 0.5 * d1 * g1; v2 -= 0.5 * d2 * g2; v3 -= 0.5 * d1 * g2; v4 -= 0.5 * d2 * g0; v5 -= 0.5 * d0 * g1;
      }
    }
    f0 = wave_sum_p(f0); f1 = wave_sum_p(f1); f2 = wave_sum_p(f2);
    v0 = wave_sum_p(v0); v1 = wave_sum_p(v1); v2 = wave_sum_p(v2); v3 = wave_sum_p(v3); v4 = wave_sum_p(v4); v5 = wave_sum_p(v5);
    if (lane == 0) {
      sv[w][0] = v0 - xi * f0; sv[w][1] = v1 - yi * f1; sv[w][2] = v2 - zi * f2; sv[w][3] = v3 - yi * f2; sv[w][4] = v4 - zi * f0; sv[w][5] = v5 - xi * f1;
      fx[i] += f0; fy[i] += f1; fz[i] += f2;
      const DevAtomP ap = ff.atom[ti];
      e13 = CEchrge * (ap.chi * qi + 0.5 * ap.eta * qi * qi) + 0.5 * ff.Kspq[ti] * (s0 * s0 + s1 * s1 + s2 * s2);   // pot.F90:818-824
    }
  }
  e11 = wave_sum_p(e11); e12 = wave_sum_p(e12); e13 = wave_sum_p(e13);
  if (lane == 0) { sm[w][0] = e11; sm[w][1] = e12; sm[w][2] = e13; }
  __syncthreads();
  if (threadIdx.x < 3) {
    double s = 0.0;
    for (int k = 0; k < 4; ++k) s += sm[k][threadIdx.x];
    if (s != 0.0) atomicAdd(pe + 11 + threadIdx.x, s);
  }
  if (threadIdx.x >= 64 && threadIdx.x < 70) {      // pe = scal + 32: the stress accumulators sit at scal + 48
    const int c = threadIdx.x - 64;
    const double s = sv[0][c] + sv[1][c] + sv[2][c] + sv[3][c];
    if (s != 0.0) atomicAdd(pe + 16 + c, s);
  }
}

// EEfield (module.F90:359-383, called at pot.F90:61): force -(q_i + Z_i) E Eev_kcal on the core of every resident along the field
__global__ void k_efield(int N, DevFF ff, const int *__restrict__ type, const double *__restrict__ q, double ev, double *__restrict__ f) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) f[i] += -(q[i] + ff.Zpq[type[i]]) * ev;
}
void Engine::efield_force() {
  if (cfg.efield_dir < 1 || cfg.efield_dir > 3) return;
  k_efield<<<nblk(N, 256), 256, 0, stream>>>(N, dff, type, q, cfg.efield_strength * 23.060538, frc[cfg.efield_dir - 1]);
}

void Engine::nonbonded_pqeq() {
  k_nonbond_pqeq<<<nblk(N, 4), 256, 0, stream>>>(N, S10, dff, nb10, n10, sorted_xyzi, sorted_shl, pos[0], pos[1], pos[2], q, type, shl[0], shl[1], shl[2],
                                                frc[0], frc[1], frc[2], scal + 32);
}

}  // namespace rxmd
