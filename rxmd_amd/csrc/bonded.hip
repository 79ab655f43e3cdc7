// bonded.hip -- bonded energy terms of FORCE (reference src/pot.F90) as own-slot accumulation kernels.
//   Ebond (pot.F90:926-977) + Elnpr (pot.F90:148-316) -> k_ebond_terms, k_elnpr_atoms, k_elnpr_bonds
//   E3b   (pot.F90:319-557)                           -> k_e3b   (thread per centre atom)
//   E4b   (pot.F90:980-1227)                          -> k_e4b   (wavefront per eight / four / two centre atoms, ballot-compacted work queue, persistent grid;
//                                                                  every torsion evaluated once, its k-l side delivered by k_e4b_deliver)
//   Ehb   (pot.F90:559-673)                           -> k_ehb_donors + k_ehb_sweep (donor list; a wavefront takes whole donors, acceptors staged in LDS)
// The reference scatters every derivative at once with atomics (ForceB/ForceBbo/ForceA3/ForceA4,
// pot.F90:1276-1521).  Here a thread owns one centre atom and accumulates ONLY into that atom's own
// stretch of the compact bond tables (bond o = boff[centre] + slot, engine.h):
//     cf1,cf2,cf3[slot]  dE/dBO of the bond in that slot, in ForceBbo's cf() form (pot.F90:1331):
//                        full BO, pi-minus-full, pipi-minus-full.  ForceB(c) == cf (c,0,0).
//     cdn[slot]          contribution to cdbnd of the NEIGHBOUR in that slot (pot.F90:304,536,539,1183)
//     fnx,fny,fnz[slot]  angle/torsion force on the neighbour in that slot
//     cds, f (self)      contribution to the centre's own cdbnd / force
// assemble.hip turns these into forces by pure gathers through the mirror-bond index brev: deterministic, no FP64 atomics
// (except the acceptor atom of a hydrogen bond, which is a 10 A partner and has no slot).
#include "engine.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace rxmd {

static inline int nblk(long long n, int b) { return n > 0 ? static_cast<int>((n + b - 1) / b) : 1; }   // an empty rank still launches (kernels guard their range)

static constexpr double MINBO0 = 1e-4, cutof2_esub = 1e-4;                                        // module.F90:61-62
static constexpr double MAXANGLE = 0.999999999999, MINANGLE = -0.999999999999, NSMALL = 1e-10;   // module.F90:85-87
static constexpr double PI_ = 3.14159265358979;                                                  // module.F90:90

__device__ inline double wave_sum_b(double v) { return wave_sum64(v); }   // DPP reduction, engine.h
// sum over the block, then one atomic per block into the energy accumulator
__device__ inline void block_energy_add(double v, double *dst) {
  __shared__ double sm[8];
  v = wave_sum_b(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[w] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int k = 0; k < (blockDim.x >> 6); ++k) s += sm[k];
    if (s != 0.0) atomicAdd(dst, s);
  }
}
__device__ inline double ipow7(double a) { const double a2 = a * a, a4 = a2 * a2; return (a * a2) * a4; }

struct V3 { double x, y, z; };
__device__ inline double dot(const V3 &a, const V3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// derivative of cos(i,j,k) (ForceA3, pot.F90:1462-1521): forces on i and on k; j gets -(fi+fk)
__device__ inline void angle_forces(double coeff, const V3 &rij, double nij, const V3 &rjk, double njk, V3 &fi, V3 &fk) {
  const double C00 = nij * nij, C01 = dot(rij, rjk), C11 = njk * njk;
  const double coCC = coeff * (1.0 / (nij * njk));
  const double Ci1 = -(C01 / C00), Ck2 = C01 / C11;
  fi.x = coCC * (Ci1 * rij.x + rjk.x); fi.y = coCC * (Ci1 * rij.y + rjk.y); fi.z = coCC * (Ci1 * rij.z + rjk.z);
  // fjk = -coCC*(Ck1*rij + Ck2*rjk), Ck1 = -1 ; force on k is -fjk
  fk.x = coCC * (-rij.x + Ck2 * rjk.x); fk.y = coCC * (-rij.y + Ck2 * rjk.y); fk.z = coCC * (-rij.z + Ck2 * rjk.z);
}

// ------------------------------------------------------------------------------------------------
// Ebond + Elnpr in three steps over the compact bond tables (round 4; one thread per atom looping over its bonds read nine per-bond arrays with a
// stride of ~5 entries between neighbouring lanes: 0.81 ms):
//   A  one lane per bond of a resident: the bond's terms of the two over-coordination sums (pot.F90:223-224), exp_coa3 * exp_coa4 of the bond as seen
//      from i (pot.F90:484-487), and Ebond with its coefficients where gid(j) < gid(i) (pot.F90:949)
//   B  one thread per resident: the sums in slot order (the order of the loop this replaces), the lone-pair / over- / under-coordination energies
//      and the six coefficients every bond of the atom needs (pot.F90:226-281)
//   C  one lane per bond of a resident: the coefficients applied to the bond (pot.F90:282-305)
__global__ void __launch_bounds__(256) k_ebond_terms(int nb_res, DevFF ff, const int *__restrict__ bown, const int *__restrict__ nbr, const unsigned char *__restrict__ btype, const int *__restrict__ type,
                                                      const long long *__restrict__ gid, const double *__restrict__ bo0, const double *__restrict__ bo1,
                                                      const double *__restrict__ bo2, const double *__restrict__ bo3, const double *__restrict__ delta, const double *__restrict__ deltalp,
                                                      double *__restrict__ cf1, double *__restrict__ cf2, double *__restrict__ cf3, double *__restrict__ ecoa,
                                                      double *__restrict__ t1, double *__restrict__ t2, double *__restrict__ pe) {
  const int o = xcd_swizzle(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  double e1 = 0.0;
  if (o < nb_res) {
    const int i = bown[o], j = nbr[o];
    const int tj = btype[o];
    const DevBondP bp = ff.bond[ff.inxn2[type[i] * ff.n1 + tj]];
    const double B0 = bo0[o], dj = delta[j];
    t1[o] = bp.povun1 * bp.Desig * B0;                                  // pot.F90:223
    t2[o] = (dj - deltalp[j]) * (bo2[o] + bo3[o]);                      // pot.F90:224
    {                                                                   // exp_coa3 * exp_coa4 of this bond as seen from i (pot.F90:484-487)
      const double bs = B0 - cutof2_esub, u = -bs + (dj + ff.atom[tj].Val);
      ecoa[o] = exp(-ff.pcoa3 * (u * u)) * exp(-ff.pcoa4 * ((bs - 1.5) * (bs - 1.5)));
    }
    double c1 = 0.0, c2 = 0.0, c3 = 0.0;
    if (gid[j] < gid[i]) {                                              // Ebond, pot.F90:949
      const double B1 = bo1[o];
      const double pw = pow(B1, bp.pbe2);
      const double ex = exp(bp.pbe1 * (1.0 - pw));
      e1 = -bp.Desig * B1 * ex - bp.Depi * bo2[o] - bp.Depipi * bo3[o];
      const double CEbo = -bp.Desig * ex * (1.0 - bp.pbe1 * bp.pbe2 * pw);
      c1 = CEbo; c2 = -bp.Depi - CEbo; c3 = -bp.Depipi - CEbo;         // coeff = (CEbo,-Depi,-Depipi)
    }
    cf1[o] = c1; cf2[o] = c2; cf3[o] = c3;                              // the first writer of these accumulators SETS them (k_bo_full clears only the ghosts' bonds): 0 + x == x, bit for bit
  }
  block_energy_add(e1, pe + 1);
}

__global__ void __launch_bounds__(256) k_elnpr_atoms(int N, DevFF ff, const int *__restrict__ boff, const int *__restrict__ type, const double *__restrict__ t1, const double *__restrict__ t2,
                                                      const double *__restrict__ delta, const double *__restrict__ deltalp, const double *__restrict__ dDlp,
                                                      double *__restrict__ ecoef, double *__restrict__ pe) {
  const int i = xcd_swizzle(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  double e2 = 0.0, e3 = 0.0, e4 = 0.0;
  if (i < N) {
    const DevAtomP ai = ff.atom[type[i]];
    double sum_ovun1 = 0.0, sum_ovun2 = 0.0;
    for (int o = boff[i], o1 = boff[i + 1]; o < o1; ++o) { sum_ovun1 += t1[o]; sum_ovun2 += t2[o]; }
    const double dlp = deltalp[i], dl = delta[i], dD = dDlp[i];
    const double expvd2 = exp(-75.0 * dlp);
    const double dElp = ai.plp2 * ((1.0 + expvd2) + 75.0 * dlp * expvd2) / ((1.0 + expvd2) * (1.0 + expvd2));
    const double expovun1 = ff.povun3 * exp(ff.povun4 * sum_ovun2);
    const double dcorr = dl - dlp / (1.0 + expovun1);
    const double expovun2 = exp(ai.povun2 * dcorr);
    const double DlpV = 1.0 / (dcorr + ai.Val + 1e-8);
    const double expovun2n = 1.0 / expovun2;
    const double expovun6 = exp(ff.povun6 * dcorr);
    const double expovun8 = ff.povun7 * exp(ff.povun8 * sum_ovun2);
    const double d1 = 1.0 / (1.0 + expovun1), d2 = 1.0 / (1.0 + expovun2), d2n = 1.0 / (1.0 + expovun2n), d8 = 1.0 / (1.0 + expovun8);
    e2 = ai.plp2 * dlp / (1.0 + expvd2);
    e3 = sum_ovun1 * DlpV * dcorr * d2;
    const double PEunder = -ai.povun5 * (1.0 - expovun6) * d2n * d8;
    e4 = PEunder;
    const double CElp1 = dElp * dD;
    const double CEo1 = dcorr * DlpV * d2;
    const double CEo2 = sum_ovun1 * DlpV * d2 * (1.0 - dcorr * DlpV - ai.povun2 * dcorr * d2n);
    const double CEo3 = CEo2 * (1.0 - dD * d1);
    const double CEo4 = CEo2 * dlp * ff.povun4 * expovun1 * (d1 * d1);
    const double CEu1 = (ai.povun5 * ff.povun6 * expovun6 * d8 + PEunder * ai.povun2 * expovun2n) * d2n;
    const double CEu2 = -PEunder * ff.povun8 * expovun8 * d8;
    const double CEu3 = CEu1 * (1.0 - dD * d1);
    const double CEu4 = CEu1 * dlp * ff.povun4 * expovun1 * (d1 * d1) + CEu2;
    double *c = ecoef + 6 * static_cast<size_t>(i);
    c[0] = CElp1; c[1] = CEo3; c[2] = CEo1; c[3] = CEu3; c[4] = CEo4; c[5] = CEu4;
  }
  block_energy_add(e2, pe + 2); block_energy_add(e3, pe + 3); block_energy_add(e4, pe + 4);
}

__global__ void __launch_bounds__(256) k_elnpr_bonds(int nb_res, DevFF ff, const int *__restrict__ bown, const int *__restrict__ nbr, const unsigned char *__restrict__ btype, const int *__restrict__ type,
                                                      const double *__restrict__ bo2, const double *__restrict__ bo3, const double *__restrict__ delta, const double *__restrict__ deltalp,
                                                      const double *__restrict__ dDlp, const double *__restrict__ ecoef,
                                                      double *__restrict__ cf1, double *__restrict__ cf2, double *__restrict__ cf3, double *__restrict__ cdn) {
  const int o = xcd_swizzle(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  if (o >= nb_res) return;
  const int i = bown[o], j = nbr[o];
  const double *c = ecoef + 6 * static_cast<size_t>(i);
  const double CElp1 = c[0], CEo3 = c[1], CEo1 = c[2], CEu3 = c[3], CEo4 = c[4], CEu4 = c[5];
  const DevBondP bp = ff.bond[ff.inxn2[type[i] * ff.n1 + btype[o]]];
  const double bpp = bo2[o] + bo3[o], dj = delta[j] - deltalp[j], oneD = 1.0 - dDlp[j];
  const double CEo5 = CEo1 * bp.povun1 * bp.Desig;
  const double CElp_b = CElp1 + CEo3 + CEo5 + CEu3;
  const double CElp_bpp = CEo4 * dj + CEu4 * dj;
  cf1[o] += CElp_b; cf2[o] += CElp_bpp; cf3[o] += CElp_bpp;           // coeff = CElp_b + (0,bpp,bpp)
  cdn[o] = CEo4 * oneD * bpp + CEu4 * oneD * bpp;                      // cdbnd(j) += CElp_d (first writer: sets)
}

// ------------------------------------------------------------------------------------------------
// wave-level helpers for the wavefront-per-centre kernels: LDS hand-off between the lanes of ONE wavefront
// (the LDS queue of a wave is in order; the fences only stop the compiler from moving accesses across)
__device__ inline void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
constexpr int E4B_SEG = ~(31 << 5);   // k_e4b queue key without its i1 field: (atom, centre bond, l)
constexpr int WSLOT = 31;   // bonded slots a wavefront-per-centre kernel stages in LDS (MAXNEIGHBS = 30; slot 31 = the centre atom itself)

// ------------------------------------------------------------------------------------------------
// occupancy knobs, measured on MI355X (bonded section of the step, ms): E3b at 1/2/3/4 workgroups per CU = 22.5/20.4/22.4/22.9
// (256 VGPRs + 3 AGPRs miss the 2-wave budget by three registers without the hint; beyond 2 the spills cost more than
// the occupancy buys), E4b at 2/3 = 21.3/20.4 (before its LDS-atomic accumulators; with them 3 stays best: 15.2, at 4: 18.9)
#ifndef E3B_MINB
#define E3B_MINB 2
#endif
#ifndef E3B_CAP
#define E3B_CAP 448     // bonds of a wavefront's 64 atoms whose accumulators k_e3b keeps in LDS (RDX: ~340 per wavefront; 4 x 5 x 448 x 8 B = 70 KB per workgroup)
#endif
#ifndef E4B_MINB
#define E4B_MINB 3
#endif
// Work order of the thread-per-centre angle kernel: atom order.  (Round 2 first handed the centres out sorted by their number of bonds
// above the cut-off -- 3.5 -> 2.8 ms while the loops walked slots; once they walk set bits (below) a lane's trip count is its own
// pair count, and the sorted order only scatters the slot-major accesses, one 64-byte line per lane: 2.7 ms sorted, 1.8 ms in atom
// order, sorted inside tiles of 128 / 256 / 512 / 1024 atoms 1.84 / 1.97 / 2.86 / 4.19 ms.)
__global__ void __launch_bounds__(256, E3B_MINB) k_e3b(int N, DevFF ff, const int *__restrict__ boff, const int *__restrict__ nbr, const unsigned char *__restrict__ btype, const int *__restrict__ type,
                                              const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
                                              const double *__restrict__ bo0, const double *__restrict__ bo2, const double *__restrict__ bo3,
                                              const double *__restrict__ delta, const double *__restrict__ nlp, const double *__restrict__ dDlp,
                                              const double *__restrict__ epen, const double *__restrict__ ecoa,
                                              double *__restrict__ cf1, double *__restrict__ cf2, double *__restrict__ cf3, double *__restrict__ cdn,
                                              double *__restrict__ fnx, double *__restrict__ fny, double *__restrict__ fnz,
                                              double *__restrict__ cds, double *__restrict__ fx, double *__restrict__ fy, double *__restrict__ fz, double *__restrict__ pe) {
  const int tid = xcd_swizzle(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;   // an XCD's workgroups own one contiguous eighth of the atoms: the neighbours they gather stay in its L2
  // Round 4: the five accumulators of the bonds of a wavefront's 64 atoms -- one contiguous stretch of the compact tables -- live in LDS while the
  // angles are walked.  Every angle adds to the j-k bond's coefficient, cdbnd term and neighbour force: five read-modify-writes of global memory per
  // angle, each a dependent round trip inside the loop (the compiler must assume that a later iteration hits the same bond).  The stretch is loaded
  // with the values the earlier kernels left (coalesced), updated in LDS by the thread that owns the atom (no atomics: a thread touches only its own
  // bonds), and stored back at the end: the same additions in the same order, bit for bit.  Bonds beyond E3B_CAP of a wavefront (SiC: 16 bonds per
  // atom) take the global path as before.
  extern __shared__ double s_e3b[];
  const int lane_ = threadIdx.x & 63, wv_ = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  double *l_cf = s_e3b + static_cast<size_t>(wv_) * 5 * E3B_CAP, *l_cd = l_cf + E3B_CAP, *l_fx = l_cd + E3B_CAP, *l_fy = l_fx + E3B_CAP, *l_fz = l_fy + E3B_CAP;
  const int a0_ = tid - lane_;                                     // first atom of this wavefront
  const int ob_w = a0_ < N ? boff[a0_] : 0, oe_w = a0_ < N ? boff[min(a0_ + 64, N)] : 0;
  const int nst = min(oe_w - ob_w, E3B_CAP);                       // staged bonds
  for (int t = lane_; t < nst; t += 64) { l_cf[t] = cf1[ob_w + t]; l_cd[t] = cdn[ob_w + t]; l_fx[t] = fnx[ob_w + t]; l_fy[t] = fny[ob_w + t]; l_fz[t] = fnz[ob_w + t]; }
  wave_lds_sync();
  auto acc1 = [&](double *__restrict__ gl, double *ld, int o, double v) { const int r = o - ob_w; if (r < nst) ld[r] += v; else gl[o] += v; };
  // the angle row of a type triple and its seven parameters from LDS (round 5): inxn3 -> angle[] were two dependent look-ups in front of every
  // angle's arithmetic, at two wavefronts per SIMD nothing hides them.  Force fields with more than 7 atom types or 63 angle rows keep the global path.
  __shared__ int s_ix3[512];
  __shared__ double s_ang[64][7];
  const bool ang_lds = ff.n1 <= 8 && ff.nvaty <= 63;
  if (ang_lds) {
    for (int t = threadIdx.x; t < ff.n1 * ff.n1 * ff.n1; t += 256) s_ix3[t] = ff.inxn3[t];
    for (int t = threadIdx.x; t < 7 * (ff.nvaty + 1); t += 256) s_ang[t / 7][t % 7] = reinterpret_cast<const double *>(ff.angle)[t];
    __syncthreads();
  }
  double e5 = 0.0, e6 = 0.0, e7 = 0.0;
  if (tid < N) {
    const int j = tid;
    const int tj = type[j], ob = boff[j], nj = min(boff[j + 1] - ob, 32);
    const DevAtomP aj = ff.atom[tj];
    const double xj = x[j], yj = y[j], zj = z[j];
    double sum_BO8 = 0.0, sum_SBO1 = 0.0;
    unsigned capmask = 0u;
    for (int n1 = 0; n1 < nj; ++n1) {
      const int o = ob + n1;
      const double b = bo0[o], b2 = b * b, b4 = b2 * b2;
      sum_BO8 -= b4 * b4;                                                  // BO**8, pot.F90:362
      sum_SBO1 += bo2[o] + bo3[o];
      if (b - cutof2_esub > 0.0) capmask |= 1u << n1;                      // pot.F90:372-373, 385-386
    }
    const double prod_SBO = exp(sum_BO8);
    const double dlj = delta[j];
    const double delta_ang = dlj + aj.Val - aj.Valangle;
    const double nlpj = nlp[j], dDj = dDlp[j];
    // per-centre factors of the penalty and conjugation terms (global parameters, pot.F90:460-462,481-483)
    const double exp_pen3 = exp(-ff.ppen3 * dlj), exp_pen4 = exp(ff.ppen4 * dlj);
    const double trm34 = 1.0 + exp_pen3 + exp_pen4;
    const double fn9 = (2.0 + exp_pen3) / trm34;
    const double Cf9j = (-ff.ppen3 * exp_pen3 * trm34 - (2.0 + exp_pen3) * (-ff.ppen3 * exp_pen3 + ff.ppen4 * exp_pen4)) / (trm34 * trm34);
    const double delta_val = dlj + aj.Val - aj.Valval;
    const double exp_coa2 = exp(ff.pcoa2 * delta_val);
    const double exp6 = exp(ff.pval6 * delta_ang);
    // SBO, SBO2, its derivative factor and exp(-pval10 (2 - SBO2)) depend on the centre only (pot.F90:415-437): once per atom
    const double SBO = sum_SBO1 + (1.0 - prod_SBO) * (-delta_ang - ff.pval8 * nlpj);
    double SBO2 = 0.0, CSBO2 = 0.0;
    if (SBO > 0.0 && SBO <= 1.0) { SBO2 = pow(SBO, ff.pval9); CSBO2 = ff.pval9 * pow(SBO, ff.pval9 - 1.0); }
    else if (SBO > 1.0 && SBO <= 2.0) { SBO2 = 2.0 - pow(2.0 - SBO, ff.pval9); CSBO2 = ff.pval9 * pow(2.0 - SBO, ff.pval9 - 1.0); }
    else if (SBO > 2.0) SBO2 = 2.0;
    const double ex10 = exp(-ff.pval10 * (2.0 - SBO2));
    const double dSBO1 = -8.0 * prod_SBO * (delta_ang + ff.pval8 * nlpj);
    const double dSBO2 = (prod_SBO - 1.0) * (1.0 - ff.pval8 * dDj);
    // sums over all angles of the terms that ForceBbo(j,n1,...) applies to EVERY bond of j (pot.F90:526-532)
    double S_d1 = 0.0, S_v6 = 0.0, S_v5 = 0.0;
    V3 fself = {0.0, 0.0, 0.0};
    for (unsigned mi = capmask; mi & (mi - 1u);) {
      const int i1 = __ffs(mi) - 1;
      mi &= mi - 1u;
      const int oi = ob + i1;
      const double BOij_f = bo0[oi], BOij = BOij_f - cutof2_esub;
      const int i = nbr[oi], ti = btype[oi];
      const V3 rij = {x[i] - xj, y[i] - yj, z[i] - zj};
      const double nij = sqrt(dot(rij, rij));
      double ai_cf = 0.0, ai_cd = 0.0, ai_fx = 0.0, ai_fy = 0.0, ai_fz = 0.0;      // the i-j bond's own sums over k1: one write after the loop
      for (unsigned mk = mi; mk; mk &= mk - 1u) {
        const int k1 = __ffs(mk) - 1;
        const int ok = ob + k1;
        const double BOjk_f = bo0[ok], BOjk = BOjk_f - cutof2_esub;
        if (!(BOij_f * BOjk_f > cutof2_esub)) continue;
        const int k = nbr[ok], tk = btype[ok];
        const int inxn = ang_lds ? s_ix3[(ti * ff.n1 + tj) * ff.n1 + tk] : ff.inxn3[(ti * ff.n1 + tj) * ff.n1 + tk];
        if (inxn == 0) continue;
        DevAngleP ap;
        if (ang_lds) { const double *a_ = s_ang[inxn]; ap = DevAngleP{a_[0], a_[1], a_[2], a_[3], a_[4], a_[5], a_[6]}; }
        else ap = ff.angle[inxn];
        const V3 rjk = {xj - x[k], yj - y[k], zj - z[k]};
        const double njk = sqrt(dot(rjk, rjk));
        double cos_ijk = -dot(rij, rjk) / (nij * njk);
        if (cos_ijk > MAXANGLE) cos_ijk = MAXANGLE;
        if (cos_ijk < MINANGLE) cos_ijk = MINANGLE;
        const double theta_ijk = acos(cos_ijk), sin_ijk = sqrt((1.0 - cos_ijk) * (1.0 + cos_ijk));   // sin(acos(c))
        const double BOij_p4 = pow(BOij, ap.pval4), exp3ij = exp(-aj.pval3 * BOij_p4), fn7ij = 1.0 - exp3ij;
        const double BOjk_p4 = pow(BOjk, ap.pval4), exp3jk = exp(-aj.pval3 * BOjk_p4), fn7jk = 1.0 - exp3jk;
        const double exp7 = exp(-ap.pval7 * delta_ang), trm8 = 1.0 + exp6 + exp7;
        const double fn8j = aj.pval5 - (aj.pval5 - 1.0) * (2.0 + exp6) / trm8;
        const double theta0 = PI_ - ap.theta00 * (1.0 - ex10);
        const double theta_diff = theta0 - theta_ijk;
        const double exp2 = exp(-ap.pval2 * theta_diff * theta_diff);
        e5 += fn7ij * fn7jk * fn8j * (ap.pval1 - ap.pval1 * exp2);
        const double Cf7ij = aj.pval3 * ap.pval4 * (BOij_p4 / BOij) * exp3ij;                 // BO**(pval4-1) = BO**pval4 / BO
        const double Cf7jk = aj.pval3 * ap.pval4 * (BOjk_p4 / BOjk) * exp3jk;
        const double Cf8j = (1.0 - aj.pval5) / (trm8 * trm8) * (ff.pval6 * exp6 * trm8 - (2.0 + exp6) * (ff.pval6 * exp6 - ap.pval7 * exp7));
        const double Ctheta0 = ff.pval10 * ap.theta00 * ex10;
        const double CEval1 = Cf7ij * fn7jk * fn8j * ap.pval1 * (1.0 - exp2);
        const double CEval2 = fn7ij * Cf7jk * fn8j * ap.pval1 * (1.0 - exp2);
        const double CEval3 = fn7ij * fn7jk * Cf8j * ap.pval1 * (1.0 - exp2);
        const double CEval4 = 2.0 * ap.pval1 * ap.pval2 * fn7ij * fn7jk * fn8j * exp2 * theta_diff;
        const double CEval5 = CEval4 * Ctheta0 * CSBO2;
        const double CEval6 = CEval5 * dSBO1, CEval7 = CEval5 * dSBO2, CEval8 = CEval4 / sin_ijk;
        // penalty, pot.F90:460-476
        const double PEpen = ap.ppen1 * fn9 * epen[oi] * epen[ok];
        e6 += PEpen;
        const double CEpen1 = Cf9j / fn9 * PEpen, CEpen2 = -2.0 * ff.ppen2 * (BOij - 2.0) * PEpen, CEpen3 = -2.0 * ff.ppen2 * (BOjk - 2.0) * PEpen;
        // three-body conjugation, pot.F90:479-497
        const double sum_BOi = delta[i] + ff.atom[ti].Val, sum_BOk = delta[k] + ff.atom[tk].Val;
        const double ui = -BOij + sum_BOi, uk = -BOjk + sum_BOk;
        const double PEcoa = ap.pcoa1 / (1.0 + exp_coa2) * ecoa[oi] * ecoa[ok];
        e7 += PEcoa;
        const double CEcoa1 = -2.0 * ff.pcoa4 * (BOij - 1.5) * PEcoa, CEcoa2 = -2.0 * ff.pcoa4 * (BOjk - 1.5) * PEcoa;
        const double CEcoa3 = -ff.pcoa2 * exp_coa2 / (1.0 + exp_coa2) * PEcoa;
        const double CEcoa4 = -2.0 * ff.pcoa3 * ui * PEcoa, CEcoa5 = -2.0 * ff.pcoa3 * uk * PEcoa;
        // accumulate, pot.F90:509-541
        ai_cf += CEpen2 + CEcoa1 - CEcoa4 + CEval1;        // ForceB on bond i-j
        acc1(cf1, l_cf, ok, CEpen3 + CEcoa2 - CEcoa5 + CEval2);      // ForceB on bond j-k
        S_d1 += CEpen1 + CEcoa3 + CEval3 + CEval7; S_v6 += CEval6; S_v5 += CEval5;
        ai_cd += CEcoa4; acc1(cdn, l_cd, ok, CEcoa5);      // cdbnd(i), cdbnd(k)
        V3 fi, fk;
        angle_forces(CEval8, rij, nij, rjk, njk, fi, fk);
        ai_fx += fi.x; ai_fy += fi.y; ai_fz += fi.z;
        acc1(fnx, l_fx, ok, fk.x); acc1(fny, l_fy, ok, fk.y); acc1(fnz, l_fz, ok, fk.z);
        fself.x -= fi.x + fk.x; fself.y -= fi.y + fk.y; fself.z -= fi.z + fk.z;
      }
      if (ai_cf != 0.0) acc1(cf1, l_cf, oi, ai_cf);
      if (ai_cd != 0.0) acc1(cdn, l_cd, oi, ai_cd);
      if (ai_fx != 0.0 || ai_fy != 0.0 || ai_fz != 0.0) { acc1(fnx, l_fx, oi, ai_fx); acc1(fny, l_fy, oi, ai_fy); acc1(fnz, l_fz, oi, ai_fz); }
    }
    if (S_d1 != 0.0 || S_v6 != 0.0 || S_v5 != 0.0)
      for (int n1 = 0; n1 < nj; ++n1) {
        const int o = ob + n1;
        acc1(cf1, l_cf, o, S_d1 + S_v6 * ipow7(bo0[o])); cf2[o] += S_v5; cf3[o] += S_v5;
      }
    fx[j] += fself.x; fy[j] += fself.y; fz[j] += fself.z;
  }
  wave_lds_sync();
  for (int t = lane_; t < nst; t += 64) { cf1[ob_w + t] = l_cf[t]; cdn[ob_w + t] = l_cd[t]; fnx[ob_w + t] = l_fx[t]; fny[ob_w + t] = l_fy[t]; fnz[ob_w + t] = l_fz[t]; }
  block_energy_add(e5, pe + 5); block_energy_add(e6, pe + 6); block_energy_add(e7, pe + 7);
}

// Valence angles through a work queue (round 6, late).  k_e3b gives every centre atom a thread: a lane's trip count is its own number of angles (1 to ~40)
// and a wavefront runs as long as its busiest lane -- the vector unit issues for 0.62 of the kernel's 1.25 ms with mostly empty lanes.  Here a wavefront owns
// SIXTEEN consecutive centre atoms (the shape of k_e4b): lanes 0..15 set the atoms up (per-centre factors -> LDS), all 64 lanes enumerate the pairs of
// qualifying bonds (four lanes per atom), apply the reference's cut-offs (pot.F90:385-400) and compact the survivors with a ballot into an LDS ring; a full
// batch of 64 angles is then evaluated one per lane.  Every angle adds to the accumulators of its two bonds in LDS (ds_add_f64, queue order: the same bits
// run to run); the centre's own force is minus the sum of its bonds' forces, its all-bond terms are sums per atom; one coalesced pass adds everything to
// the bond tables.  Persistent grid.  Sixteen centres per wavefront with bond lists up to 12, eight up to 24, four beyond; needs the type-triple table in
// LDS (at most 7 atom types, 255 angle rows; the rows' parameters from LDS up to 63 rows, from memory beyond); otherwise k_e3b runs.
constexpr int E3Q_BC = 192;                           // bonds of a wavefront's centre atoms: NA atoms x MAXL = 192 / NA list entries -- 16 x 12 (RDX, water), 8 x 24, 4 x 48 (lists up to MAXNEIGHBS: SiC, iron sulfide)
template <int MINW, int NA>                           // wavefronts per SIMD the register budget is cut for (3: 168 registers + scratch, 2: 256); centre atoms per wavefront
__global__ void __launch_bounds__(256, MINW) k_e3q(int N, int NG, DevFF ff, const int *__restrict__ boff, const int *__restrict__ nbr, const unsigned char *__restrict__ btype, const int *__restrict__ type,
                                              const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
                                              const double *__restrict__ bo0, const double *__restrict__ bo2, const double *__restrict__ bo3,
                                              const double *__restrict__ delta, const double *__restrict__ nlp, const double *__restrict__ dDlp,
                                              const double *__restrict__ epen, const double *__restrict__ ecoa,
                                              double *__restrict__ cf1, double *__restrict__ cf2, double *__restrict__ cf3, double *__restrict__ cdn,
                                              double *__restrict__ fnx, double *__restrict__ fny, double *__restrict__ fnz,
                                              double *__restrict__ cds, double *__restrict__ fx, double *__restrict__ fy, double *__restrict__ fz, double *__restrict__ pe, int probe) {
  constexpr int E3Q_NA = NA, E3Q_MAXL = (E3Q_BC / NA < 32) ? E3Q_BC / NA : 32, LPA = 64 / NA;     // longest bond list (MAXNEIGHBS = 30), lanes per atom of the enumeration
  __shared__ unsigned char s_ix3[512];             // angle row of a type triple (at most 63 rows)
  __shared__ double s_ang[64][7];
  // per wavefront: factors of the 16 centres, accumulators of their bonds and of the atoms, the qualifying slots, the queue
  __shared__ double s_c[4][16][12];            // x, y, z, delta_ang, exp6, ex10, CSBO2, dSBO1, dSBO2, fn9, Cf9j / fn9, exp_coa2
  __shared__ int s_ci[4][16][4];               // type, first bond, bonds, qualifying bonds
  __shared__ double s_acc[4][5][E3Q_BC];           // per bond: cf1, cdn, force on the neighbour
  __shared__ double s_at[4][16][3];            // per atom: S_d1, S_v6, S_v5 (the terms every bond of the centre receives, pot.F90:526-532)
  __shared__ unsigned char s_cap[4][256], s_own[4][E3Q_BC];        // qualifying slots: NA x (at most 32 = 256 / NA ... 16) entries, [a * CAPW + u]
  __shared__ float s_cbo[4][E3Q_BC];               // [a * MAXL + u]     // bond orders of the qualifying slots, single precision: the product test of the enumeration asks memory only at the edge
  __shared__ unsigned short s_q[4][128];
  constexpr int CAPW = (256 / NA < 32) ? 256 / NA : 32;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  for (int t = threadIdx.x; t < ff.n1 * ff.n1 * ff.n1; t += 256) s_ix3[t] = static_cast<unsigned char>(ff.inxn3[t]);
  const bool ang_lds = ff.nvaty <= 63;              // (more angle rows than the LDS table holds, up to 255: their parameters come from memory, requested with the angle's other operands)
  if (ang_lds) for (int t = threadIdx.x; t < 7 * (ff.nvaty + 1); t += 256) s_ang[t / 7][t % 7] = reinterpret_cast<const double *>(ff.angle)[t];
  __syncthreads();
  double e5 = 0.0, e6 = 0.0, e7 = 0.0;
  for (int vb = blockIdx.x; vb < NG; vb += gridDim.x) {
  const int jbase = (xcd_swizzle(vb, NG) * 4 + w) * E3Q_NA;
  if (jbase >= N) continue;                        // (whole wavefront; no block-level barrier below)
  const int na = min(E3Q_NA, N - jbase);
  const int ob_w = boff[jbase], nb_w = boff[jbase + na] - ob_w;             // the bonds of the wavefront's atoms: one contiguous stretch of the compact tables
  // the bond orders of the wavefront's bonds with coalesced loads into the (not yet used) accumulator rows 0 and 1: the centres' loops over their bonds
  // below read LDS -- as loads from memory they were up to twelve round trips one after the other in front of everything else
  for (int r = lane; r < nb_w; r += 64) { s_acc[w][0][r] = bo0[ob_w + r]; s_acc[w][1][r] = bo2[ob_w + r] + bo3[ob_w + r]; }
  for (int t = lane; t < 3 * E3Q_BC; t += 64) (&s_acc[w][2][0])[t] = 0.0;
  if (lane < E3Q_NA) { s_at[w][lane][0] = 0.0; s_at[w][lane][1] = 0.0; s_at[w][lane][2] = 0.0; s_ci[w][lane][2] = 0; s_ci[w][lane][3] = 0; }
  wave_lds_sync();
  if (lane < na) {                                 // set-up of a centre (pot.F90:352-437, 460-462, 481-483): everything that depends on the centre only
    const int j = jbase + lane;
    const int tj = type[j], ob = boff[j], nj = min(boff[j + 1] - ob, E3Q_MAXL);
    const DevAtomP aj = ff.atom[tj];
    double sum_BO8 = 0.0, sum_SBO1 = 0.0;
    int nc = 0;
    for (int n1 = 0; n1 < nj; ++n1) {
      const int r = ob + n1 - ob_w;
      const double b = s_acc[w][0][r], b2 = b * b, b4 = b2 * b2;
      sum_BO8 -= b4 * b4;                                                  // BO**8, pot.F90:362
      sum_SBO1 += s_acc[w][1][r];
      if (b - cutof2_esub > 0.0) { s_cbo[w][lane * E3Q_MAXL + nc] = static_cast<float>(b); s_cap[w][lane * CAPW + nc++] = static_cast<unsigned char>(n1); }   // pot.F90:372-373, 385-386
      s_own[w][r] = static_cast<unsigned char>(lane);
    }
    const double prod_SBO = exp(sum_BO8);
    const double dlj = delta[j];
    const double delta_ang = dlj + aj.Val - aj.Valangle;
    const double nlpj = nlp[j], dDj = dDlp[j];
    const double exp_pen3 = exp(-ff.ppen3 * dlj), exp_pen4 = exp(ff.ppen4 * dlj);
    const double trm34 = 1.0 + exp_pen3 + exp_pen4;
    const double fn9 = (2.0 + exp_pen3) / trm34;
    const double Cf9j = (-ff.ppen3 * exp_pen3 * trm34 - (2.0 + exp_pen3) * (-ff.ppen3 * exp_pen3 + ff.ppen4 * exp_pen4)) / (trm34 * trm34);
    const double delta_val = dlj + aj.Val - aj.Valval;
    const double SBO = sum_SBO1 + (1.0 - prod_SBO) * (-delta_ang - ff.pval8 * nlpj);
    double SBO2 = 0.0, CSBO2 = 0.0;
    // (b ** p and b ** (p - 1) from ONE exp(p log b), b > 0 in both branches: two calls of the library's pow are ~600 of the set-up's ~900 instructions)
    if (SBO > 0.0 && SBO <= 1.0) { const double pw = exp(ff.pval9 * log(SBO)); SBO2 = pw; CSBO2 = ff.pval9 * (pw / SBO); }
    else if (SBO > 1.0 && SBO < 2.0) { const double b_ = 2.0 - SBO, pw = exp(ff.pval9 * log(b_)); SBO2 = 2.0 - pw; CSBO2 = ff.pval9 * (pw / b_); }
    else if (SBO >= 2.0) SBO2 = 2.0;
    double *c = s_c[w][lane];
    c[0] = x[j]; c[1] = y[j]; c[2] = z[j]; c[3] = delta_ang; c[4] = exp(ff.pval6 * delta_ang); c[5] = exp(-ff.pval10 * (2.0 - SBO2)); c[6] = CSBO2;
    c[7] = -8.0 * prod_SBO * (delta_ang + ff.pval8 * nlpj); c[8] = (prod_SBO - 1.0) * (1.0 - ff.pval8 * dDj); c[9] = fn9; c[10] = Cf9j / fn9; c[11] = exp(ff.pcoa2 * delta_val);
    s_ci[w][lane][0] = tj; s_ci[w][lane][1] = ob; s_ci[w][lane][2] = nj; s_ci[w][lane][3] = nc;
  }
  wave_lds_sync();
  for (int t = lane; t < 2 * E3Q_BC; t += 64) (&s_acc[w][0][0])[t] = 0.0;         // rows 0 and 1 become accumulators
  wave_lds_sync();
  int qn = 0, qh = 0;
  auto evaluate = [&](int cnt) {                   // one angle per lane: queue entry = atom << 10 | slot i1 << 5 | slot k1 (i1 < k1)
#ifdef RXMD_EXPERIMENTS
    if (probe == 2) return;
    if (probe == 5) { if (lane == 0) { atomicAdd(pe + 5, 1.0); atomicAdd(pe + 6, static_cast<double>(cnt)); } return; }
#endif
    if (lane < cnt) {
      const int key = s_q[w][(qh + lane) & 127];
      const int a = key >> 10, i1 = (key >> 5) & 31, k1 = key & 31;
      const double *c = s_c[w][a];
      const int tj = s_ci[w][a][0], ob = s_ci[w][a][1];
      const int oi = ob + i1, ok = ob + k1;
      const double xj = c[0], yj = c[1], zj = c[2], delta_ang = c[3], exp6 = c[4], ex10 = c[5], CSBO2 = c[6], dSBO1 = c[7], dSBO2 = c[8], fn9 = c[9], Cf9j_fn9 = c[10], exp_coa2 = c[11];
      const DevAtomP aj = ff.atom[tj];
      const int i = nbr[oi], ti = btype[oi], k = nbr[ok], tk = btype[ok];
      const double BOij = bo0[oi] - cutof2_esub, BOjk = bo0[ok] - cutof2_esub;
      const V3 rij = {x[i] - xj, y[i] - yj, z[i] - zj};
      const double nij = sqrt(dot(rij, rij));
      const V3 rjk = {xj - x[k], yj - y[k], zj - z[k]};
      const double njk = sqrt(dot(rjk, rjk));
      const int inxn = s_ix3[(ti * ff.n1 + tj) * ff.n1 + tk];
      DevAngleP ap;
      if (ang_lds) { const double *a_ = s_ang[inxn]; ap = DevAngleP{a_[0], a_[1], a_[2], a_[3], a_[4], a_[5], a_[6]}; }
      else ap = ff.angle[inxn];
      double cos_ijk = -dot(rij, rjk) / (nij * njk);
      if (cos_ijk > MAXANGLE) cos_ijk = MAXANGLE;
      if (cos_ijk < MINANGLE) cos_ijk = MINANGLE;
      const double theta_ijk = acos(cos_ijk), sin_ijk = sqrt((1.0 - cos_ijk) * (1.0 + cos_ijk));   // sin(acos(c))
      // BO ** pval4 as exp(pval4 log BO): BO > 0 here, and the library's pow spends most of its ~300 instructions on cases that cannot occur (1e-15 relative apart)
      const double BOij_p4 = exp(ap.pval4 * log(BOij)), exp3ij = exp(-aj.pval3 * BOij_p4), fn7ij = 1.0 - exp3ij;
      const double BOjk_p4 = exp(ap.pval4 * log(BOjk)), exp3jk = exp(-aj.pval3 * BOjk_p4), fn7jk = 1.0 - exp3jk;
      const double exp7 = exp(-ap.pval7 * delta_ang), trm8 = 1.0 + exp6 + exp7;
      const double fn8j = aj.pval5 - (aj.pval5 - 1.0) * (2.0 + exp6) / trm8;
      const double theta0 = PI_ - ap.theta00 * (1.0 - ex10);
      const double theta_diff = theta0 - theta_ijk;
      const double exp2 = exp(-ap.pval2 * theta_diff * theta_diff);
      e5 += fn7ij * fn7jk * fn8j * (ap.pval1 - ap.pval1 * exp2);
      const double Cf7ij = aj.pval3 * ap.pval4 * (BOij_p4 / BOij) * exp3ij;                 // BO**(pval4-1) = BO**pval4 / BO
      const double Cf7jk = aj.pval3 * ap.pval4 * (BOjk_p4 / BOjk) * exp3jk;
      const double Cf8j = (1.0 - aj.pval5) / (trm8 * trm8) * (ff.pval6 * exp6 * trm8 - (2.0 + exp6) * (ff.pval6 * exp6 - ap.pval7 * exp7));
      const double Ctheta0 = ff.pval10 * ap.theta00 * ex10;
      const double CEval1 = Cf7ij * fn7jk * fn8j * ap.pval1 * (1.0 - exp2);
      const double CEval2 = fn7ij * Cf7jk * fn8j * ap.pval1 * (1.0 - exp2);
      const double CEval3 = fn7ij * fn7jk * Cf8j * ap.pval1 * (1.0 - exp2);
      const double CEval4 = 2.0 * ap.pval1 * ap.pval2 * fn7ij * fn7jk * fn8j * exp2 * theta_diff;
      const double CEval5 = CEval4 * Ctheta0 * CSBO2;
      const double CEval6 = CEval5 * dSBO1, CEval7 = CEval5 * dSBO2, CEval8 = CEval4 / sin_ijk;
      const double PEpen = ap.ppen1 * fn9 * epen[oi] * epen[ok];             // penalty, pot.F90:460-476
      e6 += PEpen;
      const double CEpen1 = Cf9j_fn9 * PEpen, CEpen2 = -2.0 * ff.ppen2 * (BOij - 2.0) * PEpen, CEpen3 = -2.0 * ff.ppen2 * (BOjk - 2.0) * PEpen;
      const double sum_BOi = delta[i] + ff.atom[ti].Val, sum_BOk = delta[k] + ff.atom[tk].Val;   // three-body conjugation, pot.F90:479-497
      const double ui = -BOij + sum_BOi, uk = -BOjk + sum_BOk;
      const double PEcoa = ap.pcoa1 / (1.0 + exp_coa2) * ecoa[oi] * ecoa[ok];
      e7 += PEcoa;
      const double CEcoa1 = -2.0 * ff.pcoa4 * (BOij - 1.5) * PEcoa, CEcoa2 = -2.0 * ff.pcoa4 * (BOjk - 1.5) * PEcoa;
      const double CEcoa3 = -ff.pcoa2 * exp_coa2 / (1.0 + exp_coa2) * PEcoa;
      const double CEcoa4 = -2.0 * ff.pcoa3 * ui * PEcoa, CEcoa5 = -2.0 * ff.pcoa3 * uk * PEcoa;
      V3 fi, fk;
      angle_forces(CEval8, rij, nij, rjk, njk, fi, fk);
      const int ri = oi - ob_w, rk = ok - ob_w;                             // accumulate, pot.F90:509-541
      atomicAdd(&s_acc[w][0][ri], CEpen2 + CEcoa1 - CEcoa4 + CEval1); atomicAdd(&s_acc[w][1][ri], CEcoa4);
      atomicAdd(&s_acc[w][2][ri], fi.x); atomicAdd(&s_acc[w][3][ri], fi.y); atomicAdd(&s_acc[w][4][ri], fi.z);
      atomicAdd(&s_acc[w][0][rk], CEpen3 + CEcoa2 - CEcoa5 + CEval2); atomicAdd(&s_acc[w][1][rk], CEcoa5);
      atomicAdd(&s_acc[w][2][rk], fk.x); atomicAdd(&s_acc[w][3][rk], fk.y); atomicAdd(&s_acc[w][4][rk], fk.z);
      atomicAdd(&s_at[w][a][0], CEpen1 + CEcoa3 + CEval3 + CEval7); atomicAdd(&s_at[w][a][1], CEval6); atomicAdd(&s_at[w][a][2], CEval5);
    }
    wave_lds_sync();
  };
#ifdef RXMD_EXPERIMENTS
  if (probe == 1) continue;
#endif
  // enumeration: LPA = 64 / NA lanes per atom; lane s of an atom takes the first bonds u = s, s + LPA, ... of its qualifying list, each with every later one
  {
    const int a = lane / LPA, sub = lane % LPA;
    const int nc = s_ci[w][a][3], tj = s_ci[w][a][0], ob = s_ci[w][a][1];
    int u = sub, v = sub + 1;
    for (;;) {
      bool live = u < nc - 1;
      const unsigned long long any = __ballot(live);
      if (any == 0ULL) break;
      bool go = false;
      int key = 0;
      if (live) {
        const int i1 = s_cap[w][a * CAPW + u], k1 = s_cap[w][a * CAPW + v];
        const int oi = ob + i1, ok = ob + k1;
        const float pr = s_cbo[w][a * E3Q_MAXL + u] * s_cbo[w][a * E3Q_MAXL + v];                   // pot.F90:397, BO(i,j) BO(j,k) > cut-off: decided in single precision except within 1e-3 of the
        go = pr > 1.001e-4f;                                                // cut-off, where the double-precision product from memory decides (the same answers)
        if (!go && pr > 0.999e-4f) go = bo0[oi] * bo0[ok] > cutof2_esub;
        if (go) go = s_ix3[(static_cast<int>(btype[oi]) * ff.n1 + tj) * ff.n1 + static_cast<int>(btype[ok])] != 0;
        key = (a << 10) | (i1 << 5) | k1;
        if (++v >= nc) { u += LPA; v = u + 1; }
      }
      const unsigned long long m = __ballot(go);
      if (go) s_q[w][(qh + qn + __popcll(m & ((1ULL << lane) - 1ULL))) & 127] = static_cast<unsigned short>(key);
      qn += __popcll(m);
      wave_lds_sync();
      if (qn >= 64) { evaluate(64); qh = (qh + 64) & 127; qn -= 64; }
    }
    if (qn > 0) evaluate(qn);
  }
  // write-out: every bond of the wavefront's atoms, coalesced; then the centres' own forces
  for (int r = lane; r < nb_w; r += 64) {
    const int o = ob_w + r, a = s_own[w][r];
    const double S_d1 = s_at[w][a][0], S_v6 = s_at[w][a][1], S_v5 = s_at[w][a][2];
    const double c1 = s_acc[w][0][r] + (S_d1 + S_v6 * ipow7(bo0[o]));
    if (c1 != 0.0) cf1[o] += c1;
    if (S_v5 != 0.0) { cf2[o] += S_v5; cf3[o] += S_v5; }
    const double cd = s_acc[w][1][r], f0 = s_acc[w][2][r], f1 = s_acc[w][3][r], f2 = s_acc[w][4][r];
    if (cd != 0.0) cdn[o] += cd;
    if (f0 != 0.0 || f1 != 0.0 || f2 != 0.0) { fnx[o] += f0; fny[o] += f1; fnz[o] += f2; }
  }
  if (lane < na) {                                 // the centre's force: minus the sum of the forces on its neighbours (ForceA3 books -(fi + fk) on j)
    const int j = jbase + lane, r0 = s_ci[w][lane][1] - ob_w, nj = s_ci[w][lane][2];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int t = 0; t < nj; ++t) { s0 += s_acc[w][2][r0 + t]; s1 += s_acc[w][3][r0 + t]; s2 += s_acc[w][4][r0 + t]; }
    if (s0 != 0.0 || s1 != 0.0 || s2 != 0.0) { fx[j] -= s0; fy[j] -= s1; fz[j] -= s2; }
  }
  wave_lds_sync();                                 // the next group rebuilds the tables
  }   // groups
  block_energy_add(e5, pe + 5); block_energy_add(e6, pe + 6); block_energy_add(e7, pe + 7);
}

struct BoxImg { double H[9], Hi[9], L[3]; int ortho; int probe; };   // lattice vectors for the image test of the torsion's stress correction; probe: RXMD_EXPERIMENTS builds only (RXMD_E4B_PROBE: 1 = set-up only, 2 = phase B skipped, 5 = PE(8), PE(9) count batches and entries)
// Torsion + four-body conjugation.  The reference walks centre bonds j-k with gid(j) < gid(k) and scatters to i,j,k,l.
// Here ONE WAVEFRONT owns several consecutive centre atoms; every bond slot of an atom has a lane that owns its accumulators, and
// one more lane stands for the atom itself.  Four layouts (template LSL, chosen per step from the longest bond list, see
// Engine::bonded_energies): two atoms x 32 lanes (any list, MAXNEIGHBS = 30), four atoms x 16 lanes, and the packed forms -- eight
// (lists <= 15) or four atoms with their lanes laid end to end.  More atoms per wavefront = fewer wavefronts that set up and
// enumerate, more of their lanes busy, fuller 64-lane batches (two RDX atoms queue 52 torsions: 1.38 batches per wavefront at 58 %
// of the lanes; eight atoms: 6.3 -> 3.9 ms).  Phase A enumerates every (k1,i1,l1) combination of the atoms, applies the reference's
// cheap bond-order cut-offs (pot.F90:1023,1044,1072,1078,1081) and compacts the survivors with a ballot into an LDS queue; phase B
// evaluates 64 surviving torsions at a time with every lane busy (only ~1 in 8 combinations survives in RDX).
// ONCE = false (until round 6): a torsion is visited from both ends: a lane books the energy, the j-k bond coefficient and the cdbnd
// terms only when gid(j) < gid(k) (the reference's orientation, which the index-ordered ccbnd rule depends on), the forces on i and j and
// the i-j bond coefficient always; the k/l side is booked when atom k is the centre.  Per-bond sums are formed by each
// evaluating lane adding its results to the (atom, slot) accumulators in LDS (ds_add_f64): no global atomics, and the order of the
// additions is fixed by the queue order and the lane order inside one LDS instruction.
// ONCE = true (default since round 6): only the centre bonds with gid(j) < gid(k) are enumerated -- the reference's loop, pot.F90:1021 -- and a
// visit books all four atoms: i, j and the centre bond as before, the force on k into the slot of k (k IS a neighbour of j), and the k-l side
// (ForceB coefficient of bond k-l, force on l), which belongs to a bond slot of ANOTHER wavefront's atom, through a delivery table: the queue runs
// with i fastest, so the torsions of one (centre bond, l) are neighbouring lanes; a batch is cut where such a run ends, the run's sum is formed by
// a fixed shuffle tree, and its first lane stores it at [bond (k, l1)][slot of j in the list of k] with a flag byte.  k_e4b_deliver then adds
// the flagged columns of every bond in slot order: no atomics, the same bits run to run.  Half the enumeration, 12.9 instead of 25.8
// evaluations per atom (RDX), no stress correction (all four forces are booked in the frame of j, as in the reference).
template <int LSL, bool ONCE, int WPB>
__global__ void __launch_bounds__(64 * WPB, E4B_MINB) k_e4b(int N, DevFF ff, const int *__restrict__ boff, const int *__restrict__ nbr, const unsigned char *__restrict__ btype, const int *__restrict__ nbrcnt, const int *__restrict__ type,
                                              const long long *__restrict__ gid, const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
                                              const double *__restrict__ bo0, const double *__restrict__ bo2, const double *__restrict__ delta,
                                              const double *__restrict__ etor, const double *__restrict__ econ,
                                              double *__restrict__ cf1, double *__restrict__ cf2, double *__restrict__ cdn,
                                              double *__restrict__ fnx, double *__restrict__ fny, double *__restrict__ fnz,
                                              double *__restrict__ cds, double *__restrict__ fx, double *__restrict__ fy, double *__restrict__ fz, double *__restrict__ pe, BoxImg bx,
                                              const int *__restrict__ brev, double4 *__restrict__ tkl, unsigned char *__restrict__ tflag, int TW, int VB) {
  // per (atom g, slot): the bond as seen from the centre
  __shared__ double s_bo[WPB][64], s_et[WPB][64], s_ec[WPB][64], s_rx[WPB][64], s_ry[WPB][64], s_rz[WPB][64], s_rn[WPB][64];
  __shared__ int s_nb[WPB][64], s_meta[WPB][64];     // s_meta: type of the neighbour | its bond count << 8 | "this centre owns the bond" << 16
  __shared__ int s_bofn[WPB][64];                  // first bond (CSR offset) of the neighbour in this slot: its list is bonds s_bofn .. + count
  // per (atom g, slot) as CENTRE bond k1: factors shared by all torsions around it; a bond count of 0 marks "no torsion through this bond"
  __shared__ double s_btb2[WPB][64], s_dfn11[WPB][64];
  __shared__ int s_q[WPB][128];                    // queue of surviving combinations: g<<15 | k1<<10 | i1<<5 | l1
  __shared__ int s_ql[WPB][128]; __shared__ unsigned short s_qx[WPB][128];     // ... with atom l and the torsion row of (i, j, k, l): phase A has both at hand (round 5: phase B fetched them again, four dependent round trips in front of its arithmetic)
  // phase A walks only bonds above the cut-off: the slots of each centre atom that qualify, and per centre bond the qualifying slots
  // of k with what the filter needs of them (bond order, atom l, its type) -- staged once per centre bond by the lanes side by side
  __shared__ int s_cap[WPB][64], s_capl[WPB][64], s_ll[WPB][64], s_tl[WPB][64], s_gj[WPB][8][2];   // s_gj: bond count and type of each centre atom
  __shared__ int s_cb[WPB][64];                    // the centre bonds of the pass: atom << 8 | lane of the slot
  __shared__ int s_cd[WPB][4][9];                  // per centre bond of a round: atom, lane, first lane of the atom, type of j, k, j, number of qualifying k-slots, 1/that, combinations
  __shared__ int s_base[WPB][9];                   // LSL == 0: first lane of each atom of the pass, and the end
  __shared__ double s_bokl[WPB][64];
  // per (atom g, slot) accumulators, updated with LDS atomics by the lanes that evaluate torsions: [0] cf1 and [1..3] force of the
  // i-j bond / its neighbour, [4] cf1, [5] cf2, [6] cdbnd of the centre bond; slot 31 (never a bond) holds the centre atom's own
  // force in [1..3] and cdbnd in [6].  (An earlier version wrote 11 results per torsion to LDS and let every owner lane scan all 64
  // of them: that scan cost about as many instructions as the torsion itself.)
  __shared__ double s_acc[WPB][64][7];
  // LSL = 5 / 4: two / four atoms with a fixed range of 32 / 16 lanes each.  LSL = 0 / 1 ("packed"): eight atoms with lists <= 15 /
  // four atoms with any list, each with as many lanes as it has bonds plus one, laid end to end -- RDX atoms have 5.3 bonds on
  // average, so the set-up and the enumeration run with most lanes busy instead of a third; the few groups that need more than 64
  // lanes (one group of eight in twenty in the RDX crystal) are done in two passes of half the atoms each.
  constexpr bool PACK = (LSL <= 1);
  constexpr int PKN = (LSL == 0) ? 8 : 4, CAP = (LSL == 0) ? 15 : 31;                          // packed: atoms per wavefront, longest list
  constexpr int SL = PACK ? CAP + 1 : (1 << (LSL & 31)), NG = PACK ? PKN : (64 >> (LSL & 31));  // (widest) lane range of an atom, atoms per wavefront
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));   // wave-uniform -> scalar registers
  // "is there a torsion row for these four types" (pot.F90:1078) as a bit table in LDS when the ffield has at most 7 atom types
  // (n1^4 <= 4096 bits): the enumeration asks it for every candidate, and a look-up in global memory is a dependent round trip
  // per centre bond
  extern __shared__ unsigned char s_tor[];        // the torsion row of every type quadruple (0: none), n1^4 <= 4096 entries of a force field with at most 255 rows (dynamic: e4b_tor_bytes)
  const bool tor_lds = ff.n1 <= 8 && ff.ntoty <= 255;
  if (tor_lds) {
    const int n4 = ff.n1 * ff.n1 * ff.n1 * ff.n1;
    for (int t = threadIdx.x; t < n4; t += 64 * WPB) s_tor[t] = static_cast<unsigned char>(ff.inxn4[t]);
    __syncthreads();
  }
  double e8 = 0.0, e9 = 0.0;                      // energies: per evaluating lane, summed over the wave at the end
  // The grid may be smaller than the number of atom groups (persistent: as many workgroups as fit the device at once, each walking every
  // gridDim.x-th group): a group is a "virtual workgroup" vb of VB; the XCD-aware order is that of the full grid (gridDim.x is a multiple of 8, so a
  // workgroup stays on the eighth of the atoms its XCD owns).
  for (int vb = blockIdx.x; vb < VB; vb += gridDim.x) {
  const int jbase = (xcd_swizzle(vb, VB) * WPB + w) * NG;
  if (jbase >= N) continue;                      // whole wavefront skips together; no block-level barrier below
  int npass = 1, c_me = 0, cpre = 0;
  // (round 6) packed forms: what the slot lanes need of their centre atom -- first bond, type, position, delta, global id -- is loaded by the atom's lane
  // WITH the bond count and handed over through LDS -- rows 0..7 of the slot accumulators, which the set-up clears only after it has read them --: one round
  // trip less in front of the bond data than loading it by j_me once the lanes are laid out (the rare second pass of a group loads it that way)
  if (PACK) {                                     // lanes 0..PKN-1: lanes the atom needs, inclusive prefix
    if (lane < PKN && jbase + lane < N) {
      const int ja = jbase + lane;
      c_me = min(nbrcnt[ja], CAP) + 1;
      s_acc[w][lane][0] = x[ja]; s_acc[w][lane][1] = y[ja]; s_acc[w][lane][2] = z[ja]; s_acc[w][lane][3] = delta[ja];
      s_acc[w][lane][4] = __longlong_as_double(gid[ja]); s_acc[w][lane][5] = __hiloint2double(boff[ja], type[ja]);
    }
    cpre = c_me;
#pragma unroll
    for (int o = 1; o < PKN; o <<= 1) { const int t = __shfl_up(cpre, o, 64); if (lane >= o) cpre += t; }
    npass = (__shfl(cpre, PKN - 1, 64) <= 64) ? 1 : 2;
  }
  auto gbase = [&](int g) { return PACK ? s_base[w][g] : (g << (LSL & 31)); };       // first lane of atom g of this pass
  for (int pass = 0; pass < npass; ++pass) {
  const int g0 = PACK ? (npass == 1 ? 0 : (PKN / 2) * pass) : 0, ng = PACK ? (npass == 1 ? PKN : PKN / 2) : NG;   // atoms of this pass
  int g_me, sl_me, nj_me, j_me;
  bool has_me;
  if (PACK) {
    const int off0 = __shfl(cpre - c_me, g0, 64);
    if (lane >= g0 && lane < g0 + ng) {
      s_base[w][lane - g0] = cpre - c_me - off0;
      s_gj[w][lane - g0][0] = max(c_me - 1, 0); s_gj[w][lane - g0][1] = 1;   // (an atom behind the last one has no lane to write these)
    }
    if (lane == g0 + ng - 1) s_base[w][ng] = cpre - off0;
    wave_lds_sync();
    g_me = 0;
    for (int g = 1; g < ng; ++g) g_me += (lane >= s_base[w][g]) ? 1 : 0;
    has_me = lane < s_base[w][ng];
    sl_me = has_me ? lane - s_base[w][g_me] : 1;
    nj_me = has_me ? s_base[w][g_me + 1] - s_base[w][g_me] - 1 : 0;
    j_me = jbase + g0 + g_me;
  } else {
    g_me = lane >> (LSL & 31); sl_me = lane & (SL - 1);
    j_me = jbase + g_me;
    has_me = j_me < N;
    nj_me = has_me ? min(nbrcnt[j_me], SL - 1) : 0;
  }
  const bool self_me = has_me && (PACK ? sl_me == nj_me : sl_me == SL - 1);   // the lane that stands for the atom itself (never a bond)
  int tj_me, ob_me; double xj_me, yj_me, zj_me, dj_me; long long gj_me;
  if (PACK && pass == 0) {                        // (behind the wave_lds_sync of the lane layout above)
    const int src = g0 + g_me;
    const double ot = s_acc[w][src][5];
    tj_me = has_me ? __double2loint(ot) : 1; ob_me = has_me ? __double2hiint(ot) : 0;
    xj_me = s_acc[w][src][0]; yj_me = s_acc[w][src][1]; zj_me = s_acc[w][src][2]; dj_me = s_acc[w][src][3]; gj_me = __double_as_longlong(s_acc[w][src][4]);
  } else {
    tj_me = has_me ? type[j_me] : 1;
    xj_me = has_me ? x[j_me] : 0.0; yj_me = has_me ? y[j_me] : 0.0; zj_me = has_me ? z[j_me] : 0.0;
    ob_me = has_me ? boff[j_me] : 0;              // bonds of an atom are consecutive (CSR): the lanes of this wavefront read consecutive addresses
    dj_me = has_me ? delta[j_me] : 0.0; gj_me = has_me ? gid[j_me] : 0;
  }
  s_meta[w][lane] = 0;
  bool cap_me = false, cb_me = false;              // this slot qualifies as an i / as a centre bond of this wavefront (ONCE: only in the reference's orientation gid(j) < gid(k))
  if (sl_me < nj_me) {
    const int o = ob_me + sl_me;
    const int i = nbr[o], ti = btype[o];
    const double rx = x[i] - xj_me, ry = y[i] - yj_me, rz = z[i] - zj_me;      // r_i - r_j
    const double b = bo0[o];
    // what a centre bond needs of its far end and of the bond, requested with the rest and not behind the test of the bond order (one round trip less)
    const double d_i = delta[i], b2_o = bo2[o]; const long long g_i = gid[i]; const int bi = boff[i], nc_i = nbrcnt[i], br_o = ONCE ? brev[o] : 0;
    cap_me = b > cutof2_esub;
    s_nb[w][lane] = i;
    int meta = ti;
    s_bo[w][lane] = b; s_et[w][lane] = etor[o]; s_ec[w][lane] = econ[o];
    s_rx[w][lane] = rx; s_ry[w][lane] = ry; s_rz[w][lane] = rz; s_rn[w][lane] = sqrt(rx * rx + ry * ry + rz * rz);
    if (b > cutof2_esub) {                       // this slot as a centre bond j-k (pot.F90:1023)
      const DevAtomP aj = ff.atom[tj_me];
      const double delta_ang_jk = (dj_me + aj.Val - aj.Valangle) + (d_i + ff.atom[ti].Val - ff.atom[ti].Valangle);
      const double exp_tor3 = exp(-ff.ptor3 * delta_ang_jk), exp_tor4 = exp(ff.ptor4 * delta_ang_jk);
      const double exp_tor34_i = 1.0 / (1.0 + exp_tor3 + exp_tor4);
      const double fn11 = (2.0 + exp_tor3) * exp_tor34_i;
      s_dfn11[w][lane] = (-ff.ptor3 * exp_tor3 + (ff.ptor3 * exp_tor3 - ff.ptor4 * exp_tor4) * (2.0 + exp_tor3) * exp_tor34_i) * exp_tor34_i;
      s_btb2[w][lane] = 2.0 - b2_o - fn11;
      const bool own = gj_me < g_i;
      meta |= (min(nc_i, 255) << 8) | (own ? 1 << 16 : 0);
      s_bofn[w][lane] = bi;
      if (ONCE) { meta |= (br_o - bi) << 17; cb_me = own; }   // the slot of j in the list of k (< MAXNEIGHBS = 30): where the k-l side of this centre bond is delivered
      else cb_me = true;
    }
    s_meta[w][lane] = meta;
  }
#pragma unroll
  for (int c = 0; c < 7; ++c) s_acc[w][lane][c] = 0.0;
  const unsigned long long capmask = __ballot(cap_me);
  {                                               // the qualifying slots of each centre atom, in slot order
    const int gb_me = gbase(g_me);
    const unsigned mine = static_cast<unsigned>(capmask >> gb_me) & ((SL == 32) ? 0xffffffffu : ((1u << (SL & 31)) - 1u));   // (bits of the next atom above sl_me do not matter)
    if (cap_me) s_cap[w][gb_me + __popc(mine & ((1u << sl_me) - 1u))] = sl_me;
    if (sl_me == 0 && (has_me || !PACK)) { s_gj[w][g_me][0] = nj_me; s_gj[w][g_me][1] = tj_me; }
  }
  wave_lds_sync();
  int qn = 0, qh = 0;                             // entries in the queue; its head (ONCE: a ring of 128, otherwise always 0)

  auto evaluate = [&](int cnt) {                  // phase B: the first cnt (<= 64) queue entries, one per lane
#ifdef RXMD_EXPERIMENTS
    if (bx.probe == 2) return;
    if (bx.probe == 5) { if (lane == 0) { atomicAdd(pe + 8, 1.0); atomicAdd(pe + 9, static_cast<double>(cnt)); } return; }
#endif
#ifdef RXMD_EXPERIMENTS
    const bool once_math = ONCE && bx.probe != 7;   // probe 7: the one-visit enumeration with the arithmetic of a two-visit evaluation (timing only)
#else
    constexpr bool once_math = ONCE;
#endif
    double o[7] = {0, 0, 0, 0, 0, 0, 0};
    V3 fself = {0.0, 0.0, 0.0};
    V3 fkk = {0.0, 0.0, 0.0}, fll = {0.0, 0.0, 0.0};   // ONCE: forces on k and on l
    double c3kl = 0.0;                                 // ONCE: ForceB(k-l, C4body_b(3)), pot.F90:1197-1198
    double cd_self = 0.0;
    int key = -1;
    if (lane < cnt) {
      key = s_q[w][(qh + lane) & 127];
      const int g = key >> 15, k1 = (key >> 10) & 31, i1 = (key >> 5) & 31, l1 = key & 31;
      const int gb = gbase(g);
      const int sk = gb + k1, si = gb + i1;
      const int k = s_nb[w][sk], mk_ = s_meta[w][sk], tk = mk_ & 255, tj = s_gj[w][g][1];
      const double BOjk_f = s_bo[w][sk], BOij_f = s_bo[w][si];
      const double BOjk = BOjk_f - cutof2_esub, BOij = BOij_f - cutof2_esub;
      const bool own = ONCE || (mk_ >> 16) != 0;
      const int ol = s_bofn[w][sk] + l1;
      const int l = s_ql[w][(qh + lane) & 127];
      const DevTorsP tp = ff.tors[s_qx[w][(qh + lane) & 127]];
      (void)tj; (void)tk;
      const V3 rjk = {-s_rx[w][sk], -s_ry[w][sk], -s_rz[w][sk]};          // r_j - r_k
      const double njk = s_rn[w][sk];
      const V3 rij = {s_rx[w][si], s_ry[w][si], s_rz[w][si]};
      const double nij = s_rn[w][si];
      const V3 rkl = {x[k] - x[l], y[k] - y[l], z[k] - z[l]};                 // r_k - r_l
      const double nkl = sqrt(dot(rkl, rkl));
      const double inij = 1.0 / nij, injk = 1.0 / njk, inkl = 1.0 / nkl;
      double cos_ijk = -dot(rij, rjk) * (inij * injk);
      if (cos_ijk > MAXANGLE) cos_ijk = MAXANGLE;
      if (cos_ijk < MINANGLE) cos_ijk = MINANGLE;
      const double sin_ijk = sqrt((1.0 - cos_ijk) * (1.0 + cos_ijk));       // sin(acos(c))
      const double tan_ijk_i = cos_ijk / sin_ijk;                            // 1/tan(theta)
      // cross_product(rij, rjk) and (rjk, rkl), pot.F90:1524-1543
      const V3 n1v = {rij.x * inij, rij.y * inij, rij.z * inij}, n2v = {rjk.x * injk, rjk.y * injk, rjk.z * injk}, n3v = {rkl.x * inkl, rkl.y * inkl, rkl.z * inkl};
      const V3 c1v = {n1v.y * n2v.z - n1v.z * n2v.y, n1v.z * n2v.x - n1v.x * n2v.z, n1v.x * n2v.y - n1v.y * n2v.x};
      const V3 c2v = {n2v.y * n3v.z - n2v.z * n3v.y, n2v.z * n3v.x - n2v.x * n3v.z, n2v.x * n3v.y - n2v.y * n3v.x};
      double nc1 = sqrt(dot(c1v, c1v)), nc2 = sqrt(dot(c2v, c2v));
      if (nc1 < NSMALL) nc1 = NSMALL;
      if (nc2 < NSMALL) nc2 = NSMALL;
      const double BOkl = bo0[ol] - cutof2_esub; (void)BOkl;
      const double et1 = s_et[w][si], et2 = s_et[w][sk], et3 = etor[ol];
      const double fn10 = (1.0 - et1) * (1.0 - et2) * (1.0 - et3);
      const double fn12 = s_ec[w][si] * s_ec[w][sk] * econ[ol];
      const double btb2 = s_btb2[w][sk];
      const double exp_tor1 = exp(tp.ptor1 * (btb2 * btb2));
      double cos_jkl = -dot(rjk, rkl) * (injk * inkl);
      if (cos_jkl > MAXANGLE) cos_jkl = MAXANGLE;
      if (cos_jkl < MINANGLE) cos_jkl = MINANGLE;
      const double sin_jkl = sqrt((1.0 - cos_jkl) * (1.0 + cos_jkl));
      const double tan_jkl_i = cos_jkl / sin_jkl;
      double c1 = dot(c1v, c2v) / (nc1 * nc2);
      if (c1 > MAXANGLE) c1 = MAXANGLE;
      if (c1 < MINANGLE) c1 = MINANGLE;
      const double c1sq = c1 * c1;
      const double cos_2w = 2.0 * c1sq - 1.0;                               // cos(2 acos c)
      const double c2 = 1.0 - cos_2w, c3 = 1.0 + (4.0 * c1sq - 3.0) * c1;   // 1 + cos(3 acos c)
      const double vsum = tp.V1 * (1.0 + c1) + tp.V2 * exp_tor1 * c2 + tp.V3 * c3;
      const double ss = sin_ijk * sin_jkl;
      const double PEconj = tp.pcot1 * fn12 * (1.0 + (c1sq - 1.0) * ss);
      if (own) { e8 += 0.5 * fn10 * ss * vsum; e9 += PEconj; }
      const double CEt1 = 0.5 * ss * vsum;
      const double CEt2 = -tp.ptor1 * fn10 * ss * tp.V2 * exp_tor1 * btb2 * c2;
      const double CEt3 = CEt2 * s_dfn11[w][sk];
      const double CEt4 = CEt1 * ff.ptor2 * et1 * (1.0 - et2) * (1.0 - et3);
      const double CEt5 = CEt1 * ff.ptor2 * (1.0 - et1) * et2 * (1.0 - et3);
      const double cmn = -0.5 * fn10 * vsum;
      const double CEt7 = cmn * sin_jkl * tan_ijk_i, CEt8 = cmn * sin_ijk * tan_jkl_i;
      const double CEt9 = fn10 * ss * (0.5 * tp.V1 - 2.0 * tp.V2 * exp_tor1 * c1 + 1.5 * tp.V3 * (cos_2w + 2.0 * c1sq));
      const double Cconj = -2.0 * ff.pcot2 * PEconj;
      const double CEc1 = Cconj * (BOij - 1.5), CEc2 = Cconj * (BOjk - 1.5);
      if (once_math) c3kl = Cconj * (BOkl - 1.5) + CEt1 * ff.ptor2 * (1.0 - et1) * (1.0 - et2) * et3;    // CEconj(3) + CEtors(6)
      const double CEc4 = -tp.pcot1 * fn12 * (c1sq - 1.0) * tan_ijk_i * sin_jkl;
      const double CEc5 = -tp.pcot1 * fn12 * (c1sq - 1.0) * sin_ijk * tan_jkl_i;
      const double CEc6 = 2.0 * tp.pcot1 * fn12 * c1 * ss;
      o[0] = CEc1 + CEt4;                                                   // ForceB(i-j, C4body_b(1)), pot.F90:1185-1186
      if (own) { o[4] = CEc2 + CEt5; o[5] = CEt2; o[6] = CEt3; if (!once_math) cd_self += CEt3; }   // pot.F90:1178-1194 (one visit: the centre's own cdbnd is the sum of its slots' terms, formed at the end)
      // angle i-j-k (coefficient C4body_a(1)): forces on i and j
      V3 fi, fk;
      angle_forces(CEc4 + CEt7, rij, nij, rjk, njk, fi, fk);
      o[1] = fi.x; o[2] = fi.y; o[3] = fi.z;
      if (once_math) fkk = fk;                                               // (one visit: the force on j is minus the sum of the other three, formed at the end)
      else { fself.x -= fi.x + fk.x; fself.y -= fi.y + fk.y; fself.z -= fi.z + fk.z; }
      // angle j-k-l (coefficient C4body_a(2)): j is the first atom
      V3 fj2, fl2;
      angle_forces(CEc5 + CEt8, rjk, njk, rkl, nkl, fj2, fl2);
      if (once_math) { fkk.x -= fj2.x + fl2.x; fkk.y -= fj2.y + fl2.y; fkk.z -= fj2.z + fl2.z; fll = fl2; }
      else { fself.x += fj2.x; fself.y += fj2.y; fself.z += fj2.z; }
      // dihedral (ForceA4, pot.F90:1369-1459): forces on i (fij) and j (-fij + fjk)
      {
        const double coeff = CEc6 + CEt9;
        const double C00 = nij * nij, C01 = dot(rij, rjk), C02 = dot(rij, rkl), C11 = njk * njk, C12 = dot(rjk, rkl), C22 = nkl * nkl;
        const double D0 = C00 * C11 - C01 * C01, Dm1 = C11 * C22 - C12 * C12;
        const double coDD = coeff * (1.0 / sqrt(D0 * Dm1));
        const double com = C01 * C12 - C02 * C11;
        const double cD0 = com / D0, cDm = com / Dm1;
        const double Cwi1 = C11 * cD0, Cwi2 = -(C12 + C01 * cD0), Cwi3 = C11;
        const double Cwj1 = -(C12 + (C11 + C01) * cD0);
        const double Cwj2 = -(-C12 - 2 * C02 - C22 * cDm - (C00 + C01) * cD0);
        const double Cwj3 = -(C01 + C11 + C12 * cDm);
        const V3 fij = {coDD * (Cwi1 * rij.x + Cwi2 * rjk.x + Cwi3 * rkl.x), coDD * (Cwi1 * rij.y + Cwi2 * rjk.y + Cwi3 * rkl.y),
                        coDD * (Cwi1 * rij.z + Cwi2 * rjk.z + Cwi3 * rkl.z)};
        const V3 fjk = {coDD * ((Cwj1 + Cwi1) * rij.x + (Cwj2 + Cwi2) * rjk.x + (Cwj3 + Cwi3) * rkl.x),
                        coDD * ((Cwj1 + Cwi1) * rij.y + (Cwj2 + Cwi2) * rjk.y + (Cwj3 + Cwi3) * rkl.y),
                        coDD * ((Cwj1 + Cwi1) * rij.z + (Cwj2 + Cwi2) * rjk.z + (Cwj3 + Cwi3) * rkl.z)};
        o[1] += fij.x; o[2] += fij.y; o[3] += fij.z;
        if (!once_math) { fself.x += -fij.x + fjk.x; fself.y += -fij.y + fjk.y; fself.z += -fij.z + fjk.z; }
        if (once_math) {                                                     // k gets -fjk + fkl, l gets -fkl (pot.F90:1417-1452)
          const double Cwl2 = C01 + C12 * cDm, Cwl3 = -(C11 * cDm);         // Cwl1 = -C11
          const V3 fkl = {-coDD * (-C11 * rij.x + Cwl2 * rjk.x + Cwl3 * rkl.x), -coDD * (-C11 * rij.y + Cwl2 * rjk.y + Cwl3 * rkl.y),
                          -coDD * (-C11 * rij.z + Cwl2 * rjk.z + Cwl3 * rkl.z)};
          fkk.x += -fjk.x + fkl.x; fkk.y += -fjk.y + fkl.y; fkk.z += -fjk.z + fkl.z;
          fll.x -= fkl.x; fll.y -= fkl.y; fll.z -= fkl.z;
        }
      }
      // stress: this visit books f_i and f_j in the frame of centre j; f_k and f_l are booked by the visit whose centre is the
      // OWNER of k.  When k is an image beyond the periodic box that owner sits one lattice vector T away, and the virial
      // sum_a pos_a f_a (pot.F90:65-72) would hold T (f_k + f_l) = -T (f_i + f_j) too much.  The owner's visit sees j as an
      // image at -T and finds the same product, so each of the two visits takes out half (rare lanes only).
      if (!ONCE) {
        const double xk = x[k], yk = y[k], zk = z[k];
        // k is bonded to a resident: it lies inside the box or in the first image layer, so comparisons name the lattice vector
        // (floor(x / L) costs three FP64 divisions per torsion)
        double t0, t1, t2;
        if (bx.ortho) {
          t0 = xk < 0.0 ? bx.L[0] : (xk >= bx.L[0] ? -bx.L[0] : 0.0); t1 = yk < 0.0 ? bx.L[1] : (yk >= bx.L[1] ? -bx.L[1] : 0.0);
          t2 = zk < 0.0 ? bx.L[2] : (zk >= bx.L[2] ? -bx.L[2] : 0.0);
        } else {                                   // skewed box: which lattice vectors bring k home, from its normalised coordinates
          const double n0 = -floor(bx.Hi[0] * xk + bx.Hi[1] * yk + bx.Hi[2] * zk), n1 = -floor(bx.Hi[3] * xk + bx.Hi[4] * yk + bx.Hi[5] * zk),
                       n2 = -floor(bx.Hi[6] * xk + bx.Hi[7] * yk + bx.Hi[8] * zk);
          t0 = bx.H[0] * n0 + bx.H[1] * n1 + bx.H[2] * n2; t1 = bx.H[3] * n0 + bx.H[4] * n1 + bx.H[5] * n2; t2 = bx.H[6] * n0 + bx.H[7] * n1 + bx.H[8] * n2;
        }
        if (t0 != 0.0 || t1 != 0.0 || t2 != 0.0) {
          const double F0 = 0.5 * (o[1] + fself.x), F1 = 0.5 * (o[2] + fself.y), F2 = 0.5 * (o[3] + fself.z);
          if (t0 != 0.0) { atomicAdd(pe + 16, t0 * F0); atomicAdd(pe + 21, t0 * F1); }     // xx, xy
          if (t1 != 0.0) { atomicAdd(pe + 17, t1 * F1); atomicAdd(pe + 19, t1 * F2); }     // yy, yz
          if (t2 != 0.0) { atomicAdd(pe + 18, t2 * F2); atomicAdd(pe + 20, t2 * F0); }     // zz, zx
        }
      }
    }
#ifdef RXMD_EXPERIMENTS
    if (ONCE && bx.probe != 3 && bx.probe != 7)
#else
    if (ONCE)
#endif
    {
      // the k-l side: C4body_b(3) and the force on l, summed over the i of one (centre bond, l) run -- neighbouring lanes of this batch (a fixed
      // tree: the same bits run to run) -- and stored ONCE by the run's first lane where the bond k-l finds it: row = bond (k, l1), column = the
      // slot of j in the list of k.  k_e4b_deliver adds the columns of a row in slot order.
      const int seg = (key >= 0) ? (key & E4B_SEG) : (-1 - lane);
      for (int d = 1; d < 32; d <<= 1) {
        const int sd = __shfl_down(seg, d, 64);                              // (every lane takes part in the exchange: no short circuit in front of it)
        const bool same = (lane + d < 64) && sd == seg;
        if (__ballot(same) == 0ULL) break;                                  // runs are contiguous: none at this distance, none beyond
        const double u0 = __shfl_down(c3kl, d, 64), u1 = __shfl_down(fll.x, d, 64), u2 = __shfl_down(fll.y, d, 64), u3 = __shfl_down(fll.z, d, 64);
        if (same) { c3kl += u0; fll.x += u1; fll.y += u2; fll.z += u3; }
      }
      const int segup = __shfl_up(seg, 1, 64);
      if (key >= 0 && (lane == 0 || segup != seg)) {
        const int g = key >> 15, sk = gbase(g) + ((key >> 10) & 31);
        const size_t ti = static_cast<size_t>(s_bofn[w][sk] + (key & 31)) * TW + ((s_meta[w][sk] >> 17) & 31);
#ifdef RXMD_EXPERIMENTS
        if (bx.probe != 6)
#endif
        {   // (written once, read once by another kernel: past the L2's write-allocate)
          typedef double d4v __attribute__((ext_vector_type(4)));
          d4v tv; tv.x = c3kl; tv.y = fll.x; tv.z = fll.y; tv.w = fll.z;
          __builtin_nontemporal_store(tv, reinterpret_cast<d4v *>(tkl) + ti);
          __builtin_nontemporal_store(static_cast<unsigned char>(1), tflag + ti);
        }
        // the centre atom's own force is minus the sum of the three others of every torsion: the l part once per run, here; the i and k parts from
        // the slot accumulators at the end (four LDS atomics per lane less, the ones with the most lanes per address)
        double *as = &s_acc[w][PACK ? s_base[w][g + 1] - 1 : gbase(g) + SL - 1][0];
        atomicAdd(as + 1, -fll.x); atomicAdd(as + 2, -fll.y); atomicAdd(as + 3, -fll.z);
      }
    }
#ifdef RXMD_EXPERIMENTS
    if (bx.probe == 4) { if (o[0] + o[1] + o[2] + o[3] + o[4] + o[5] + o[6] + fkk.x + fkk.y + fkk.z + fself.x + fself.y + fself.z + cd_self + c3kl + fll.x == 1.2345e300) e8 += 1.0; key = -1; }
#endif
    if (key >= 0) {
      const int g = key >> 15, k1 = (key >> 10) & 31, i1 = (key >> 5) & 31;
      const int gb = gbase(g), gself = PACK ? s_base[w][g + 1] - 1 : gb + SL - 1;
      double *ai = &s_acc[w][gb + i1][0], *ak = &s_acc[w][gb + k1][0], *as = &s_acc[w][gself][0];
      atomicAdd(ai + 0, o[0]); atomicAdd(ai + 1, o[1]); atomicAdd(ai + 2, o[2]); atomicAdd(ai + 3, o[3]);
      atomicAdd(ak + 4, o[4]); atomicAdd(ak + 5, o[5]); atomicAdd(ak + 6, o[6]);
      if (once_math) { atomicAdd(ak + 1, fkk.x); atomicAdd(ak + 2, fkk.y); atomicAdd(ak + 3, fkk.z); }     // k is the neighbour in slot k1: its force rides where f_i does
      if (!once_math) { atomicAdd(as + 1, fself.x); atomicAdd(as + 2, fself.y); atomicAdd(as + 3, fself.z); atomicAdd(as + 6, cd_self); }
    }
    wave_lds_sync();
  };

#ifdef RXMD_EXPERIMENTS
  if (bx.probe == 1) break;                       // timing experiment: set-up only
#endif
  // phase A: enumerate, filter, compact.  The centre bonds of the pass (every slot above the cut-off, in (atom, slot) order) are taken
  // CPB at a time: the lanes stage the qualifying slots of their k atoms (pot.F90:1072: bond order, atom l, its type -- three
  // dependent loads) side by side, KW lanes per centre bond, so that a wavefront of eight atoms with ~40 centre bonds waits for ~10
  // such rounds instead of 40 (one centre bond at a time: set-up + enumeration 2.43 ms of the 3.96 ms kernel); the enumeration of a
  // centre bond then reads LDS only.
  constexpr int KW = (LSL == 5 || LSL == 1) ? 32 : 16, CPB = 64 / KW;   // lanes per centre bond (longest list of any atom + 1), centre bonds per round
  const unsigned long long cbmask = ONCE ? __ballot(cb_me) : capmask;
  const int ncb = __popcll(cbmask);
  if (cb_me) { const int r = __popcll(cbmask & ((1ULL << lane) - 1ULL)); s_cb[w][r] = (g_me << 8) | lane; }
  wave_lds_sync();
  // The k-side list of a round -- bond order, atom l and its type of every slot of k, KW lanes per centre bond -- is requested ONE ROUND AHEAD into
  // three registers (round 6): all three loads at once, whatever the bond order turns out to be, and behind them the enumeration and evaluation of the
  // round before.  (Until then: the bond order, then -- for the slots above the cut-off -- l and its type: two dependent round trips at the head of
  // every round, ~10 rounds per wavefront.)
  double nx_bl = 0.0; int nx_l = 0, nx_tl = 0;
  auto fetch_round = [&](int r0) {
    const int cbl = lane / KW, ks = lane % KW, r = r0 + cbl;
    nx_bl = 0.0; nx_l = 0; nx_tl = 0;
    if (r < ncb) {
      const int sk = s_cb[w][r] & 255;
      const int bk = s_bofn[w][sk];                 // first bond of atom k (the far end of centre bond r): its slots are bonds bk, bk + 1, ...
      if (ks < min((s_meta[w][sk] >> 8) & 255, KW == 32 ? WSLOT : KW)) { nx_bl = bo0[bk + ks]; nx_l = nbr[bk + ks]; nx_tl = btype[bk + ks]; }
    }
  };
  fetch_round(0);
  for (int r0 = 0; r0 < ncb; r0 += CPB) {
    unsigned long long ml;
    {
      const int cbl = lane / KW, ks = lane % KW;
      const double bl = nx_bl; const int l = nx_l, tl = nx_tl;
      if (r0 + CPB < ncb) fetch_round(r0 + CPB);
      // a slot of k stays in the round's list only if it can be the k-l bond of THIS centre bond: above the cut-off, BO(j,k) BO(k,l) above it, l not j
      // (pot.F90:1072,1081 -- tests that do not depend on i: applied here, once per slot, instead of once per (i, l) combination)
      bool capl = bl > cutof2_esub;
      if (r0 + cbl < ncb) {
        const int ent = s_cb[w][r0 + cbl];
        capl = capl && (s_bo[w][ent & 255] * bl > cutof2_esub) && (l != jbase + g0 + (ent >> 8));
      }
      ml = __ballot(capl);
      if (capl) {
        const unsigned sub = static_cast<unsigned>(ml >> (cbl * KW)) & ((KW == 32) ? 0xffffffffu : 0xffffu);
        const int pos = cbl * KW + __popc(sub & ((1u << ks) - 1u));
        s_capl[w][pos] = ks; s_bokl[w][pos] = bl; s_ll[w][pos] = l; s_tl[w][pos] = tl;
      }
    }
    // what the combinations of each centre bond of the round need, written by lanes 0..CPB-1: the combinations of all of them are
    // then enumerated together, 64 at a time (a centre bond of RDX has ~25: one pass per centre bond left most lanes idle)
    if (lane < CPB) {
      int T = 0;
      if (r0 + lane < ncb) {
        const int ent = s_cb[w][r0 + lane];
        const int g = ent >> 8, sk = ent & 255, gb = gbase(g), nj = s_gj[w][g][0];
        const int gw = PACK ? nj + 1 : SL;                                  // lanes of this atom
        const int ncj = __popc(static_cast<unsigned>(capmask >> gb) & ((gw == 32) ? 0xffffffffu : ((1u << (gw & 31)) - 1u)));
        const int nck = __popc(static_cast<unsigned>(ml >> (lane * KW)) & ((KW == 32) ? 0xffffffffu : 0xffffu));
        int *d = &s_cd[w][lane][0];
#ifdef RXMD_EXPERIMENTS
        const int dv = (ONCE && bx.probe != 8) ? ncj : nck;                 // probe 8: one-visit enumeration in the two-visit order, batches of 64 (timing only)
#else
        const int dv = ONCE ? ncj : nck;
#endif                                    // the index that runs fastest: l (two visits) / i (one visit: the torsions of one (centre bond, l) are neighbours in the queue)
        d[0] = g; d[1] = sk; d[2] = gb; d[3] = s_gj[w][g][1]; d[4] = s_nb[w][sk]; d[5] = jbase + g0 + g; d[6] = dv;
        d[7] = __float_as_int(1.0f / static_cast<float>(max(dv, 1)));      // c / dv for c < 1024, dv <= 31: exact through (c + 0.5) * (1 / dv) in FP32
        T = ncj * nck;
      }
      s_cd[w][lane][8] = T;
    }
    wave_lds_sync();
    int P[CPB + 1];
    P[0] = 0;
#pragma unroll
    for (int c = 0; c < CPB; ++c) P[c + 1] = P[c] + __builtin_amdgcn_readfirstlane(s_cd[w][c][8]);
    const int total = P[CPB];
    for (int c0 = 0; c0 < total; c0 += 64) {
      const int idx = c0 + lane;
      bool go = false;
      int key = 0, ql = 0, qx = 0;
      if (idx < total) {
        int cb = 0;
#pragma unroll
        for (int c = 1; c < CPB; ++c) cb += (idx >= P[c]) ? 1 : 0;
        int pb = P[0];
#pragma unroll
        for (int c = 1; c < CPB; ++c) pb = (cb >= c) ? P[c] : pb;
        const int *d = &s_cd[w][cb][0];
        const int g = d[0], sk = d[1], gb = d[2], tjc = d[3], k = d[4], j = d[5], dv = d[6], k1 = sk - gb, c = idx - pb;
        const int cq = static_cast<int>((static_cast<float>(c) + 0.5f) * __int_as_float(d[7])), cr = c - cq * dv;
#ifdef RXMD_EXPERIMENTS
        const bool ifast = ONCE && bx.probe != 8;
#else
        constexpr bool ifast = ONCE;
#endif
        const int ci = ifast ? cr : cq, cl = cb * KW + (ifast ? cq : cr);
        const int i1 = s_cap[w][gb + ci], l1 = s_capl[w][cl];
        const double BOjk_f = s_bo[w][sk], BOij_f = s_bo[w][gb + i1], BOkl_f = s_bokl[w][cl];
        const int i = s_nb[w][gb + i1], l = s_ll[w][cl];
        go = (i1 != k1) && (BOij_f * BOjk_f > cutof2_esub) && (i != k) &&
             (BOjk_f * BOkl_f > cutof2_esub) && (BOij_f * (BOjk_f * BOjk_f) * BOkl_f > MINBO0) && (l != i) && (l != j);
        if (go) {
          const int i4 = (((s_meta[w][gb + i1] & 255) * ff.n1 + tjc) * ff.n1 + (s_meta[w][sk] & 255)) * ff.n1 + s_tl[w][cl];
          qx = tor_lds ? static_cast<int>(s_tor[i4]) : ff.inxn4[i4];
          go = qx != 0;
        }
        key = (g << 15) | (k1 << 10) | (i1 << 5) | l1;
        ql = l;
      }
      const unsigned long long m = __ballot(go);
      if (go) { const int qp = (qh + qn + __popcll(m & ((1ULL << lane) - 1ULL))) & 127; s_q[w][qp] = key; s_ql[w][qp] = ql; s_qx[w][qp] = static_cast<unsigned short>(qx); }
      qn += __popcll(m);
      wave_lds_sync();
      if (ONCE) {
        // one visit: the queue is a ring, and a batch ends where a (centre bond, l) run ends -- the run's sum over i is formed inside the batch and
        // leaves the wavefront ONCE (a run has <= 31 entries: among 64 there is always a boundary; the run at the tail may still grow)
        while (qn >= 64) {
          const int a = s_q[w][(qh + lane) & 127] & E4B_SEG, b = s_q[w][(qh + lane + 1) & 127] & E4B_SEG;
          const unsigned long long bm = __ballot(lane + 1 < qn && a != b);
#ifdef RXMD_EXPERIMENTS
          const int n = (bm && bx.probe != 8) ? 64 - __clzll(bm) : 64;
#else
          const int n = bm ? 64 - __clzll(bm) : 64;
#endif
          evaluate(n);
          qh = (qh + n) & 127; qn -= n;
        }
      } else if (qn >= 64) {
        evaluate(64);
        const int rest = qn - 64;
        const int v = (lane < rest) ? s_q[w][64 + lane] : 0, vl = (lane < rest) ? s_ql[w][64 + lane] : 0, vx = (lane < rest) ? s_qx[w][64 + lane] : 0;
        wave_lds_sync();
        if (lane < rest) { s_q[w][lane] = v; s_ql[w][lane] = vl; s_qx[w][lane] = static_cast<unsigned short>(vx); }
        wave_lds_sync();
        qn = rest;
      }
    }
    wave_lds_sync();                                // the next round overwrites the k-side lists
  }
  if (qn > 0) evaluate(qn);

  wave_lds_sync();
  const double a_cf = s_acc[w][lane][0], a_fx = s_acc[w][lane][1], a_fy = s_acc[w][lane][2], a_fz = s_acc[w][lane][3];
  const double a_cjk1 = s_acc[w][lane][4], a_cjk2 = s_acc[w][lane][5], a_cdk = s_acc[w][lane][6];
  if (sl_me < nj_me) {
    const int o = ob_me + sl_me;
    if (a_cf != 0.0 || a_cjk1 != 0.0) cf1[o] += a_cf + a_cjk1;
    if (a_cjk2 != 0.0) cf2[o] += a_cjk2;
    if (a_cdk != 0.0) cdn[o] += a_cdk;
    if (a_fx != 0.0 || a_fy != 0.0 || a_fz != 0.0) { fnx[o] += a_fx; fny[o] += a_fy; fnz[o] += a_fz; }
  }
  if (self_me) {
    double sx = a_fx, sy = a_fy, sz = a_fz, scd = a_cdk;
    if (ONCE) {                                     // (this lane's accumulators hold minus the forces on the l atoms)
      const int gb = gbase(g_me);
      for (int t = 0; t < nj_me; ++t) { sx -= s_acc[w][gb + t][1]; sy -= s_acc[w][gb + t][2]; sz -= s_acc[w][gb + t][3]; scd += s_acc[w][gb + t][6]; }
    }
    cds[j_me] += scd;
    fx[j_me] += sx; fy[j_me] += sy; fz[j_me] += sz;
  }
  wave_lds_sync();                                // the next pass (packed form, two passes) rebuilds the tables
  }   // pass
  }   // groups
  e8 = wave_sum_b(e8); e9 = wave_sum_b(e9);
  if (lane == 0) {
    if (e8 != 0.0) atomicAdd(pe + 8, e8);
    if (e9 != 0.0) atomicAdd(pe + 9, e9);
  }
}

// The k-l side of the one-visit torsions (k_e4b<*, true>): row = bond (k, l1) of the compact tables, column = slot of j in the list of k, TW columns
// per row (TW = the longest bond list of the step, a multiple of 4).  A thread per bond adds the flagged columns of its row in slot order to the
// bond's own accumulators -- cf1 (ForceB on k-l) and the force on the neighbour l -- and clears the flags it used: the flag array is all zero
// again behind this kernel.
__global__ void __launch_bounds__(256) k_e4b_deliver(int nbonds, int TW, const double4 *__restrict__ tkl, unsigned *__restrict__ tflag4,
                                                      double *__restrict__ cf1, double *__restrict__ fnx, double *__restrict__ fny, double *__restrict__ fnz) {
  const int o = xcd_swizzle(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  if (o >= nbonds) return;
  const int nw = TW >> 2;
  unsigned *fw = tflag4 + static_cast<size_t>(o) * nw;
  const double4 *row = tkl + static_cast<size_t>(o) * TW;
  double c = 0.0, f0 = 0.0, f1 = 0.0, f2 = 0.0;
  bool any = false;
  for (int wd = 0; wd < nw; ++wd) {
    unsigned m = fw[wd];
    if (m == 0u) continue;
    fw[wd] = 0u;
    any = true;
    for (int b = 0; b < 4; ++b, m >>= 8)
      if (m & 255u) { const double4 t = row[4 * wd + b]; c += t.x; f0 += t.y; f1 += t.z; f2 += t.w; }
  }
  if (any) { cf1[o] += c; fnx[o] += f0; fny[o] += f1; fnz[o] += f2; }
}

// ------------------------------------------------------------------------------------------------
// Hydrogen bonds, round 5: two kernels instead of one search-and-sweep.  The counters of the one-kernel form (profiles/r05_a_sq_*): 61,240
// wavefronts that each live ~60 us, 87 % of it parked on s_waitcnt behind ~66 dependent loads (the donor search walks type[nbr[o]] and
// bo0[o] slot by slot; then position, row, hydrogen, candidates one round trip after the other), the vector unit busy 6 % of the time, and
// 22 M atomic requests at the memory side for the 51 M acceptor-force additions.
//   k_ehb_donors  one thread per resident: the slots of its bond list that hold a hydrogen (type 2, pot.F90:595) with BO > MINBO0, from the
//                 CSR tables (the partner's type rides with the bond: no gather); atoms with one are appended to a donor list (ballot + one
//                 atomic per wavefront; the order of the list is arbitrary -- every sum behind it is an atomic or owned by one wavefront).
//   k_ehb_sweep   a persistent grid; a wavefront takes whole donors: the row's acceptor candidates are compacted into LDS (<= EHB_CAP at a time),
//                 their positions gathered ONCE into LDS, then every hydrogen slot of the donor walks the staged candidates four dense batches
//                 at a time out of LDS -- no load inside the evaluation -- and the acceptor's force is summed over the hydrogen slots in
//                 registers before its three atomics.
// Measured (979,776 atoms, profiles/r05_*): without the acceptor atomics the sweep takes 0.40 ms (the one-kernel form: 0.70), with them 1.07 --
// 51 M FP64 adds at the memory side are 0.67 ms whatever the wavefronts do around them (candidates sorted by atom index so that the acceptors of
// one molecule sit in neighbouring lanes: 1.00; RDX donors have ONE hydrogen, so summing over slots saves nothing there).  The atomics are the floor.
constexpr int EHB_CAP = 256;                              // candidates staged per flush (a 447-entry RDX row has <= 255 N / O partners)
constexpr int EHB_REGIONS = 64;                           // sub-lists of the donor list, each with its own counter
__global__ void __launch_bounds__(256) k_ehb_donors(int N, unsigned donor_types, const int *__restrict__ boff, const unsigned char *__restrict__ btype, const double *__restrict__ bo0,
                                                     const int *__restrict__ type, int2 *__restrict__ don, int *__restrict__ cnt, int region_cap) {
  // the list is EHB_REGIONS lists (workgroup b appends to region b mod EHB_REGIONS, region_cap entries each): 15,000 returning atomics on ONE
  // counter were 0.15 ms of this 0.18 ms kernel
  const int i = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63, reg = blockIdx.x & (EHB_REGIONS - 1);
  unsigned hm = 0u;
  if (i < N && ((donor_types >> (type[i] & 31)) & 1u)) {
    const int ob = boff[i], n = min(boff[i + 1] - ob, 32);
    unsigned hs = 0u;                                                   // the slots that hold a hydrogen: one pass over the type bytes (independent loads) ...
    for (int s = 0; s < n; ++s) hs |= (btype[ob + s] == 2 ? 1u : 0u) << s;
    for (unsigned m_ = hs; m_; m_ &= m_ - 1u) {                         // ... then the bond order of those only (most N / O atoms have none or one)
      const int s = __ffs(m_) - 1;
      if (bo0[ob + s] > MINBO0) hm |= 1u << s;                          // pot.F90:595
    }
  }
  const unsigned long long m = __ballot(hm != 0u);
  if (m == 0ULL) return;
  const int first = __ffsll(static_cast<long long>(m)) - 1;
  int base = 0;
  if (lane == first) base = atomicAdd(cnt + reg, __popcll(m));
  base = __shfl(base, first, 64);
  if (hm != 0u) don[static_cast<size_t>(reg) * region_cap + base + __popcll(m & ((1ULL << lane) - 1ULL))] = make_int2(i, static_cast<int>(hm));
}

__global__ void __launch_bounds__(256) k_ehb_sweep(int S10, DevFF ff, const int2 *__restrict__ don, const int *__restrict__ ndon, int region_cap, const int *__restrict__ boff, const int *__restrict__ nbr,
                                                    const int *__restrict__ type, const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
                                                    const double *__restrict__ bo0, const int *__restrict__ nb10, const int *__restrict__ n10, const double4 *__restrict__ pk, const int *__restrict__ perm,
                                                    double *__restrict__ cf1, double *__restrict__ fnx, double *__restrict__ fny, double *__restrict__ fnz,
                                                    double *__restrict__ fx, double *__restrict__ fy, double *__restrict__ fz, double *__restrict__ pe,
                                                    double *__restrict__ fsx, double *__restrict__ fsy, double *__restrict__ fsz
#ifdef RXMD_EHB_DEBUG
                                                    , int *dbg, int lim_n, int lim_b, int lim_nb
#endif
                                                    ) {
#ifdef RXMD_EHB_DEBUG
#define EHB_CHECK(code, v, lim, aux) if ((v) < 0 || (v) >= (lim)) { if (atomicCAS(dbg, 0, (code)) == 0) { dbg[1] = (v); dbg[2] = (aux); dbg[3] = (lim); } return; }
#define EHB_CHECKC(code, v, lim, aux) if ((v) < 0 || (v) >= (lim)) { if (atomicCAS(dbg, 0, (code)) == 0) { dbg[1] = (v); dbg[2] = (aux); dbg[3] = (lim); } continue; }
#else
#define EHB_CHECK(code, v, lim, aux)
#define EHB_CHECKC(code, v, lim, aux)
#endif
  __shared__ unsigned s_cand[4][EHB_CAP];                 // compacted list entries of the row (type bits select the parameter row)
  __shared__ double s_px[4][EHB_CAP], s_py[4][EHB_CAP], s_pz[4][EHB_CAP];
  __shared__ int s_k[4][EHB_CAP];
  __shared__ double s_hp[4][16][4];                       // (ti, H, type k) parameter rows of the current donor, by acceptor type
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  // the donor list is EHB_REGIONS sub-lists: lane r holds the inclusive prefix of their lengths; donor d lives in the first region whose prefix exceeds d
  int pre = ndon[lane & (EHB_REGIONS - 1)];
#pragma unroll
  for (int o = 1; o < EHB_REGIONS; o <<= 1) { const int t = __shfl_up(pre, o, 64); if (lane >= o) pre += t; }
  const int nw = gridDim.x * 4, nd = __shfl(pre, EHB_REGIONS - 1, 64);
  double e10 = 0.0;
  for (int d = blockIdx.x * 4 + w; d < nd; d += nw) {
    const int reg = __popcll(__ballot(pre <= d));                       // (EHB_REGIONS == 64: one lane per region)
    const int before = reg > 0 ? __shfl(pre, reg - 1, 64) : 0;
    const int2 rec = don[static_cast<size_t>(reg) * region_cap + (d - before)];
    const int i = __builtin_amdgcn_readfirstlane(rec.x);
    const unsigned hmask = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(rec.y));
    EHB_CHECK(1, i, lim_n, d)
    const int ti = type[i], ob = boff[i], n = n10[i] & N10_COUNT;
    EHB_CHECK(2, ob, lim_b - 32, i)
    EHB_CHECK(3, n, S10 + 1, i)
    const double xi = x[i], yi = y[i], zi = z[i];
    const size_t row = static_cast<size_t>(i) * S10;
    const int inx_l = (lane >= 1 && lane <= ff.nso && lane < 16) ? ff.inxn3hb[(ti * ff.n1 + 2) * ff.n1 + lane] : 0;
    if (lane < 16) { const DevHbP hp = ff.hb[inx_l]; s_hp[w][lane][0] = hp.r0hb; s_hp[w][lane][1] = hp.phb1; s_hp[w][lane][2] = hp.phb2; s_hp[w][lane][3] = hp.phb3; }   // (row 0 is never used: the compaction drops its candidates)
    int jl = 0; double bl = 0.0, xjl = 0.0, yjl = 0.0, zjl = 0.0;        // lane s holds atom, bond order and position of hydrogen slot s
    if (lane < 32 && ((hmask >> lane) & 1u)) { jl = nbr[ob + lane];
#ifdef RXMD_EHB_DEBUG
      if (jl < 0 || jl >= lim_nb) { if (atomicCAS(dbg, 0, 4) == 0) { dbg[1] = jl; dbg[2] = i; dbg[3] = lane; } jl = 0; }
#endif
      bl = bo0[ob + lane]; xjl = x[jl]; yjl = y[jl]; zjl = z[jl]; }
    V3 fi_t = {0, 0, 0};
    int qn = 0;

    auto flush = [&]() {                                    // the staged candidates against every hydrogen slot of the donor
      wave_lds_sync();
      for (int q = lane; q < qn; q += 64) {
        EHB_CHECKC(5, static_cast<int>(s_cand[w][q] & NB10_IDX_MASK), lim_nb, i)
        const int ks = static_cast<int>(s_cand[w][q] & NB10_IDX_MASK);       // list entries are cell-sorted positions; the atom comes from perm
        // x, y, z ONLY: the w component of the packed copy is rewritten with the charge by k_sorted_charge on the MAIN stream while this kernel runs
        // on the bonded chain's stream (assemble.hip: bonded_chain_begin) -- a 32-byte load of the element would be an unordered read of that word
        const double2 pxy = *reinterpret_cast<const double2 *>(pk + ks);
        s_px[w][q] = pxy.x; s_py[w][q] = pxy.y; s_pz[w][q] = reinterpret_cast<const double *>(pk + ks)[2]; s_k[w][q] = perm[ks];
      }
      wave_lds_sync();
      double akx[EHB_CAP / 64], aky[EHB_CAP / 64], akz[EHB_CAP / 64];       // acceptor force of this lane's candidate of each batch, summed over the hydrogen slots
#pragma unroll
      for (int b = 0; b < EHB_CAP / 64; ++b) { akx[b] = 0.0; aky[b] = 0.0; akz[b] = 0.0; }
      for (unsigned hm = hmask; hm; hm &= hm - 1u) {
        const int s = __ffs(hm) - 1;
        const int j = __shfl(jl, s, 64);
        const double BOij = __shfl(bl, s, 64);
        const double xj = __shfl(xjl, s, 64), yj = __shfl(yjl, s, 64), zj = __shfl(zjl, s, 64);
        const V3 rij = {xi - xj, yi - yj, zi - zj};
        const double nij = sqrt(dot(rij, rij));
        double cfs = 0.0, nterm = 0.0;
        V3 fi_s = {0, 0, 0}, fj_s = {0, 0, 0};
#pragma unroll
        for (int b = 0; b < EHB_CAP / 64; ++b) {
          const int qq = 64 * b + lane;
          if (qq >= qn) continue;
          const int k = s_k[w][qq];
          if (k == j || k == i) continue;
          const double *hp = s_hp[w][(s_cand[w][qq] >> NB10_IDX_BITS) & 15u];
          const double r0hb = hp[0], phb1 = hp[1], phb2 = hp[2], phb3 = hp[3];
          const double pxk = s_px[w][qq], pyk = s_py[w][qq], pzk = s_pz[w][qq];
          const V3 rik = {xi - pxk, yi - pyk, zi - pzk};
          if (!(dot(rik, rik) < 100.0)) continue;                          // rchb2, pot.F90:610
          const V3 rjk = {xj - pxk, yj - pyk, zj - pzk};
          const double njk = sqrt(dot(rjk, rjk));
          double cos_ijk = -dot(rij, rjk) / (nij * njk);
          if (cos_ijk > MAXANGLE) cos_ijk = MAXANGLE;
          if (cos_ijk < MINANGLE) cos_ijk = MINANGLE;
          const double sh2 = 0.5 * (1.0 - cos_ijk);                        // sin^2(theta/2)
          const double sin_xhz4 = sh2 * sh2, cos_xhz1 = 1.0 - cos_ijk;
          const double exp_hb2 = exp(-phb2 * BOij);
          const double exp_hb3 = exp(-phb3 * (r0hb / njk + njk / r0hb - 2.0));
          const double PEhb = phb1 * (1.0 - exp_hb2) * exp_hb3 * sin_xhz4;
          e10 += PEhb; nterm += 1.0;
          cfs += phb1 * phb2 * exp_hb2 * exp_hb3 * sin_xhz4;               // CEhb(1) -> ForceB(i,j)
          const double CEhb2 = -0.5 * phb1 * (1.0 - exp_hb2) * exp_hb3 * cos_xhz1;
          const double CEhb3 = -PEhb * phb3 * (-r0hb / (njk * njk) + 1.0 / r0hb) * (1.0 / njk);
          V3 fi, fk;
          angle_forces(CEhb2, rij, nij, rjk, njk, fi, fk);
          const V3 ff3 = {CEhb3 * rjk.x, CEhb3 * rjk.y, CEhb3 * rjk.z};   // f(j) -= ff ; f(k) += ff
          fi_s.x += fi.x; fi_s.y += fi.y; fi_s.z += fi.z;
          fj_s.x += -(fi.x + fk.x) - ff3.x; fj_s.y += -(fi.y + fk.y) - ff3.y; fj_s.z += -(fi.z + fk.z) - ff3.z;
          akx[b] += fk.x + ff3.x; aky[b] += fk.y + ff3.y; akz[b] += fk.z + ff3.z;
        }
        cfs = wave_sum_b(cfs); nterm = wave_sum_b(nterm);
        fj_s.x = wave_sum_b(fj_s.x); fj_s.y = wave_sum_b(fj_s.y); fj_s.z = wave_sum_b(fj_s.z);
        fi_t.x += fi_s.x; fi_t.y += fi_s.y; fi_t.z += fi_s.z;
        if (lane == 0 && nterm > 0.0) {                     // the donor's bonds belong to this wavefront alone
          const int o = ob + s;
          cf1[o] += cfs;
          fnx[o] += fj_s.x; fny[o] += fj_s.y; fnz[o] += fj_s.z;
        }
      }
#pragma unroll
      for (int b = 0; b < EHB_CAP / 64; ++b) {
        const int qq = 64 * b + lane;
        if (qq < qn && (akx[b] != 0.0 || aky[b] != 0.0 || akz[b] != 0.0)) {
          // the acceptor's force goes to its CELL-SORTED position: the candidates of a row are neighbours in that order, so the lanes of one atomic
          // instruction hit neighbouring addresses and share memory-side requests (by atom index they are scattered: 2.2 lanes per request)
          const int ks = static_cast<int>(s_cand[w][qq] & NB10_IDX_MASK);
          EHB_CHECKC(6, ks, lim_nb, qq)
          atomicAdd(fsx + ks, akx[b]); atomicAdd(fsy + ks, aky[b]); atomicAdd(fsz + ks, akz[b]);
        }
      }
      wave_lds_sync();
      qn = 0;
    };

    for (int c0 = 0; c0 < n; c0 += 256) {                   // four batches of entry words requested together
      unsigned ent[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int kk = c0 + 64 * u + lane; ent[u] = kk < n ? static_cast<unsigned>(nb10[row + kk]) : 0xffffffffu; }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (c0 + 64 * u >= n) break;                                        // wave-uniform
        const int inx = __shfl(inx_l, static_cast<int>((ent[u] >> NB10_IDX_BITS) & 15u), 64);   // (ti, H, type k) row, held by lane = type k
        const bool keep = ent[u] != 0xffffffffu && inx != 0;
        const unsigned long long m = __ballot(keep);
        if (keep) s_cand[w][qn + __popcll(m & ((1ULL << lane) - 1ULL))] = ent[u];
        qn += __popcll(m);
        if (qn > EHB_CAP - 64) flush();
      }
    }
    if (qn > 0) flush();
    fi_t.x = wave_sum_b(fi_t.x); fi_t.y = wave_sum_b(fi_t.y); fi_t.z = wave_sum_b(fi_t.z);
    if (lane == 0 && (fi_t.x != 0.0 || fi_t.y != 0.0 || fi_t.z != 0.0)) { atomicAdd(fx + i, fi_t.x); atomicAdd(fy + i, fi_t.y); atomicAdd(fz + i, fi_t.z); }
    wave_lds_sync();                                        // the next donor rewrites the parameter rows
  }
  block_energy_add(e10, pe + 10);
}

__global__ void k_zero3(int n, double *__restrict__ a, double *__restrict__ b, double *__restrict__ c) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) { a[k] = 0.0; b[k] = 0.0; c[k] = 0.0; }
}
// frc[atom at sorted position k] += the acceptor forces the hydrogen-bond sweep left at position k (perm is a permutation: no two threads meet)
__global__ void k_add_sorted3(int G, const int *__restrict__ perm, const double *__restrict__ sx, const double *__restrict__ sy, const double *__restrict__ sz,
                              double *__restrict__ fx, double *__restrict__ fy, double *__restrict__ fz) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= G) return;
  const double a = sx[k], b = sy[k], c = sz[k];
  if (a != 0.0 || b != 0.0 || c != 0.0) { const int i = perm[k]; fx[i] += a; fy[i] += b; fz[i] += c; }
}

void Engine::bonded_energies() {
  double *pe_d = scal + 32;   // 14 energy accumulators live behind the CG scalars
  k_ebond_terms<<<nblk(nbonds_res, 256), 256, 0, stream>>>(nbonds_res, dff, bown, nbr, btype, type, gid, bo0, bo1, bo2, bo3, delta, deltalp, cf1, cf2, cf3, ecoa, bt1, bt2, pe_d);
  k_elnpr_atoms<<<nblk(N, 256), 256, 0, stream>>>(N, dff, boff, type, bt1, bt2, delta, deltalp, dDlp, ecoef, pe_d);
  k_elnpr_bonds<<<nblk(nbonds_res, 256), 256, 0, stream>>>(nbonds_res, dff, bown, nbr, btype, type, bo2, bo3, delta, deltalp, dDlp, ecoef, cf1, cf2, cf3, cdn);
  const bool kt3 = kt_begin(&st.ms_k_e3b);
  // the work-queue form (k_e3q) when the angle tables fit LDS: sixteen centre atoms per wavefront where no bond list of the step is longer than 12
  // (h_err[2]: the longest list, 0 up to 8), eight up to 24, four beyond
  if (opt.e3b_queue != 0 && dff.n1 <= 8 && dff.nvaty <= 255) {
    const int maxl = h_err[2];
    auto go = [&](auto kern, int na, int minw) {
      const int NG = nblk(N, 4 * na);
      kern<<<std::min(NG, (num_cu * minw + 7) & ~7), 256, 0, stream>>>(N, NG, dff, boff, nbr, btype, type, pos[0], pos[1], pos[2], bo0, bo2, bo3, delta, nlp, dDlp, epen, ecoa, cf1, cf2, cf3, cdn, fnx, fny, fnz,
                                                                     cds, frc[0], frc[1], frc[2], pe_d, static_cast<int>(opt.e4b_probe));
    };
    const bool w3 = opt.e3b_queue == 3;
    if (maxl <= 12) { if (w3) go(k_e3q<3, 16>, 16, 3); else go(k_e3q<2, 16>, 16, 2); }
    else if (maxl <= 24) { if (w3) go(k_e3q<3, 8>, 8, 3); else go(k_e3q<2, 8>, 8, 2); }
    else { if (w3) go(k_e3q<3, 4>, 4, 3); else go(k_e3q<2, 4>, 4, 2); }
  } else
    k_e3b<<<nblk(N, 256), 256, 4 * 5 * E3B_CAP * sizeof(double), stream>>>(N, dff, boff, nbr, btype, type, pos[0], pos[1], pos[2], bo0, bo2, bo3, delta, nlp, dDlp, epen, ecoa, cf1, cf2, cf3, cdn, fnx, fny, fnz,
                                          cds, frc[0], frc[1], frc[2], pe_d);
  RX_HIP(hipGetLastError());                       // 70 KB of dynamic LDS: a refused launch must not pass for "no valence angles"
  kt_end(kt3);
  BoxImg bx;
  for (int a = 0; a < 3; ++a) { for (int c = 0; c < 3; ++c) { bx.H[3 * a + c] = box.H[a][c]; bx.Hi[3 * a + c] = box.Hi[a][c]; } bx.L[a] = box.lat[a]; }
  bx.ortho = grid.ortho; bx.probe = 0;
  bx.probe = static_cast<int>(opt.e4b_probe);       // (experiments build only: 0 otherwise)
  // four atoms per wavefront when every bond list of this step fits 15 slots (h_err[2] = the largest list if longer than 8, read with the error word
  // after the list build); RXMD_E4B_SLOTS=32 forces the general kernel (tests)
  // instances (RXMD_E4B_SLOTS forces one, tests): packed eight atoms (default when no list of the step is longer than 15), two atoms
  // x 32 slots (32; default otherwise), packed four atoms (4: any list; on the SiC workload, 16+ bonds per atom, it needs two passes
  // for most groups and is slower than the default there, 3.69 against 3.47 ms), four atoms x 16 slots (16, needs lists <= 15)
  const int want = static_cast<int>(opt.e4b_slots);
  const bool narrow = h_err[2] <= 15 && want != 32 && want != 4;
  const bool kt4 = kt_begin(&st.ms_k_e4b);
  bool once = opt.e4b_once != 0;
  const int TW = (std::max(h_err[2], 8) + 3) & ~3;                          // columns of the k-l delivery table: the longest bond list of the step (h_err[2]: 0 when none is longer than 8)
  if (once) {
    const size_t need = static_cast<size_t>(std::max(nbonds, 1)) * TW;
    if (need > e4b_cap) {                                                   // 33 bytes per (bond, column), touched sparsely: RDX 18^3 2.9 GB; a table that would take more than
      size_t fre = 0, tot = 0;                                              // half of what is free is not allocated -- the step falls back to the two-visit form
      RX_HIP(hipMemGetInfo(&fre, &tot));
      const size_t want_entries = need + need / 4;
      if (want_entries * 33 > (fre + e4b_cap * 33) / 2) once = false; else alloc_e4b_delivery(want_entries);
    }
  }
  if (once) {
    if (e4b_dirty) RX_HIP(hipMemsetAsync(e4b_flag, 0, e4b_cap, stream));   // (an error between the torsion kernel and the delivery of an earlier call left flags behind)
    e4b_dirty = true;
  }
  // A persistent grid (default): three workgroups per CU, each walking every gridDim.x-th group of centre atoms.  One workgroup per group -- 30,618 on
  // RDX 18^3, each needing four 168-register wavefront slots and 49 KB of LDS at once -- kept only 2.0-2.4 of the 3 wavefronts per SIMD resident (SQ
  // counters, profiles/r06_e4b_*): 3.16 -> 2.54 ms for the two-visit form, 3.54 -> 1.98 ms for the one-visit form.  RXMD_E4B_WPB=1 (workgroups of one
  // wavefront, tables released per wavefront) gains 0.05 ms without the persistent grid and loses with it (the dispatcher does not spread them evenly
  // over the SIMDs).
  const int n4 = dff.n1 * dff.n1 * dff.n1 * dff.n1;
  const size_t tor_bytes = (dff.n1 <= 8 && dff.ntoty <= 255) ? static_cast<size_t>((n4 + 15) & ~15) : 0;
  const bool wpb1 = opt.e4b_wpb == 1;
  auto launch = [&](auto inst, int per_wave) {
    constexpr int L = decltype(inst)::value;
    auto go = [&](auto kern, int wpb) {
      const int VB = nblk(N, per_wave * wpb);                              // groups of atoms = workgroups of the full grid
      const int grid = opt.e4b_persist != 0 ? std::min(VB, (num_cu * (12 / wpb) + 7) & ~7) : VB;   // persistent: 3 wavefronts per SIMD resident (LDS and registers allow no more)
      kern<<<grid, 64 * wpb, tor_bytes, stream>>>(N, dff, boff, nbr, btype, nbrcnt, type, gid, pos[0], pos[1], pos[2], bo0, bo2, delta, etor, econ, cf1, cf2, cdn, fnx, fny, fnz,
                                                  cds, frc[0], frc[1], frc[2], pe_d, bx, brev, e4b_t, e4b_flag, TW, VB);
    };
    if (once) { if (wpb1) go(k_e4b<L, true, 1>, 1); else go(k_e4b<L, true, 4>, 4); }
    else      { if (wpb1) go(k_e4b<L, false, 1>, 1); else go(k_e4b<L, false, 4>, 4); }
    RX_HIP(hipGetLastError());
  };
  if (want == 4) launch(std::integral_constant<int, 1>{}, 4);
  else if (narrow && want != 16) launch(std::integral_constant<int, 0>{}, 8);
  else if (narrow) launch(std::integral_constant<int, 4>{}, 4);
  else launch(std::integral_constant<int, 5>{}, 2);
  if (once) {
    k_e4b_deliver<<<nblk(nbonds, 256), 256, 0, stream>>>(nbonds, TW, e4b_t, reinterpret_cast<unsigned *>(e4b_flag), cf1, fnx, fny, fnz);
    e4b_dirty = false;
  }
  kt_end(kt4);
  const bool kth = kt_begin(&st.ms_k_ehb);
  if (ehb_donor_types != 0u) {                    // (a force field without a hydrogen-bond row for hydrogen = type 2 has no donors: water, pot.F90:595)
    const int region_cap = (nblk(N, 256) + EHB_REGIONS - 1) / EHB_REGIONS * 256;            // every atom of the workgroups of a region a donor: cannot overflow
    if (static_cast<size_t>(region_cap) * EHB_REGIONS > ehb_don_cap) throw EngineError(RXMD_E_STATE, "hydrogen-bond donor list: capacity");
    RX_HIP(hipMemsetAsync(ehb_cnt, 0, sizeof(int) * EHB_REGIONS, stream));
    k_ehb_donors<<<nblk(N, 256), 256, 0, stream>>>(N, ehb_donor_types, boff, btype, bo0, type, ehb_don, ehb_cnt, region_cap);
    k_zero3<<<nblk(G, 256), 256, 0, stream>>>(G, fsort[0], fsort[1], fsort[2]);
    if (ehb_blocks_per_cu == 0) {                 // the persistent grid fills the device exactly: workgroups per CU from the kernel's own register / LDS footprint
      int nbk = 0;
      RX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nbk, k_ehb_sweep, 256, 0));
      ehb_blocks_per_cu = std::max(nbk, 1);
    }
#ifdef RXMD_EHB_DEBUG
    RX_HIP(hipMemsetAsync(ehb_cnt + EHB_REGIONS, 0, 4 * sizeof(int), stream));
    k_ehb_sweep<<<num_cu * ehb_blocks_per_cu, 256, 0, stream>>>(S10, dff, ehb_don, ehb_cnt, region_cap, boff, nbr, type, pos[0], pos[1], pos[2], bo0, nb10, n10, sorted_xyzi, perm, cf1, fnx, fny, fnz, frc[0], frc[1], frc[2], pe_d, fsort[0], fsort[1], fsort[2], ehb_cnt + EHB_REGIONS, N, static_cast<int>(bcap), NB);
    { int hd[EHB_REGIONS + 4]; RX_HIP(hipMemcpyAsync(hd, ehb_cnt, sizeof(hd), hipMemcpyDeviceToHost, stream)); RX_HIP(hipStreamSynchronize(stream));
      for (int r = 1; r < EHB_REGIONS; ++r) hd[0] += hd[r];
      hd[4] = hd[EHB_REGIONS]; hd[5] = hd[EHB_REGIONS + 1]; hd[6] = hd[EHB_REGIONS + 2]; hd[7] = hd[EHB_REGIONS + 3];
      std::fprintf(stderr, "[ehb debug] donors %d  blocks/CU %d  first violation: code %d value %d aux %d lim %d   (N %d bcap %zu NB %d S10 %d rows10 %d)\n", hd[0], ehb_blocks_per_cu, hd[4], hd[5], hd[6], hd[7], N, bcap, NB, S10, rows10); }
#else
    k_ehb_sweep<<<num_cu * ehb_blocks_per_cu, 256, 0, stream>>>(S10, dff, ehb_don, ehb_cnt, region_cap, boff, nbr, type, pos[0], pos[1], pos[2], bo0, nb10, n10, sorted_xyzi, perm, cf1, fnx, fny, fnz, frc[0], frc[1], frc[2], pe_d, fsort[0], fsort[1], fsort[2]);
#endif
    k_add_sorted3<<<nblk(G, 256), 256, 0, stream>>>(G, perm, fsort[0], fsort[1], fsort[2], frc[0], frc[1], frc[2]);
  }
  kt_end(kth);
}

}  // namespace rxmd
