// assemble.hip -- turns the per-slot accumulators of bonded.hip into forces, and the MD step.
//   ForceBondedTerms + ForceD   reference src/pot.F90:113-144, 1230-1273  -> k_cd_gather, k_ccbnd_terms / _sum, k_bond_force_terms / _sum
//   COPYATOMS(MODE_CPBK)        reference src/comm.F90:385-396, 474-482   -> Engine::fold_ghost_forces (engine.hip)
//   FORCE driver                reference src/pot.F90:2-90                -> Engine::force
//   MD loop body                reference src/main.F90:64-98              -> Engine::step
//
// The reference's ForceBondedTerms is a SERIAL loop over the local index i that (1) applies ForceD(i)
// -- which also adds Cbond(3) to ccbnd of every neighbour -- and (2) consumes and zeroes ccbnd(i) at
// once.  An increment that ForceD(k) makes to ccbnd(i) is therefore used only when k < i.  The same
// result as a gather:
//     cd(i)  = all cdbnd contributions to i
//     cc(i)  = ccbnd from the energy terms
//            + sum_s BO(i,s) A2(i,s) * [ cd(i) + (nbr(i,s) < i ? cd(nbr(i,s)) : 0) ]
//     f(i)   = self + sum_s fn(from neighbour) - sum_s { Cbond1(i,s) + (cd(i)+cd(j)) (A0+BO A1) dBOp
//                                                         + (cc(i)+cc(j)) dBOp } * (r_i - r_j)
// Atoms are stored in the reference's local order, so "k < i" is a plain index compare.
#include "engine.h"

#include <cmath>
#include <cstdlib>
#include <vector>

namespace rxmd {

static inline int nblk(long long n, int b) { return n > 0 ? static_cast<int>((n + b - 1) / b) : 1; }   // an empty rank still launches (kernels guard their range)

// Bond tables are compact (CSR, engine.h): bond o = boff[i] + s, its mirror image -- the same bond in the partner's list -- is brev[o].
__global__ void __launch_bounds__(256) k_cd_gather(int G, const int *__restrict__ boff, const int *__restrict__ brev,
                                                    const double *__restrict__ cds, const double *__restrict__ cdn, double *__restrict__ cd) {
  const int i = xcd_swizzle(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  if (i >= G) return;
  double s = cds[i];
  for (int o = boff[i], o1 = boff[i + 1]; o < o1; ++o) s += cdn[brev[o]];
  cd[i] = s;
}

// ccbnd and the bond forces in two steps each: ONE LANE PER BOND forms the bond's terms (every per-bond array read coalesced, the mirror bond's
// accumulators gathered through brev), then a thread per atom adds its bonds' terms in slot order -- the order of the thread-per-atom loops these
// replace, so cc(i) and the forces keep their bits.  (A thread per atom over the compact tables reads 13 arrays with a stride of ~5 entries
// between neighbouring lanes: 1.6 ms for the three kernels against 1.2 ms slot-major; in this form the strided reads are the two or three term
// arrays only.)
__global__ void __launch_bounds__(256) k_ccbnd_terms(int nbonds, const int *__restrict__ bown, const int *__restrict__ nbr, const int *__restrict__ brev,
                                                      const double *__restrict__ bo0, const double *__restrict__ bo2, const double *__restrict__ bo3,
                                                      const double *__restrict__ A2, const double *__restrict__ A3,
                                                      const double *__restrict__ cf1, const double *__restrict__ cf2, const double *__restrict__ cf3,
                                                      const double *__restrict__ cd, double *__restrict__ tu, double *__restrict__ tv) {
  const int o = xcd_swizzle(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  if (o >= nbonds) return;
  const int i = bown[o], j = nbr[o], oj = brev[o];
  const double c1 = cf1[o] + cf1[oj], c2 = cf2[o] + cf2[oj], c3 = cf3[o] + cf3[oj];
  const double B0 = bo0[o];
  // ForceBbo: Cbond(2) = cBO(1) A2 + (cBO(2)+cBO(3)) A3   (pot.F90:1354-1357)
  tu[o] = (c1 * B0) * A2[o] + (c2 * bo2[o] + c3 * bo3[o]) * A3[o];
  // ForceD(i): Cbond(2) ; ForceD(j) with j < i: Cbond(3)   (pot.F90:1262-1268, 125-138)
  tv[o] = B0 * A2[o] * (cd[i] + ((j < i) ? cd[j] : 0.0));
}
__global__ void __launch_bounds__(256) k_ccbnd_sum(int G, const int *__restrict__ boff, const double *__restrict__ tu, const double *__restrict__ tv, double *__restrict__ cc) {
  const int i = xcd_swizzle(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  if (i >= G) return;
  double s = 0.0;
  for (int o = boff[i], o1 = boff[i + 1]; o < o1; ++o) { s += tu[o]; s += tv[o]; }
  cc[i] = s;
}

__global__ void __launch_bounds__(256) k_bond_force_terms(int nbonds, const int *__restrict__ bown, const int *__restrict__ nbr, const int *__restrict__ brev,
                                                           const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
                                                           const double *__restrict__ bo0, const double *__restrict__ bo2, const double *__restrict__ bo3,
                                                           const double *__restrict__ dln2, const double *__restrict__ dln3, const double *__restrict__ dBOp,
                                                           const double *__restrict__ A0, const double *__restrict__ A1,
                                                           const double *__restrict__ cf1, const double *__restrict__ cf2, const double *__restrict__ cf3,
                                                           const double *__restrict__ fnx, const double *__restrict__ fny, const double *__restrict__ fnz,
                                                           const double *__restrict__ cd, const double *__restrict__ cc,
                                                           double *__restrict__ tx, double *__restrict__ ty, double *__restrict__ tz) {
  const int o = xcd_swizzle(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  if (o >= nbonds) return;
  const int i = bown[o], j = nbr[o], oj = brev[o];
  const double c1 = cf1[o] + cf1[oj], c2 = cf2[o] + cf2[oj], c3 = cf3[o] + cf3[oj];
  const double B0 = bo0[o], dB = dBOp[o], a1 = A1[o];
  const double a01 = A0[o] + B0 * a1;
  // ForceBbo Cbond(1) (pot.F90:1333-1335) + ForceD of both ends (pot.F90:1244-1246) + ccbnd of both ends (pot.F90:129-135)
  const double cb = c1 * a01 * dB + c2 * bo2[o] * (dln2[o] + a1 * dB) + c3 * bo3[o] * (dln3[o] + a1 * dB)
                  + (cd[i] + cd[j]) * a01 * dB + (cc[i] + cc[j]) * dB;
  tx[o] = fnx[oj] - cb * (x[i] - x[j]);
  ty[o] = fny[oj] - cb * (y[i] - y[j]);
  tz[o] = fnz[oj] - cb * (z[i] - z[j]);
}
__global__ void __launch_bounds__(256) k_bond_force_sum(int G, const int *__restrict__ boff, const double *__restrict__ tx, const double *__restrict__ ty, const double *__restrict__ tz,
                                                         double *__restrict__ fx, double *__restrict__ fy, double *__restrict__ fz) {
  const int i = xcd_swizzle(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  if (i >= G) return;
  double f0 = fx[i], f1 = fy[i], f2 = fz[i];
  for (int o = boff[i], o1 = boff[i + 1]; o < o1; ++o) { f0 += tx[o]; f1 += ty[o]; f2 += tz[o]; }
  fx[i] = f0; fy[i] = f1; fz[i] = f2;
}

void Engine::assemble_forces() {
  k_cd_gather<<<nblk(G, 256), 256, 0, stream>>>(G, boff, brev, cds, cdn, cd);
  k_ccbnd_terms<<<nblk(nbonds, 256), 256, 0, stream>>>(nbonds, bown, nbr, brev, bo0, bo2, bo3, A2, A3, cf1, cf2, cf3, cd, bt1, bt2);
  k_ccbnd_sum<<<nblk(G, 256), 256, 0, stream>>>(G, boff, bt1, bt2, cc_);
  k_bond_force_terms<<<nblk(nbonds, 256), 256, 0, stream>>>(nbonds, bown, nbr, brev, pos[0], pos[1], pos[2], bo0, bo2, bo3, dln2, dln3, dBOp, A0, A1, cf1, cf2, cf3,
                                                             fnx, fny, fnz, cd, cc_, bt1, bt2, bt3);
  k_bond_force_sum<<<nblk(G, 256), 256, 0, stream>>>(G, boff, bt1, bt2, bt3, frc[0], frc[1], frc[2]);
}

// ------------------------------------------------------------------------------------------------
// stress accumulators astr(1:6) = (xx, yy, zz, yz, zx, xy):  sum_i a_i b_i^T [* mass_i]
//   virial  : a = local position of residents AND ghosts, b = their force before the CPBK fold  (pot.F90:65-72)
//   kinetic : a = b = velocity of residents, weighted by the mass                                (main.F90:86-94)
// fixed grid + fixed-order final sum: the result does not depend on scheduling
__global__ void __launch_bounds__(256) k_stress_partial(int n, const double *__restrict__ ax, const double *__restrict__ ay, const double *__restrict__ az,
                                                         const double *__restrict__ bx, const double *__restrict__ by, const double *__restrict__ bz,
                                                         const int *__restrict__ type, DevFF ff, int use_mass, double *__restrict__ partials) {
  __shared__ double sm[256];
  double a[6] = {0, 0, 0, 0, 0, 0};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const double w = use_mass ? ff.atom[type[i]].mass : 1.0;
    const double p0 = ax[i], p1 = ay[i], p2 = az[i], f0 = bx[i] * w, f1 = by[i] * w, f2 = bz[i] * w;
    a[0] += p0 * f0; a[1] += p1 * f1; a[2] += p2 * f2; a[3] += p1 * f2; a[4] += p2 * f0; a[5] += p0 * f1;
  }
  for (int c = 0; c < 6; ++c) {
    sm[threadIdx.x] = a[c];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) partials[blockIdx.x * 6 + c] = sm[0];
    __syncthreads();
  }
}
// one wavefront per component: a lane adds every 64th partial sum, then the wavefront's fixed-order sum (240 dependent loads by one thread were 29 us)
__global__ void __launch_bounds__(384) k_stress_final(int nblocks, const double *__restrict__ partials, double *__restrict__ acc6) {
  const int c = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double s = 0.0;
  for (int b = lane; b < nblocks; b += 64) s += partials[b * 6 + c];
  s = wave_sum64(s);
  if (lane == 0) acc6[c] += s;
}

void Engine::accumulate_stress(bool kinetic) {
  const int n = kinetic ? N : G, nb = 240;
  if (kinetic) k_stress_partial<<<nb, 256, 0, stream>>>(n, vel[0], vel[1], vel[2], vel[0], vel[1], vel[2], type, dff, 1, partials);
  else k_stress_partial<<<nb, 256, 0, stream>>>(n, pos[0], pos[1], pos[2], frc[0], frc[1], frc[2], type, dff, 0, partials);
  k_stress_final<<<1, 384, 0, stream>>>(nb, partials, scal + 48);
}

// The charge-free part of FORCE on the third stream (engine.h: bond_stream): bond orders, every bonded term, the assembly of the bonded forces, next
// to the ghost-charge halo and ENbond (bound by L2 lines; the bonded kernels by FP64 chains).  Queued behind everything the main stream holds at this
// point; joined by Engine::force in front of the stress sums.  The force array belongs to this chain until then (k_delta_lp clears it, k_bond_force_sum finishes it); ENbond leaves its part in fnb.
__global__ void k_add_force3(int n, const double *__restrict__ ax, const double *__restrict__ ay, const double *__restrict__ az, double *__restrict__ fx, double *__restrict__ fy, double *__restrict__ fz) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { fx[i] += ax[i]; fy[i] += ay[i]; fz[i] += az[i]; }
}
// INVARIANT of the two chains: from here to the join the bonded chain (this function, on bond_stream) may READ positions, types, bond tables and the
// x, y, z of the cell-sorted packed copy, and WRITES frc, fsort, the bond accumulators and pe; the main stream meanwhile writes q of the ghosts, the
// w component of sorted_xyzi (k_sorted_charge) and fnb.  No kernel of the bonded chain may decode sorted_xyzi[].w or read q.
void Engine::bonded_chain_begin() {
  RX_HIP(hipMemsetAsync(scal + 32, 0, sizeof(double) * 16, stream));     // the energy accumulators: both chains add to them
  RX_HIP(hipEventRecord(ev_fork, stream));
  RX_HIP(hipStreamWaitEvent(bond_stream, ev_fork, 0));
  std::swap(stream, bond_stream);
  try {
    { const bool kt = kt_begin(&st.ms_k_bondorder, &st.ms_bo); bond_orders(); kt_end(kt); }
    const KtPair t_bonded = outer_begin(&st.ms_bonded);
    bonded_energies();
    { const bool kt = kt_begin(&st.ms_k_assemble); assemble_forces(); kt_end(kt); }
    outer_end(t_bonded);
  } catch (...) { std::swap(stream, bond_stream); throw; }
  std::swap(stream, bond_stream);
  RX_HIP(hipEventRecord(ev_bond, bond_stream));
}

void Engine::force(bool defer_host_read) {
  if (!atoms_set) throw EngineError(RXMD_E_STATE, "atoms were never set");
  const KtPair t_force = outer_begin(&st.ms_force);
  if (!lists_valid) build_ghosts_and_lists();
  double *pe_d = scal + 32;
  if (bond_overlap()) {
    // (Starting the chain earlier, behind the list build and underneath the CG iterations, was measured: the matrix passes that share the GPU with it
    //  slow down by exactly what the chain saves -- 48.2-48.5 against 48.0-48.7 ms per step, the pass 1.02 against 0.82 ms -- so it starts here.)
    bonded_chain_begin();
    if (multi()) { on_comm_stream([&] { charge_halo(); }); join_comm_stream(); } else charge_halo();
    { const bool kt = kt_begin(&st.ms_k_nonbond, &st.ms_nonbond); nonbonded(true); kt_end(kt); }        // pot.F90:48-52
    { const bool kt = kt_begin(&st.ms_bond_exposed); RX_HIP(hipStreamWaitEvent(stream, ev_bond, 0)); kt_end(kt); }   // what of the bonded chain ENbond did not hide
    st.bond_overlap = 1;
    k_add_force3<<<nblk(N, 256), 256, 0, stream>>>(N, fnb[0], fnb[1], fnb[2], frc[0], frc[1], frc[2]);
    accumulate_stress(false);                            // pot.F90:65-72, before the ghost forces are folded back
    { const bool kt = kt_begin(&st.ms_fold); fold_ghost_forces(); kt_end(kt); }
    RX_HIP(hipMemcpyAsync(h_scal + 32, pe_d, sizeof(double) * 16, hipMemcpyDeviceToHost, stream));
    outer_end(t_force);
    force_pending = true;
    if (!defer_host_read) finish_force();
    return;
  }
  RX_HIP(hipMemsetAsync(pe_d, 0, sizeof(double) * 16, stream));
  // the ghost-charge halo needs nothing from the bond orders and they need no charges: on a multi-rank run the exchange goes to
  // the second stream and meets the main stream again in front of the nonbonded kernel
  if (multi()) on_comm_stream([&] { charge_halo(); }); else charge_halo();
  { const bool kt = kt_begin(&st.ms_k_bondorder, &st.ms_bo); bond_orders(); kt_end(kt); }
  if (multi()) join_comm_stream();
  { const bool kt = kt_begin(&st.ms_k_nonbond, &st.ms_nonbond); if (ff.pqeq) nonbonded_pqeq(); else nonbonded(); kt_end(kt); }     // pot.F90:48-52
  const KtPair t_bonded = outer_begin(&st.ms_bonded);
  bonded_energies();
  if (ff.pqeq) efield_force();                         // pot.F90:61, before ForceBondedTerms
  { const bool kt = kt_begin(&st.ms_k_assemble); assemble_forces(); kt_end(kt); }
  accumulate_stress(false);                            // pot.F90:65-72, before the ghost forces are folded back
  { const bool kt = kt_begin(&st.ms_fold); fold_ghost_forces(); kt_end(kt); }
  outer_end(t_bonded);
  RX_HIP(hipMemcpyAsync(h_scal + 32, pe_d, sizeof(double) * 16, hipMemcpyDeviceToHost, stream));
  outer_end(t_force);
  force_pending = true;
  if (!defer_host_read) finish_force();
}
// the host side of FORCE: wait, energies, NaN trap.  step() calls it once behind its last step: the steps in between queue the next
// step's kernels behind FORCE without a host round trip (50-60 us of idle GPU per step until round 5)
void Engine::finish_force() {
  if (!force_pending) return;
  force_pending = false;
  sync_stream();
  pe[0] = 0.0;
  for (int k = 1; k < 14; ++k) { pe[k] = h_scal[32 + k]; pe[0] += pe[k]; }   // PE(0)=sum(PE(1:13)), main.F90:236
  collect_timers();
  if (!std::isfinite(pe[0])) throw EngineError(RXMD_E_NAN, "non-finite potential energy (degenerate geometry?)");
}

// ------------------------------------------------------------------------------------------------
// vkick (main.F90:192-207), extended-Lagrangian charges (main.F90:67-68,98) and the drift (main.F90:72)
__global__ void k_kick_drift(int N, DevFF ff, double dt, double lex_w2, const int *__restrict__ type,
                             double *__restrict__ vx, double *__restrict__ vy, double *__restrict__ vz,
                             const double *__restrict__ fx, const double *__restrict__ fy, const double *__restrict__ fz,
                             double *__restrict__ x, double *__restrict__ y, double *__restrict__ z,
                             const double *__restrict__ q, double *__restrict__ qsfp, double *__restrict__ qsfv) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const double dthm = dt * 0.5 / ff.atom[type[i]].mass;       // init.F90:106
  const double v0 = vx[i] + 1.0 * dthm * fx[i], v1 = vy[i] + 1.0 * dthm * fy[i], v2 = vz[i] + 1.0 * dthm * fz[i];
  vx[i] = v0; vy[i] = v1; vz[i] = v2;
  const double w = qsfv[i] + 0.5 * dt * lex_w2 * (q[i] - qsfp[i]);
  qsfv[i] = w;
  qsfp[i] = qsfp[i] + dt * w;
  x[i] = x[i] + dt * v0; y[i] = y[i] + dt * v1; z[i] = z[i] + dt * v2;
}
__global__ void k_kick(int N, DevFF ff, double dt, double lex_w2, const int *__restrict__ type,
                       double *__restrict__ vx, double *__restrict__ vy, double *__restrict__ vz,
                       const double *__restrict__ fx, const double *__restrict__ fy, const double *__restrict__ fz,
                       const double *__restrict__ q, const double *__restrict__ qsfp, double *__restrict__ qsfv) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const double dthm = dt * 0.5 / ff.atom[type[i]].mass;
  vx[i] = vx[i] + 1.0 * dthm * fx[i]; vy[i] = vy[i] + 1.0 * dthm * fy[i]; vz[i] = vz[i] + 1.0 * dthm * fz[i];
  qsfv[i] = qsfv[i] + 0.5 * dt * lex_w2 * (q[i] - qsfp[i]);
}

// ------------------------------------------------------------------------------------------------
// velocity scaling of the MD loop head (main.F90:45-61; ScaleTemperature :722-763, AdjustTemperature :684-719,
// LinearMomentum :766-797) and the momentum removal of every step under an electric field (main.F90:70-71).  Everything stays on the
// device and in stream order (round 6; until then: per type two kernels + a copy + a host wait, then the factors on the host -- seven host
// waits per e-field step of the SiC force field, which broke the run-ahead of the whole step):
//   k_type_sums        ONE launch forms the 6 sums (count, kinetic energy, momentum x3, mass) of every atom type: per type a grid-stride
//                      pass with a fixed per-thread order, a tree sum per workgroup, partials[(t * 6 + c) * nblocks + block]
//   k_type_sums_final  a thread per (type, component) adds the workgroup partials in block order -> tsum[t * 6 + c]
//   (vprocs > 1: one all-reduce of the 6 (nso + 1) sums -- MPI_ALLREDUCE of main.F90:699,738,783 -- RCCL in stream order)
//   k_scale_coeffs     one thread: the per-type factors and the centre-of-mass velocity of the mode -> ScaleArgs in device memory
//   k_scale_velocities v = c(type) v - vcm
constexpr int TSUM_BLOCKS = 240;
__global__ void __launch_bounds__(256) k_type_sums(int n, int nso, const int *__restrict__ type, DevFF ff, const double *__restrict__ vx, const double *__restrict__ vy,
                                                    const double *__restrict__ vz, double *__restrict__ partials) {
  __shared__ double sm[256];
  for (int tsel = 1; tsel <= nso; ++tsel) {
    double a[6] = {0, 0, 0, 0, 0, 0};
    const double m = ff.atom[tsel].mass;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
      if (type[i] != tsel) continue;
      const double v0 = vx[i], v1 = vy[i], v2 = vz[i];
      a[0] += 1.0; a[1] += 0.5 * m * (v0 * v0 + v1 * v1 + v2 * v2); a[2] += m * v0; a[3] += m * v1; a[4] += m * v2; a[5] += m;
    }
    for (int c = 0; c < 6; ++c) {
      sm[threadIdx.x] = a[c];
      __syncthreads();
      for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s]; __syncthreads(); }
      if (threadIdx.x == 0) partials[(tsel * 6 + c) * gridDim.x + blockIdx.x] = sm[0];
      __syncthreads();
    }
  }
}
__global__ void k_type_sums_final(int nblocks, int nso, const double *__restrict__ partials, double *__restrict__ tsum) {
  const int e = threadIdx.x;                       // t * 6 + c
  if (e >= 6 * (nso + 1)) return;
  double s = 0.0;
  if (e >= 6) for (int b = 0; b < nblocks; ++b) s += partials[e * nblocks + b];
  tsum[e] = s;
}
__global__ void k_sum6(int nblocks, const double *__restrict__ partials, double *__restrict__ out6) {
  const int c = threadIdx.x;
  if (c >= 6) return;
  double s = 0.0;
  for (int b = 0; b < nblocks; ++b) s += partials[b * 6 + c];
  out6[c] = s;
}
// PRINTE's two sums over the residents (main.F90:225-230): kinetic energy sum hmas(ity) v.v and total charge; fixed grid and a fixed
// order of the final sum, so the numbers do not depend on scheduling
__global__ void __launch_bounds__(256) k_ke_qsum(int n, const int *__restrict__ type, DevFF ff, const double *__restrict__ vx, const double *__restrict__ vy,
                                                  const double *__restrict__ vz, const double *__restrict__ q, double *__restrict__ partials) {
  __shared__ double sm[256];
  double a[2] = {0.0, 0.0};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const double v0 = vx[i], v1 = vy[i], v2 = vz[i];
    a[0] += 0.5 * ff.atom[type[i]].mass * (v0 * v0 + v1 * v1 + v2 * v2);       // hmas = mass / 2, init.F90:106
    a[1] += q[i];
  }
  for (int c = 0; c < 2; ++c) {
    sm[threadIdx.x] = a[c];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) partials[blockIdx.x * 6 + c] = sm[0];
    __syncthreads();
  }
  if (threadIdx.x < 4) partials[blockIdx.x * 6 + 2 + threadIdx.x] = 0.0;
}
// INITVELOCITY (init.F90:292-360), first half: unit-variance Gaussian velocity components, the same distribution for every element as the
// reference draws them (its Box-Muller pairs atoms i, i+1 off one random_number stream; which atom gets which deviate is not defined
// beyond that).  Here a counter-based generator keyed by (seed, draw, GLOBAL atom id, component): the velocities of an atom do not depend
// on the rank that owns it, on its local index or on the number of ranks.
__device__ inline unsigned long long mix64(unsigned long long z) {     // splitmix64 finaliser
  z += 0x9E3779B97F4A7C15ULL; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31);
}
__global__ void k_random_velocities(int n, unsigned long long seed, unsigned long long draw, const long long *__restrict__ gid,
                                    double *__restrict__ vx, double *__restrict__ vy, double *__restrict__ vz) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long key = mix64(seed ^ mix64(draw)) ^ mix64(static_cast<unsigned long long>(gid[i]) * 0xD1342543DE82EF95ULL);
  double v[3];
  for (int k = 0; k < 3; ++k) {
    const unsigned long long a = mix64(key + 2ULL * k), b = mix64(key + 2ULL * k + 1ULL);
    const double u1 = (static_cast<double>(a >> 11) + 0.5) * (1.0 / 9007199254740992.0);      // (0, 1)
    const double u2 = (static_cast<double>(b >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    v[k] = sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925 * u2);                          // Box-Muller
  }
  vx[i] = v[0]; vy[i] = v[1]; vz[i] = v[2];
}
struct ScaleArgs { double c[16]; double vcm[3]; };
// the factors of one velocity-scaling action from the (all-reduced) per-type sums; mdmode -1: LinearMomentum alone (c = 1)
__global__ void k_scale_coeffs(int mdmode, int nso, double treq, double vsfact, double gke, const double *__restrict__ sums, ScaleArgs *__restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double UTEMP0 = 503.398008, UTEMP = UTEMP0 * 2.0 / 3.0;      // module.F90:198-199
  ScaleArgs sa;
  for (int t = 0; t < 16; ++t) sa.c[t] = 1.0;
  sa.vcm[0] = sa.vcm[1] = sa.vcm[2] = 0.0;
  double ntot = 0, ektot = 0, mtot = 0;
  for (int t = 1; t <= nso; ++t) { ntot += sums[6 * t]; ektot += sums[6 * t + 1]; mtot += sums[6 * t + 5]; }
  bool remove_momentum = mdmode == -1;
  if (mdmode == 4) { for (int t = 0; t < 16; ++t) sa.c[t] = vsfact; }
  else if (mdmode == 5) {
    const double g = gke > 0.0 ? gke : ektot / ntot;                 // GKE of the last PRINTE, main.F90:49,250
    const double c = sqrt((treq * UTEMP0) / (g * UTEMP));
    for (int t = 0; t < 16; ++t) sa.c[t] = c;
  } else if (mdmode == 8) {
    const double c = sqrt((treq * UTEMP0) / (ektot / ntot * UTEMP));
    if (fabs(c - 1.0) > 0.05) {                                      // within 5 %: the velocities are left alone (main.F90:704-705): c = 1, vcm = 0
      for (int t = 0; t < 16; ++t) sa.c[t] = c;
      remove_momentum = true;
    }
  } else if (mdmode == 7) {
    for (int t = 1; t <= nso; ++t) {
      const double n = sums[6 * t];
      sa.c[t] = n > 1.0 ? sqrt((treq * UTEMP0) / (sums[6 * t + 1] / n * UTEMP)) : 0.0;   // main.F90:742-751
    }
    remove_momentum = true;
  } else if (mdmode == 0 || mdmode == 6) {
    // centre-of-mass velocity off (MPI_ALLREDUCE of sum m v and sum m, init.F90:333-341), then every velocity scaled so that the kinetic energy
    // per atom is 1.5 treq (init.F90:343-358).  KE of the shifted velocities from the sums at hand: sum m/2 |v - c|^2 = KE - |P|^2 / (2 M).
    double P2 = 0.0, Pa[3] = {0, 0, 0};
    for (int a = 0; a < 3; ++a) { for (int t = 1; t <= nso; ++t) Pa[a] += sums[6 * t + 2 + a]; P2 += Pa[a] * Pa[a]; }
    const double gke_new = (ektot - 0.5 * P2 / mtot) / ntot;
    const double vfactor = sqrt(1.5 * treq / gke_new);
    for (int t = 0; t < 16; ++t) sa.c[t] = vfactor;
    for (int a = 0; a < 3; ++a) sa.vcm[a] = vfactor * Pa[a] / mtot;
  }
  if (remove_momentum)                                               // LinearMomentum of the scaled velocities (main.F90:766-797)
    for (int a = 0; a < 3; ++a) {
      double p = 0.0;
      for (int t = 1; t <= nso; ++t) p += sa.c[t] * sums[6 * t + 2 + a];
      sa.vcm[a] = p / mtot;
    }
  *out = sa;
}
__global__ void k_scale_velocities(int n, const ScaleArgs *__restrict__ ap, const int *__restrict__ type, double *__restrict__ vx, double *__restrict__ vy, double *__restrict__ vz) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double c = ap->c[type[i]];
  vx[i] = c * vx[i] - ap->vcm[0]; vy[i] = c * vy[i] - ap->vcm[1]; vz[i] = c * vz[i] - ap->vcm[2];
}

void Engine::kinetic_and_charge(double &ke, double &qsum) {
  const int nb = 240;
  k_ke_qsum<<<nb, 256, 0, stream>>>(N, type, dff, vel[0], vel[1], vel[2], q, partials);
  k_sum6<<<1, 64, 0, stream>>>(nb, partials, scal + 56);
  RX_HIP(hipMemcpyAsync(h_scal + 56, scal + 56, sizeof(double) * 2, hipMemcpyDeviceToHost, stream));
  sync_stream();
  ke = h_scal[56]; qsum = h_scal[57];
}

// the per-type sums of the residents' velocities, summed over all ranks, in tsum (device memory) -- no host wait with the native transport
void Engine::type_sums_device() {
  k_type_sums<<<TSUM_BLOCKS, 256, 0, stream>>>(N, ff.nso, type, dff, vel[0], vel[1], vel[2], partials);
  k_type_sums_final<<<1, 128, 0, stream>>>(TSUM_BLOCKS, ff.nso, partials, tsum);
  if (nprocs > 1) {                                                  // MPI_ALLREDUCE, main.F90:699,738,783
    const int n = 6 * (ff.nso + 1);
    const bool kt = kt_begin(&st.ms_allreduce, nullptr, &st.allreduce_calls);
    struct End { Engine *e; bool kt; ~End() { e->kt_end(kt); } } end_{this, kt};
    if (nccl) { rccl_allreduce_dev(tsum, n); return; }               // in stream order
    if (!has_comm || !comm.allreduce_sum) throw EngineError(RXMD_E_COMM, "vprocs > 1 needs a transport: call rxmd_hip_set_comm or rxmd_hip_comm_init_rccl first");
    RX_HIP(hipMemcpyAsync(h_scal + 192, tsum, sizeof(double) * n, hipMemcpyDeviceToHost, stream));      // a host transport (MPI callbacks): its all-reduce IS a host call
    sync_stream();
    if (comm.allreduce_sum(comm.ctx, h_scal + 192, n)) throw EngineError(RXMD_E_COMM, "allreduce callback failed");
    RX_HIP(hipMemcpyAsync(tsum, h_scal + 192, sizeof(double) * n, hipMemcpyHostToDevice, stream));
  }
}

void Engine::thermostat(int mdmode, double treq_K, double vsfact, double gke) {
  if (!atoms_set) throw EngineError(RXMD_E_STATE, "atoms were never set");
  if (ff.nso > 15) throw EngineError(RXMD_E_ARG, "more than 15 atom types");
  if (!(mdmode == 0 || (mdmode >= 4 && mdmode <= 8))) throw EngineError(RXMD_E_ARG, "thermostat: mdmode must be 0, 4, 5, 6, 7 or 8");
  const double UTEMP0 = 503.398008;                                  // module.F90:198
  const double treq = treq_K / UTEMP0;                               // init.F90:72
  if (mdmode == 0 || mdmode == 6) {                                  // main.F90:54-55 -> INITVELOCITY: fresh Gaussian velocities before the sums
    const unsigned long long seed = static_cast<unsigned long long>(opt.seed);
    k_random_velocities<<<nblk(N, 256), 256, 0, stream>>>(N, seed, velocity_draws++, gid, vel[0], vel[1], vel[2]);
  }
  const bool need_sums = (mdmode == 0 || mdmode == 6 || mdmode == 7 || mdmode == 8 || (mdmode == 5 && gke <= 0.0));
  if (need_sums) type_sums_device();
  k_scale_coeffs<<<1, 64, 0, stream>>>(mdmode, ff.nso, treq, vsfact, gke, tsum, sargs);
  k_scale_velocities<<<nblk(N, 256), 256, 0, stream>>>(N, sargs, type, vel[0], vel[1], vel[2]);
}

__global__ void k_drift(int N, double dt, const double *__restrict__ vx, const double *__restrict__ vy, const double *__restrict__ vz,
                        double *__restrict__ x, double *__restrict__ y, double *__restrict__ z) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  x[i] = x[i] + dt * vx[i]; y[i] = y[i] + dt * vy[i]; z[i] = z[i] + dt * vz[i];
}
__global__ void k_lex_drift(int N, double dt, double *__restrict__ qsfp, const double *__restrict__ qsfv) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) qsfp[i] = qsfp[i] + dt * qsfv[i];
}

void Engine::remove_momentum() {                  // LinearMomentum (main.F90:766-797) inside the step: sums, all-reduce and factors in stream order
  type_sums_device();
  k_scale_coeffs<<<1, 64, 0, stream>>>(-1, ff.nso, 0.0, 1.0, 0.0, tsum, sargs);
  k_scale_velocities<<<nblk(N, 256), 256, 0, stream>>>(N, sargs, type, vel[0], vel[1], vel[2]);
}

void Engine::step(int nsteps) {
  if (!atoms_set) throw EngineError(RXMD_E_STATE, "atoms were never set");
  // The event pairs of the per-kernel timers (rxmd_stats.ms_k_*, ms_bo, ms_nonbond, ...) cost the stream ~5 us per event, ~12 pairs per step.  A call of 8
  // steps or more times them on every 4th step only and counts their milliseconds four-fold (bench.py's per-kernel table is such an average; the kernels
  // do the same work every step); shorter calls time every step.  The sections (QEq, FORCE, list build) keep their pair on every step.
  kt_every = nsteps >= 8 ? 4 : 1;
  struct Restore { Engine *e; ~Restore() { e->kt_every = 1; e->kt_phase = 0; } } restore_{this};
  for (int s = 0; s < nsteps; ++s) {
    kt_phase = s;
    if (cfg.efield_dir != 0) {             // always correct the linear momentum when an electric field is applied (main.F90:70-71)
      k_kick<<<nblk(N, 256), 256, 0, stream>>>(N, dff, dt, Lex_w2, type, vel[0], vel[1], vel[2], frc[0], frc[1], frc[2], q, qsfp, qsfv);
      k_lex_drift<<<nblk(N, 256), 256, 0, stream>>>(N, dt, qsfp, qsfv);
      remove_momentum();
      k_drift<<<nblk(N, 256), 256, 0, stream>>>(N, dt, vel[0], vel[1], vel[2], pos[0], pos[1], pos[2]);
    } else
      k_kick_drift<<<nblk(N, 256), 256, 0, stream>>>(N, dff, dt, Lex_w2, type, vel[0], vel[1], vel[2], frc[0], frc[1], frc[2], pos[0], pos[1], pos[2], q, qsfp, qsfv);
    { const bool kt = kt_begin(&st.ms_migrate); migrate(); kt_end(kt); }                                 // main.F90:75
    const int qs = cfg.qstep > 0 ? cfg.qstep : 1;
    if (step_count % qs == 0) qeq();                                     // main.F90:77-83
    force(true);                                                         // main.F90:84 (energies read once, behind the last step)
    accumulate_stress(true);                                             // main.F90:86-94
    k_kick<<<nblk(N, 256), 256, 0, stream>>>(N, dff, dt, Lex_w2, type, vel[0], vel[1], vel[2], frc[0], frc[1], frc[2], q, qsfp, qsfv);
    ++step_count;
  }
  finish_force();
  sync_stream();
}

}  // namespace rxmd
