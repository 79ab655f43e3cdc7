// bondorder.hip -- BOCALC on the device (reference src/bo.F90).
//   BOPRIM (bo.F90:28-118)  -> k_bo_prime + k_deltap : uncorrected BO' sigma/pi/pipi, d ln BO'/dr ; Delta'
//   BOFULL (bo.F90:121-298) -> k_bo_full + k_delta_lp : corrections f1,f4,f5 -> BO, A0..A3 ; Delta and the lone-pair
//                              preparation of Elnpr (pot.F90:183-209), which needs only the atom's own Delta
// Round 4: the bond tables are COMPACT (CSR): bond b = boff[i] + s is slot s of atom i (engine.h), and the two bond-order kernels run ONE LANE PER
// BOND END over b -- every load and store of a per-bond array is coalesced and no lane idles while a neighbour with more bonds loops (a
// thread-per-atom wavefront of RDX ran 11 rounds for 5.3 bonds per atom on average: k_bo_full was bound by its FP64 exponentials, not by bytes).
// A bond is evaluated from both ends (no mirrored scatter, no atomics): all the pair quantities are symmetric functions of the two ends,
// so both evaluations agree bit for bit.  The per-atom sums (Delta', Delta) are formed by a thread per atom in slot order -- the order of the
// thread-per-atom kernels these replace, so the sums keep their bits.
#include "engine.h"

namespace rxmd {

static inline int nblk(long long n, int b) { return n > 0 ? static_cast<int>((n + b - 1) / b) : 1; }   // an empty rank still launches (kernels guard their range)

__global__ void __launch_bounds__(256) k_bo_prime(int nbonds, DevFF ff, const int *__restrict__ bown, const int *__restrict__ nbr, const unsigned char *__restrict__ btype,
                                                   const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z, const int *__restrict__ type,
                                                   double *__restrict__ bo0, double *__restrict__ bo2, double *__restrict__ bo3,
                                                   double *__restrict__ dln2, double *__restrict__ dln3, double *__restrict__ dBOp) {
  const int o = xcd_swizzle(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  if (o >= nbonds) return;
  const int i = bown[o], j = nbr[o];
  const double cut = ff.cutoff_vpar30;
  const DevBondP bp = ff.bond[ff.inxn2[type[i] * ff.n1 + btype[o]]];
  const double d0 = x[i] - x[j], d1 = y[i] - y[j], d2 = z[i] - z[j];
  const double r2 = d0 * d0 + d1 * d1 + d2 * d2;
  const double a1 = bp.cBOp1 * pow(r2, bp.pbo2h), a2 = bp.cBOp3 * pow(r2, bp.pbo4h), a3 = bp.cBOp5 * pow(r2, bp.pbo6h);  // bo.F90:67-69
  double b1 = bp.sw0 * exp(a1), b2 = bp.sw1 * exp(a2), b3 = bp.sw2 * exp(a3);
  b1 = (1.0 + cut) * b1;                             // bo.F90:79
  double b0 = 0.0, l2 = 0.0, l3 = 0.0, dB = 0.0;
  if (b1 + b2 + b3 > cut) {                          // bo.F90:83
    const double l1 = bp.sw0 * bp.pbo2 * a1 / r2;
    l2 = bp.sw1 * bp.pbo4 * a2 / r2;
    l3 = bp.sw2 * bp.pbo6 * a3 / r2;
    dB = b1 * l1 + b2 * l2 + b3 * l3;                // bo.F90:91 (uses sigma' before the subtraction)
    b1 = b1 - cut;                                   // bo.F90:96
    b0 = b1 + b2 + b3;
  } else { b2 = 0.0; b3 = 0.0; }
  bo0[o] = b0; bo2[o] = b2; bo3[o] = b3; dln2[o] = l2; dln3[o] = l3; dBOp[o] = dB;
}

// Delta'(i) = -Val + sum over the bonds of i, in slot order (bo.F90:44,101)
__global__ void __launch_bounds__(256) k_deltap(int G, DevFF ff, const int *__restrict__ boff, const int *__restrict__ type, const double *__restrict__ bo0, double *__restrict__ deltap) {
  const int i = xcd_swizzle(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  if (i >= G) return;
  double sum = -ff.atom[type[i]].Val;                  // deltap(i,1) = -Val, bo.F90:44
  for (int o = boff[i], o1 = boff[i + 1]; o < o1; ++o) sum += bo0[o];
  deltap[i] = sum;
}

__global__ void __launch_bounds__(256) k_bo_full(int nbonds, int nb_res, DevFF ff, const int *__restrict__ bown, const int *__restrict__ nbr, const unsigned char *__restrict__ btype, const int *__restrict__ type,
                                                  const double *__restrict__ deltap, double *__restrict__ bo0, double *__restrict__ bo1, double *__restrict__ bo2, double *__restrict__ bo3,
                                                  double *__restrict__ A0, double *__restrict__ A1, double *__restrict__ A2, double *__restrict__ A3,
                                                  double *__restrict__ cf1, double *__restrict__ cf2, double *__restrict__ cf3, double *__restrict__ cdn,
                                                  double *__restrict__ fnx, double *__restrict__ fny, double *__restrict__ fnz,
                                                  double *__restrict__ etor, double *__restrict__ econ, double *__restrict__ epen) {
  const int o = xcd_swizzle(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  if (o >= nbonds) return;
  const int i = bown[o], j = nbr[o];
  const int ti = type[i], tj = btype[o];
  const DevAtomP ai = ff.atom[ti], aj = ff.atom[tj];
  const double dpi = deltap[i];
  const double dp2i = dpi + ai.Val - ai.Valval;        // deltap(i,2), bo.F90:151
  const double e1i = exp(-ff.vpar1 * dpi), e2i = exp(-ff.vpar2 * dpi);
  const DevBondP bp = ff.bond[ff.inxn2[ti * ff.n1 + tj]];
  const double dpj = deltap[j], dp2j = dpj + aj.Val - aj.Valval;
  const double e1j = exp(-ff.vpar1 * dpj), e2j = exp(-ff.vpar2 * dpj);
  const double fn2 = e1i + e1j;
  const double fn3 = (-1.0 / ff.vpar2) * log(0.5 * (e2i + e2j));
  const double fn23 = fn2 + fn3;
  const double BOp0 = bo0[o];
  double fn1 = 0.5 * ((ai.Val + fn2) / (ai.Val + fn23) + (aj.Val + fn2) / (aj.Val + fn23));
  if (bp.ovc < 1e-3) fn1 = 1.0;
  const double BOpsqr = BOp0 * BOp0;
  const double pboc34 = bp.pboc3 * bp.pboc4;
  const double u45i = bp.pboc5 + bp.pboc3 * dp2i - pboc34 * BOpsqr;   // bo.F90:242-243
  const double u45j = bp.pboc5 + bp.pboc3 * dp2j - pboc34 * BOpsqr;
  double fn4 = 1.0 / (1.0 + exp(-bp.pboc3 * (bp.pboc4 * BOpsqr - dp2i) + bp.pboc5));
  double fn5 = 1.0 / (1.0 + exp(-bp.pboc3 * (bp.pboc4 * BOpsqr - dp2j) + bp.pboc5));
  if (bp.v13cor < 1e-3) { fn4 = 1.0; fn5 = 1.0; }
  const double fn45 = fn4 * fn5, fn145 = fn1 * fn45, fn1145 = fn1 * fn145;
  double B0 = BOp0 * fn145, B2 = bo2[o] * fn1145, B3 = bo3[o] * fn1145;
  if (B0 < 1e-10) B0 = 0.0;
  if (B2 < 1e-10) B2 = 0.0;
  if (B3 < 1e-10) B3 = 0.0;
  const double B1 = B0 - B2 - B3;                    // bo.F90:215
  const double u1i = ai.Val + fn23, u1j = aj.Val + fn23;
  const double u1i_inv2 = 1.0 / (u1i * u1i), u1j_inv2 = 1.0 / (u1j * u1j);
  const double Cf1A = 0.5 * fn3 * (u1i_inv2 + u1j_inv2);
  const double Cf1B = -0.5 * ((u1i - fn3) * u1i_inv2 + (u1j - fn3) * u1j_inv2);
  double Cf1i = (-Cf1A * ff.vpar1 * e1i) + (Cf1B * e2i) / (e2i + e2j);     // pboc1 == vpar(1), param.F90:174
  const double x45i = exp(u45i), x45j = exp(u45j);
  const double p1 = 1.0 / (1.0 + x45i), p2 = 1.0 / (1.0 + x45j), p12 = p1 * p2;
  double Cf45i = -x45i * p12 * p1, Cf45j = -x45j * p12 * p2;
  if (bp.ovc < 1e-3) Cf1i = 0.0;
  if (bp.v13cor < 1e-3) { Cf45i = 0.0; Cf45j = 0.0; }
  const double fn45_inv = 1.0 / fn45, Cf1i_div1 = Cf1i / fn1;
  const double a2v = Cf1i_div1 + (bp.pboc3 * Cf45i * fn45_inv);
  bo0[o] = B0; bo1[o] = B1; bo2[o] = B2; bo3[o] = B3;
  A0[o] = fn145;
  A1[o] = -2.0 * pboc34 * BOp0 * (Cf45i + Cf45j) * fn45_inv;
  A2[o] = a2v;
  A3[o] = a2v + Cf1i_div1;
  // accumulators of pot.F90:20-26: cleared here -- except cf1..3 and cdn of the residents' bonds, which their first writers SET (k_ebond_terms, k_elnpr_bonds)
  if (o >= nb_res) { cf1[o] = 0.0; cf2[o] = 0.0; cf3[o] = 0.0; cdn[o] = 0.0; }
  fnx[o] = 0.0; fny[o] = 0.0; fnz[o] = 0.0;
  // exponentials of this bond that every angle / torsion through it re-uses (global parameters only):
  const double bs = B0 - 1e-4;                                   // BO - cutof2_esub, pot.F90:372,1022
  etor[o] = exp(-ff.ptor2 * bs);                           // exp_tor2, pot.F90:1086-1088
  econ[o] = exp(-ff.pcot2 * ((bs - 1.5) * (bs - 1.5)));    // factor of fn12, pot.F90:1097-1099
  epen[o] = exp(-ff.ppen2 * ((bs - 2.0) * (bs - 2.0)));    // exp_pen2, pot.F90:463-464
}

// Delta(i) (bo.F90:294) in slot order, and the lone-pair preparation (pot.F90:183-209); clears what FORCE clears per call (pot.F90:20-26)
__global__ void __launch_bounds__(256) k_delta_lp(int G, DevFF ff, const int *__restrict__ boff, const int *__restrict__ type, const double *__restrict__ bo0,
                                                   double *__restrict__ delta, double *__restrict__ nlp, double *__restrict__ dDlp, double *__restrict__ deltalp, double *__restrict__ cds,
                                                   double *__restrict__ fx, double *__restrict__ fy, double *__restrict__ fz) {
  const int i = xcd_swizzle(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  if (i >= G) return;
  const DevAtomP ai = ff.atom[type[i]];
  double sum = 0.0;
  for (int o = boff[i], o1 = boff[i + 1]; o < o1; ++o) sum += bo0[o];
  const double dl = -ai.Val + sum;                     // bo.F90:294
  delta[i] = dl;
  // lone-pair preparation, pot.F90:183-209
  const double deltaE = -ai.Vale + ai.Val + dl;
  const int idEh = static_cast<int>(deltaE * 0.5);     // is_idEh = 1, int() truncates toward zero
  const double u = 2.0 + deltaE - 2 * idEh;
  const double explp1 = exp(-ff.plp1 * (u * u));
  dDlp[i] = 2.0 * ff.plp1 * explp1 * u;
  const double nl = explp1 - static_cast<double>(idEh);
  nlp[i] = nl;
  deltalp[i] = (ai.mass > 21.0) ? 0.0 : (ai.nlpopt - nl);
  cds[i] = 0.0; fx[i] = 0.0; fy[i] = 0.0; fz[i] = 0.0;
}

void Engine::bond_orders() {
  k_bo_prime<<<nblk(nbonds, 256), 256, 0, stream>>>(nbonds, dff, bown, nbr, btype, pos[0], pos[1], pos[2], type, bo0, bo2, bo3, dln2, dln3, dBOp);
  k_deltap<<<nblk(G, 256), 256, 0, stream>>>(G, dff, boff, type, bo0, deltap);
  k_bo_full<<<nblk(nbonds, 256), 256, 0, stream>>>(nbonds, nbonds_res, dff, bown, nbr, btype, type, deltap, bo0, bo1, bo2, bo3, A0, A1, A2, A3, cf1, cf2, cf3, cdn, fnx, fny, fnz, etor, econ, epen);
  k_delta_lp<<<nblk(G, 256), 256, 0, stream>>>(G, dff, boff, type, bo0, delta, nlp, dDlp, deltalp, cds, frc[0], frc[1], frc[2]);
}

}  // namespace rxmd
