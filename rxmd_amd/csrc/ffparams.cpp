// ffparams.cpp -- see ffparams.h.  Host only (no HIP).
#include "ffparams.h"
#include <sstream>

#include <cctype>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <stdexcept>

namespace rxmd {
namespace {

// One line of a Fortran fixed-format record.  The reference reads ffield with formats such as
// (2i3,8f9.4) (param.F90:344-351): fields are taken by COLUMN, blanks read as zero, short lines are
// blank padded, and an F field without a decimal point has its last `d` digits as the fraction.
class FixedLine {
 public:
  explicit FixedLine(std::string s) : s_(std::move(s)) {
    while (!s_.empty() && (s_.back() == '\n' || s_.back() == '\r')) s_.pop_back();
  }
  void skip(int w) { col_ += w; }
  int integer(int w) {
    std::string t = take(w);
    return t.empty() ? 0 : std::atoi(t.c_str());
  }
  double real(int w, int d) {
    std::string t = take(w);
    if (t.empty()) return 0.0;
    bool dot = false, expo = false, digit = false;
    for (char &c : t) {
      if (c == '.') dot = true;
      if (c == 'd' || c == 'D') c = 'e';
      if (c == 'e' || c == 'E') expo = true;
      if (std::isdigit(static_cast<unsigned char>(c))) digit = true;
    }
    if (!digit) return 0.0;
    double v = std::strtod(t.c_str(), nullptr);
    if (!dot && !expo) v /= std::pow(10.0, d);
    return v;
  }
  std::string text(int w) { return take(w); }

 private:
  std::string take(int w) {
    std::string out;
    for (int c = col_; c < col_ + w && c < static_cast<int>(s_.size()); ++c)
      if (s_[c] != ' ' && s_[c] != '\t') out.push_back(s_[c]);
    col_ += w;
    return out;
  }
  std::string s_;
  int col_ = 0;
};

std::string next_line(std::ifstream &in, const std::string &path) {
  std::string l;
  if (!std::getline(in, l)) throw std::runtime_error("ffield '" + path + "': unexpected end of file");
  return l;
}

}  // namespace

void ForceField::parse(const std::string &path, bool lg_format) {
  lg = lg_format;
  std::ifstream in(path);
  if (!in) throw std::runtime_error("cannot open ffield '" + path + "'");
  header = next_line(in, path);
  int npar = std::atoi(next_line(in, path).c_str());  // list-directed read, param.F90:42
  if (npar < 39 || npar > 1000) throw std::runtime_error("ffield: implausible number of general parameters");
  vpar.assign(npar + 1, 0.0);
  for (int i = 1; i <= npar; ++i) vpar[i] = FixedLine(next_line(in, path)).real(10, 4);
  pvdW1 = vpar[29]; vpar30 = vpar[30]; vpar1 = vpar[1]; vpar2 = vpar[2];
  plp1 = vpar[16]; povun3 = vpar[33]; povun4 = vpar[32]; povun6 = vpar[7]; povun7 = vpar[9]; povun8 = vpar[10];
  pval6 = vpar[15]; pval8 = vpar[34]; pval9 = vpar[17]; pval10 = vpar[18];
  ppen2 = vpar[20]; ppen3 = vpar[21]; ppen4 = vpar[22]; pcoa2 = vpar[3]; pcoa3 = vpar[39]; pcoa4 = vpar[31];
  ptor2 = vpar[24]; ptor3 = vpar[25]; ptor4 = vpar[26]; pcot2 = vpar[28];

  std::vector<double> lg_diag;
  nso = FixedLine(next_line(in, path)).integer(3);
  if (nso < 1 || nso > 30) throw std::runtime_error("ffield: bad number of atom types");
  for (int k = 0; k < 3; ++k) next_line(in, path);
  atom.assign(nso + 1, AtomTypeParams{});
  for (int t = 1; t <= nso; ++t) {
    AtomTypeParams &a = atom[t];
    FixedLine l1(next_line(in, path));
    l1.skip(1); a.name = l1.text(2);
    a.rat = l1.real(9, 4); a.Val = l1.real(9, 4); a.mass = l1.real(9, 4); a.rvdw1 = l1.real(9, 4);
    a.eps = l1.real(9, 4); a.gam = l1.real(9, 4); a.rapt = l1.real(9, 4); a.Vale = l1.real(9, 4);
    FixedLine l2(next_line(in, path));
    l2.skip(3); a.alf = l2.real(9, 4); a.vop = l2.real(9, 4); a.Valboc = l2.real(9, 4); a.povun5 = l2.real(9, 4);
    l2.skip(9); a.chi = l2.real(9, 4); a.eta = l2.real(9, 4);
    FixedLine l3(next_line(in, path));
    l3.skip(3); a.vnq = l3.real(9, 4); a.plp2 = l3.real(9, 4); l3.skip(9);
    a.bo131 = l3.real(9, 4); a.bo132 = l3.real(9, 4); a.bo133 = l3.real(9, 4);
    FixedLine l4(next_line(in, path));
    l4.skip(3); a.povun2 = l4.real(9, 4); a.pval3 = l4.real(9, 4); l4.skip(9); a.Valval = l4.real(9, 4); a.pval5 = l4.real(9, 4);
    if (lg) {                                              // param.F90:107-109
      a.rcore2 = l4.real(9, 4); a.ecore2 = l4.real(9, 4); a.acore2 = l4.real(9, 4);
      FixedLine l5(next_line(in, path));
      l5.skip(3); lg_diag.push_back(l5.real(9, 4)); a.Re_lg = l5.real(9, 4);
    }
  }
  for (int t = 1; t <= nso; ++t) {
    AtomTypeParams &a = atom[t];
    if (a.mass < 21.0 && a.Valboc != a.Valval) a.Valboc = a.Valval;  // param.F90:117-119
    a.nlpopt = 0.5 * (a.Vale - a.Val);
    a.Valangle = a.Valboc;
  }
  const int n = n1();
  r0s.assign(n * n, 0); r0p = r0pp = rvdW = Dij = alpij = gamW = gamij = r0s;
  for (int a = 1; a <= nso; ++a)
    for (int b = 1; b <= nso; ++b) {
      const int k = pair(a, b);
      r0s[k] = 0.5 * (atom[a].rat + atom[b].rat);
      r0p[k] = 0.5 * (atom[a].rapt + atom[b].rapt);
      r0pp[k] = 0.5 * (atom[a].vnq + atom[b].vnq);
      rvdW[k] = std::sqrt(4.0 * atom[a].rvdw1 * atom[b].rvdw1);
      Dij[k] = std::sqrt(atom[a].eps * atom[b].eps);
      alpij[k] = std::sqrt(atom[a].alf * atom[b].alf);
      gamW[k] = std::sqrt(atom[a].vop * atom[b].vop);
      gamij[k] = std::pow(atom[a].gam * atom[b].gam, -1.5);
    }
  if (lg) {                                                // param.F90:140-145; C_lg pairs no row names stay 0 (unset in the reference, :83)
    C_lg.assign(n * n, 0); rcore = ecore = acore = C_lg;
    for (int a = 1; a <= nso; ++a) {
      C_lg[pair(a, a)] = lg_diag[a - 1];
      for (int b = 1; b <= nso; ++b) {
        const int k = pair(a, b);
        rcore[k] = std::sqrt(atom[a].rcore2 * atom[b].rcore2);
        ecore[k] = std::sqrt(atom[a].ecore2 * atom[b].ecore2);
        acore[k] = std::sqrt(atom[a].acore2 * atom[b].acore2);
      }
    }
  }

  nboty = FixedLine(next_line(in, path)).integer(3);
  if (nboty < 1 || nboty > 1000) throw std::runtime_error("ffield: bad number of bond types");
  next_line(in, path);
  bond.assign(nboty + 1, BondTypeParams{});
  inxn2.assign(n * n, 0);
  for (int r = 1; r <= nboty; ++r) {
    BondTypeParams &b = bond[r];
    FixedLine l1(next_line(in, path));
    int ta = l1.integer(3), tb = l1.integer(3);
    b.Desig = l1.real(9, 4); b.Depi = l1.real(9, 4); b.Depipi = l1.real(9, 4); b.pbe1 = l1.real(9, 4);
    b.pbo5 = l1.real(9, 4); b.v13cor = l1.real(9, 4); b.pbo6 = l1.real(9, 4); b.povun1 = l1.real(9, 4);
    FixedLine l2(next_line(in, path));
    l2.skip(6); b.pbe2 = l2.real(9, 4); b.pbo3 = l2.real(9, 4); b.pbo4 = l2.real(9, 4); b.bom = l2.real(9, 4);
    b.pbo1 = l2.real(9, 4); b.pbo2 = l2.real(9, 4); b.ovc = l2.real(9, 4);
    if (ta < 1 || ta > nso || tb < 1 || tb > nso) throw std::runtime_error("ffield: bond row names an unknown atom type");
    inxn2[pair(ta, tb)] = r; inxn2[pair(tb, ta)] = r;
  }
  for (int a = 1; a <= nso; ++a)
    for (int b = 1; b <= nso; ++b)
      if (int r = ix2(a, b)) {
        bond[r].pboc3 = std::sqrt(atom[a].bo132 * atom[b].bo132);
        bond[r].pboc4 = std::sqrt(atom[a].bo131 * atom[b].bo131);
        bond[r].pboc5 = std::sqrt(atom[a].bo133 * atom[b].bo133);
      }

  const int nodm = FixedLine(next_line(in, path)).integer(3);
  for (int r = 0; r < nodm; ++r) {
    FixedLine l(next_line(in, path));
    int a = l.integer(3), b = l.integer(3);
    double de = l.real(9, 4), ro = l.real(9, 4), go = l.real(9, 4), rs = l.real(9, 4), rp = l.real(9, 4), rpp = l.real(9, 4);
    if (a < 1 || a > nso || b < 1 || b > nso) throw std::runtime_error("ffield: off-diagonal row names an unknown atom type");
    auto put = [&](std::vector<double> &v, double x) { v[pair(a, b)] = x; v[pair(b, a)] = x; };
    if (lg) put(C_lg, l.real(9, 4));                       // param.F90:197-200
    if (rs > 0) put(r0s, rs);
    if (rp > 0) put(r0p, rp);
    if (rpp > 0) put(r0pp, rpp);
    if (ro > 0) put(rvdW, 2.0 * ro);
    if (de > 0) put(Dij, de);
    if (go > 0) put(alpij, go);
  }
  for (int a = 1; a <= nso; ++a)
    for (int b = 1; b <= nso; ++b) {
      const int r = ix2(a, b);
      if (!r) continue;
      BondTypeParams &p = bond[r];
      if (atom[a].rat > 0 && atom[b].rat > 0) p.sw[0] = 1;
      if (atom[a].rapt > 0 && atom[b].rapt > 0) p.sw[1] = 1;
      if (atom[a].vnq > 0 && atom[b].vnq > 0) p.sw[2] = 1;
      const int k = pair(a, b);
      p.cBOp1 = r0s[k] <= 0 ? 0.0 : p.pbo1 / std::pow(r0s[k], p.pbo2);
      p.cBOp3 = r0p[k] <= 0 ? 0.0 : p.pbo3 / std::pow(r0p[k], p.pbo4);
      p.cBOp5 = r0pp[k] <= 0 ? 0.0 : p.pbo5 / std::pow(r0pp[k], p.pbo6);
      p.pbo2h = 0.5 * p.pbo2; p.pbo4h = 0.5 * p.pbo4; p.pbo6h = 0.5 * p.pbo6;
    }

  nvaty = FixedLine(next_line(in, path)).integer(3);
  angle.assign(nvaty + 1, AngleTypeParams{});
  inxn3.assign(n * n * n, 0);
  const double pi = 3.14159265358979;  // the reference's literal, module.F90:90
  for (int r = 1; r <= nvaty; ++r) {
    FixedLine l(next_line(in, path));
    int a = l.integer(3), b = l.integer(3), c = l.integer(3);
    AngleTypeParams &p = angle[r];
    p.theta00 = l.real(9, 4); p.pval1 = l.real(9, 4); p.pval2 = l.real(9, 4); p.pcoa1 = l.real(9, 4);
    p.pval7 = l.real(9, 4); p.ppen1 = l.real(9, 4); p.pval4 = l.real(9, 4);
    p.theta00 = (pi / 180.0) * p.theta00;
    if (a < 1 || a > nso || b < 1 || b > nso || c < 1 || c > nso) throw std::runtime_error("ffield: angle row names an unknown atom type");
    inxn3[(a * n + b) * n + c] = r; inxn3[(c * n + b) * n + a] = r;
  }

  ntoty = FixedLine(next_line(in, path)).integer(3);
  tors.assign(ntoty + 1, TorsTypeParams{});
  inxn4.assign(n * n * n * n, 0);
  auto i4 = [&](int a, int b, int c, int d) -> int & { return inxn4[((a * n + b) * n + c) * n + d]; };
  for (int r = 1; r <= ntoty; ++r) {
    FixedLine l(next_line(in, path));
    int a = l.integer(3), b = l.integer(3), c = l.integer(3), d = l.integer(3);
    TorsTypeParams &p = tors[r];
    p.V1 = l.real(9, 4); p.V2 = l.real(9, 4); p.V3 = l.real(9, 4); p.ptor1 = l.real(9, 4); p.pcot1 = l.real(9, 4);
    if (b < 1 || b > nso || c < 1 || c > nso) throw std::runtime_error("ffield: torsion row names an unknown atom type");
    if (a == 0) {  // wildcard row fills what is still unset, param.F90:304-314
      for (int x = 1; x <= nso; ++x)
        for (int y = 1; y <= nso; ++y)
          if (i4(x, b, c, y) == 0 && i4(x, c, b, y) == 0) { i4(x, b, c, y) = r; i4(y, b, c, x) = r; i4(x, c, b, y) = r; i4(y, c, b, x) = r; }
    } else {
      i4(a, b, c, d) = r; i4(d, b, c, a) = r; i4(a, c, b, d) = r; i4(d, c, b, a) = r;
    }
  }

  nhbty = FixedLine(next_line(in, path)).integer(3);
  hb.assign(nhbty + 1, HbTypeParams{});
  inxn3hb.assign(n * n * n, 0);
  for (int r = 1; r <= nhbty; ++r) {
    FixedLine l(next_line(in, path));
    int a = l.integer(3), b = l.integer(3), c = l.integer(3);
    HbTypeParams &p = hb[r];
    p.r0hb = l.real(9, 4); p.phb1 = l.real(9, 4); p.phb2 = l.real(9, 4); p.phb3 = l.real(9, 4);
    if (a < 1 || a > nso || b < 1 || b > nso || c < 1 || c > nso) throw std::runtime_error("ffield: hbond row names an unknown atom type");
    inxn3hb[(a * n + b) * n + c] = r;  // not symmetric, param.F90:336
  }
  for (int t = 1; t <= nso; ++t) atom[t].eta *= 2.0;  // param.F90:361
}

void ForceField::compute_cutoffs(const std::vector<long long> &natoms_per_type) {
  cutoff_vpar30 = 1e-3 * vpar30;  // cutof2_bo * vpar30, module.F90:64, init.F90:371
  for (int a = 1; a <= nso; ++a)
    for (int b = a; b <= nso; ++b) {
      const int r = ix2(a, b);
      if (!r) continue;
      double dr = 1.0, bos = 1.0;
      while (bos > 1e-3) {  // MINBOSIG, init.F90:387-392
        dr = dr + 0.01;
        bos = std::exp(bond[r].pbo1 * std::pow(dr / r0s[pair(a, b)], bond[r].pbo2));
      }
      bond[r].rc = dr;
      bond[r].rc2 = dr * dr;
    }
  std::vector<double> rc(nboty + 1);
  for (int r = 1; r <= nboty; ++r) rc[r] = bond[r].rc;
  for (int a = 1; a <= nso; ++a)
    if (natoms_per_type[a] == 0)
      for (int b = 1; b <= nso; ++b) {
        if (int r = ix2(a, b)) rc[r] = 0.0;
        if (int r = ix2(b, a)) rc[r] = 0.0;
      }
  maxrc = 0.0;
  for (int r = 1; r <= nboty; ++r) maxrc = std::max(maxrc, rc[r]);
}

void ForceField::build_taper(double rc) {
  rctap = rc; rctap2 = rc * rc;
  auto ipow = [](double a, int b) { double r = 1.0; while (true) { if (b & 1) r *= a; b /= 2; if (!b) break; a *= a; } return r; };
  CTap[0] = 1.0; CTap[1] = CTap[2] = CTap[3] = 0.0;
  CTap[4] = -35.0 / ipow(rc, 4); CTap[5] = 84.0 / ipow(rc, 5); CTap[6] = -70.0 / ipow(rc, 6); CTap[7] = 20.0 / ipow(rc, 7);
}

void ForceField::build_tables() {
  const size_t stride = NTABLE + 2;
  tblEvdw.assign((nboty + 1) * stride, 0.0);
  tbldEvdw = tblEclmb = tbldEclmb = tblQEq = tblEvdw;
  UDR = rctap2 / NTABLE; UDRi = 1.0 / UDR;
  const double Cclmb0 = 332.0638, Cclmb0_qeq = 14.4;  // module.F90:681-682
  const double pvdW1h = 0.5 * pvdW1, pvdW1inv = 1.0 / pvdW1;
  for (int a = 1; a <= nso; ++a)
    for (int b = a; b <= nso; ++b) {
      const int r = ix2(a, b);
      if (!r) continue;
      const int k = pair(a, b);
      const double gw = std::pow(1.0 / gamW[k], pvdW1);
      for (int i = 1; i <= NTABLE; ++i) {
        const double dr2 = UDR * i, dr1 = std::sqrt(dr2);
        const double dr3 = dr1 * dr2, dr4 = dr2 * dr2, dr5 = dr1 * dr2 * dr2, dr6 = dr2 * dr2 * dr2, dr7 = dr1 * dr2 * dr2 * dr2;
        const double rv = std::pow(dr2, pvdW1h);
        const double Tap = CTap[7] * dr7 + CTap[6] * dr6 + CTap[5] * dr5 + CTap[4] * dr4 + CTap[0];
        const double fn13 = std::pow(rv + gw, pvdW1inv);
        const double e1 = std::exp(alpij[k] * (1.0 - fn13 / rvdW[k])), e2 = std::sqrt(e1);
        const double g3 = std::pow(dr3 + gamij[k], -1.0 / 3.0);
        const size_t o = r * stride + i;
        tblEvdw[o] = Tap * Dij[k] * (e1 - 2.0 * e2);
        tblEclmb[o] = Tap * Cclmb0 * g3;
        tblQEq[o] = Tap * Cclmb0_qeq * g3;
        const double dTap = 7.0 * CTap[7] * dr5 + 6.0 * CTap[6] * dr4 + 5.0 * CTap[5] * dr3 + 4.0 * CTap[4] * dr2;
        const double dfn13 = std::pow(rv + gw, pvdW1inv - 1.0) * std::pow(dr2, pvdW1h - 1.0);
        tbldEvdw[o] = Dij[k] * (dTap * (e1 - 2.0 * e2) - Tap * (alpij[k] / rvdW[k]) * (e1 - e2) * dfn13);
        tbldEclmb[o] = Cclmb0 * g3 * (dTap - (g3 * g3 * g3) * Tap * dr1);
        if (lg && a <= 4 && b <= 4) {                      // init.F90:496-514: low-gradient dispersion + core repulsion, C H O N only (:499)
          const double dr_lg = 2 * std::sqrt(atom[a].Re_lg * atom[b].Re_lg);
          const double d2 = dr_lg * dr_lg, dr6_lg = d2 * (d2 * d2);
          const double Elg = -C_lg[k] / (dr6 + dr6_lg);
          const double Ecore = ecore[k] * std::exp(acore[k] * (1.0 - (dr1 / rcore[k])));
          const double dElg = C_lg[k] * (6.0 * dr5) / ((dr6 + dr6_lg) * (dr6 + dr6_lg)) / dr1;
          const double dEcore = -acore[k] * Ecore / rcore[k] / dr1;
          tblEvdw[o] = tblEvdw[o] + Tap * (Elg + Ecore);
          tbldEvdw[o] = tbldEvdw[o] + dTap * Elg + Tap * dElg + dTap * Ecore + Tap * dEcore;
        }
      }
    }
}

// ---- PQEq -------------------------------------------------------------------------------------------------------
// parameter file (reference src/cmdline.F90:160-235): '#' comment lines, "NPARMS n", then n lines
//   name  flag  X0  J0  Z  Rc  Rs  Ks      (the flag column is read and ignored there: every listed type is polarizable, :217)
void ForceField::parse_pqeq(const std::string &path) {
  std::ifstream f(path);
  if (!f) throw std::runtime_error("cannot open PQEq parameter file " + path);
  std::string line;
  int n = 0;
  npq = 0;
  while (std::getline(f, line)) {
    size_t p0 = line.find_first_not_of(" \t");
    if (p0 == std::string::npos || line[p0] == '#') continue;
    const size_t pn = line.find("NPARMS");
    if (pn != std::string::npos) {
      npq = std::atoi(line.c_str() + pn + 6);
      if (npq < 1) throw std::runtime_error("PQEq: bad NPARMS line");
      X0pq.assign(npq + 1, 0.0); J0pq = Zpq = Rcpq = Rspq = Kspq = X0pq;
      continue;
    }
    if (!npq || n >= npq) continue;
    std::istringstream is(line);
    std::string nm; int flag; double x0, j0, z, rc, rs, ks;
    if (!(is >> nm >> flag >> x0 >> j0 >> z >> rc >> rs >> ks)) continue;
    ++n;
    X0pq[n] = x0; J0pq[n] = j0; Zpq[n] = z; Rcpq[n] = rc; Rspq[n] = rs; Kspq[n] = ks;
  }
  if (npq < 1 || n != npq) throw std::runtime_error("PQEq: parameter file " + path + " is incomplete");
  // initialize_pqeq, module.F90:501-522: chi <- X0, eta <- 2*J0 for every listed (polarizable) type.  The file may list fewer types
  // than the ffield (the reference's examples/3-reaxpq+ lists C and H only): the remaining types keep their ffield values, their
  // eta takes the second doubling of module.F90:522, and atoms of such a type are refused when the atoms are set (the reference
  // would index its parameter arrays out of bounds).
  for (int t = 1; t <= nso && t <= npq; ++t) { atom[t].chi = X0pq[t]; atom[t].eta = 2.0 * J0pq[t]; }
  for (int t = npq + 1; t <= nso; ++t) atom[t].eta *= 2.0;
  inxnpq.assign((npq + 1) * (npq + 1), 0);
  int c = 0;
  for (int a = 1; a <= npq; ++a) for (int b = a; b <= npq; ++b) { ++c; inxnpq[a * (npq + 1) + b] = c; inxnpq[b * (npq + 1) + a] = c; }
  pqeq = true;
}

void ForceField::build_pqeq_tables() {
  const double lambda = 0.462770;                                  // module.F90:298
  const double sqrtpi_inv = 1.0 / std::sqrt(3.14159265358979);    // module.F90:90-91
  const size_t stride = NTABLE + 2;
  const int nrow = npq * (npq + 1) / 2;
  tblPcc.assign((nrow + 1) * stride * 2, 0.0); tblPsc = tblPss = tblPcc;
  UDR = rctap2 / NTABLE; UDRi = 1.0 / UDR;
  for (int a = 1; a <= npq; ++a)
    for (int b = a; b <= npq; ++b) {
      const double aci = 0.5 * lambda / (Rcpq[a] * Rcpq[a]), asi = 0.5 * lambda / (Rspq[a] * Rspq[a]);
      const double acj = 0.5 * lambda / (Rcpq[b] * Rcpq[b]), asj = 0.5 * lambda / (Rspq[b] * Rspq[b]);
      // set_alphaij_pqeq, module.F90:448-485; the tables are filled from alpha(ity,jty) with ity <= jty (:541-547)
      const double A[3] = {std::sqrt((aci * acj) / (aci + acj)), std::sqrt((asi * acj) / (asi + acj)), std::sqrt((asi * asj) / (asi + asj))};
      std::vector<double> *T[3] = {&tblPcc, &tblPsc, &tblPss};
      const int row = ipq(a, b);
      for (int i = 1; i <= NTABLE; ++i) {
        const double dr2 = UDR * i, dr1 = std::sqrt(dr2);
        const double dr3 = dr1 * dr2, dr4 = dr2 * dr2, dr5 = dr1 * dr2 * dr2, dr6 = dr2 * dr2 * dr2, dr7 = dr1 * dr2 * dr2 * dr2;
        const double Tap = CTap[7] * dr7 + CTap[6] * dr6 + CTap[5] * dr5 + CTap[4] * dr4 + CTap[0];
        const double dTap = 7.0 * CTap[7] * dr5 + 6.0 * CTap[6] * dr4 + 5.0 * CTap[5] * dr3 + 4.0 * CTap[4] * dr2;
        const double dr1i = 1.0 / dr1, clmb = dr1i, dclmb = -dr1i * dr1i * dr1i;
        for (int k = 0; k < 3; ++k) {
          const double screen = std::erf(A[k] * dr1), dscreen = 2.0 * A[k] * sqrtpi_inv * std::exp(-A[k] * A[k] * dr2) * dr1i;
          (*T[k])[(row * stride + i) * 2] = clmb * screen * Tap;
          (*T[k])[(row * stride + i) * 2 + 1] = dclmb * screen * Tap + clmb * dscreen * Tap + clmb * screen * dTap;
        }
      }
    }
}

}  // namespace rxmd
