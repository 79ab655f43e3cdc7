// ffparams.h -- host-side ReaxFF parameter set: ffield parser + derived constants + lookup tables.
// Own implementation of what the reference does in src/param.F90 (GETPARAMS), src/init.F90
// (CUTOFFLENGTH :363-418, POTENTIALTABLE :421-522, taper :36-38).  Index conventions follow the
// force-field file: atom types 1..nso, bond rows 1..nboty, etc.; slot 0 of every table is unused
// (inxn == 0 means "no such interaction", param.F90:162).
#pragma once
#include <string>
#include <vector>

namespace rxmd {

constexpr int NTABLE = 5000;  // reference src/module.F90:251

struct AtomTypeParams {
  std::string name;
  double rat, Val, mass, rvdw1, eps, gam, rapt, Vale;      // ffield atom line 1 (param.F90:103)
  double alf, vop, Valboc, povun5, chi, eta;               // line 2 (:104); eta doubled (:361)
  double vnq, plp2, bo131, bo132, bo133;                   // line 3 (:105)
  double povun2, pval3, Valval, pval5;                     // line 4 (:111)
  double nlpopt, Valangle;                                 // derived (:121-123)
  double rcore2 = 0, ecore2 = 0, acore2 = 0, Re_lg = 0;    // --lg: line 4 fields 6-8 and line 5 (:107-109)
};
struct BondTypeParams {
  double Desig, Depi, Depipi, pbe1, pbo5, v13cor, pbo6, povun1;  // bond line 1 (:166)
  double pbe2, pbo3, pbo4, bom, pbo1, pbo2, ovc;                 // bond line 2 (:167)
  double pboc3, pboc4, pboc5;                                    // (:181-190)
  double cBOp1, cBOp3, cBOp5, pbo2h, pbo4h, pbo6h, sw[3];        // (:226-261)
  double rc, rc2;                                                // CUTOFFLENGTH
};
struct AngleTypeParams { double theta00, pval1, pval2, pcoa1, pval7, ppen1, pval4; };
struct TorsTypeParams { double V1, V2, V3, ptor1, pcot1; };
struct HbTypeParams { double r0hb, phb1, phb2, phb3; };

struct ForceField {
  std::string header;
  std::vector<double> vpar;           // 1-based general parameters
  int nso = 0, nboty = 0, nvaty = 0, ntoty = 0, nhbty = 0;
  std::vector<AtomTypeParams> atom;   // [0..nso]
  std::vector<BondTypeParams> bond;   // [0..nboty]
  std::vector<AngleTypeParams> angle; // [0..nvaty]
  std::vector<TorsTypeParams> tors;   // [0..ntoty]
  std::vector<HbTypeParams> hb;       // [0..nhbty]
  // pair tables [(nso+1)^2]
  std::vector<double> r0s, r0p, r0pp, rvdW, Dij, alpij, gamW, gamij;
  std::vector<int> inxn2, inxn3, inxn3hb, inxn4;   // (nso+1)^k dense lookups
  // scalars derived from vpar (param.F90:51-56, 90-95, 174-179, 280-291, 324-327)
  double pvdW1 = 0, vpar30 = 0, vpar1 = 0, vpar2 = 0;
  double plp1 = 0, povun3 = 0, povun4 = 0, povun6 = 0, povun7 = 0, povun8 = 0;
  double pval6 = 0, pval8 = 0, pval9 = 0, pval10 = 0, ppen2 = 0, ppen3 = 0, ppen4 = 0, pcoa2 = 0, pcoa3 = 0, pcoa4 = 0;
  double ptor2 = 0, ptor3 = 0, ptor4 = 0, pcot2 = 0;
  // run-time derived
  double rctap = 10.0, rctap2 = 100.0, CTap[8] = {0};
  double cutoff_vpar30 = 0, maxrc = 0, UDR = 0, UDRi = 0;
  // tables, layout [inxn][i] with i = 0..NTABLE+1 (entries 0 and NTABLE+1 are zero guards)
  std::vector<double> tblEvdw, tbldEvdw, tblEclmb, tbldEclmb, tblQEq;

  int n1() const { return nso + 1; }
  int pair(int a, int b) const { return a * n1() + b; }
  int ix2(int a, int b) const { return inxn2[pair(a, b)]; }

  // PQEq (reference src/cmdline.F90:160-235 get_pqeq_parms, src/module.F90:448-611): per type 1..npq, pair index ipq(a,b)
  bool pqeq = false;
  int npq = 0;
  std::vector<double> X0pq, J0pq, Zpq, Rcpq, Rspq, Kspq;   // [0..npq]
  std::vector<int> inxnpq;                                  // [(npq+1)^2]: 1-based pair row, symmetric
  std::vector<double> tblPcc, tblPsc, tblPss;               // [row][i][0:1]: energy kernel and (1/r) derivative, i = 0..NTABLE+1
  int ipq(int a, int b) const { return inxnpq[a * (npq + 1) + b]; }
  void parse_pqeq(const std::string &path);                 // also replaces chi / eta (module.F90:501-522); call before build_tables
  void build_pqeq_tables();                                 // needs the taper of rctap0_pqeq

  // throws std::runtime_error with a message on malformed input
  // lg: the low-gradient format (--lg, cmdline.F90:148-151): five-line atom blocks and a C_lg column on the off-diagonal rows
  void parse(const std::string &path, bool lg = false);
  bool lg = false;
  std::vector<double> C_lg, rcore, ecore, acore;            // [(nso+1)^2] (param.F90:83-86,140-145,197-200)
  // bond-order cutoffs per bond row; types with zero atoms are ignored for maxrc (init.F90:404-413)
  void compute_cutoffs(const std::vector<long long> &natoms_per_type);
  void build_taper(double rc);
  void build_tables();
};

}  // namespace rxmd
