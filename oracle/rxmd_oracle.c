/* rxmd_oracle.c -- TEST INFRASTRUCTURE.  NOT part of the product; never linked, imported or
 * executed by rxmd_amd/ (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg).
 *
 * A plain-C (double precision, sequential summation order) restatement of the USCCACS/RXMD
 * per-step hot path: ffield parsing + derived tables, ghost-atom copy, cell lists, bonded and
 * 10 A neighbour lists, two-vector QEq conjugate gradient, bond orders, all bonded/nonbonded
 * energy+force terms and the velocity-Verlet step.  Every function cites the reference
 * file:line (relative to /root/reference) whose behaviour it restates.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this file against fixtures
 * produced by the real reference (oracle/_ref/rxmd, built by oracle/Makefile from the
 * unmodified Fortran sources) -- per-atom forces/charges (f20.12 dumps), QEq iteration counts
 * and per-iteration energy trace, hessian row sums, MDstep energies.
 *
 * Semantics kept on purpose:  real(4) CG step lengths, real(4) r^2 in the
 * QEq list build, the index-ordered ccbnd rule of ForceBondedTerms, hydrogen == type 2 in Ehb.
 *
 * Multi-rank: a "world" holds vprocs(1)*vprocs(2)*vprocs(3) rank states in ONE process and
 * executes the reference's 6-stage exchange bulk-synchronously (message = memcpy), so that
 * MPI-decomposed runs of the reference can be restated without MPI.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <ctype.h>

#define MAXNEIGHBS 30          /* module.F90:81 */
#define NTABLE 5000            /* module.F90:251 */
#define NMINCELL 4             /* module.F90:84 */
#define MAXLAYERS 5            /* module.F90:44 */
#define MAXLAYERS_NB 10        /* module.F90:45 */
#define MODE_COPY 1            /* module.F90:38-39 */
#define MODE_MOVE 2
#define MODE_CPBK 3
#define MODE_QCOPY1 4
#define MODE_QCOPY2 5

static const double MINBOSIG = 1e-3, MINBO0 = 1e-4, cutof2_esub = 1e-4, cutof2_bo = 1e-3; /* module.F90:60-64 */
static const int is_idEh = 1;                                                             /* module.F90:65 */
static const double MAXANGLE = 0.999999999999, MINANGLE = -0.999999999999, NSMALL = 1e-10; /* :85-87 */
static const double PI_ = 3.14159265358979;                                               /* :90 */
static const double UTIME = 1e3 / 20.455;                                                 /* :202 */
static const double Cclmb0 = 332.0638, Cclmb0_qeq = 14.4, CEchrge = 23.02;                /* :681-683 */
static const double rchb2 = 100.0;                                                        /* :677-678 */

/* ------------------------------------------------------------------ parameters (module parameters) */
typedef struct {
  int nso, nboty, nvaty, ntoty, nhbty, npar;
  char name[16][4];
  double *vpar;
  /* per type (1-based) */
  double *rat, *rapt, *vnq, *Val, *Valboc, *mass, *Vale, *plp1, *nlpopt, *plp2;
  double *povun2, *povun3, *povun4, *povun5, *povun6, *povun7, *povun8;
  double *pval3, *pval5, *Valangle, *Valval, *chi, *eta, *gam;
  double *rvdw1, *eps, *alf, *vop, *bo131, *bo132, *bo133;
  /* pair (type x type) */
  double *r0s, *r0p, *r0pp, *rvdW, *Dij, *alpij, *gamW, *gamij;
  int *inxn2, *inxn3, *inxn3hb, *inxn4;
  /* per bond type */
  double *pbo1, *pbo2, *pbo3, *pbo4, *pbo5, *pbo6, *bom, *pboc1, *pboc2, *pboc3, *pboc4, *pboc5;
  double *Desig, *Depi, *Depipi, *pbe1, *pbe2, *povun1, *ovc, *v13cor;
  double *cBOp1, *cBOp3, *cBOp5, *pbo2h, *pbo4h, *pbo6h, *swtch; /* swtch[3*(inxn)+k] */
  /* angle types */
  double *pval1, *pval2, *pval4, *pval6, *pval7, *pval8, *pval9, *pval10, *theta00;
  double *ppen1, *ppen2, *ppen3, *ppen4, *pcoa1, *pcoa2, *pcoa3, *pcoa4;
  /* torsion types */
  double *ptor1, *ptor2, *ptor3, *ptor4, *V1, *V2, *V3, *pcot1, *pcot2;
  /* hbond types */
  double *phb1, *phb2, *phb3, *r0hb;
  double pvdW1, pvdW1h, pvdW1inv, vpar30, vpar1, vpar2;
  /* derived, init.F90 */
  double *rc, *rc2, maxrc, cutoff_vpar30;
  int isLG; double *C_lg, *Re_lg, *rcore2, *ecore2, *acore2, *rcore, *ecore, *acore; /* --lg, param.F90:82-86 */
  double *TBL_Eclmb, *TBL_Evdw, *TBL_Eclmb_QEq; /* [inxn][i][0:1], [inxn][i] ; i in 0..NTABLE+1 */
  double UDR, UDRi, rctap, rctap2, CTap[8];
  /* PQEq (module.F90:286-303, cmdline.F90:160-235): per type 1..ntype_pqeq; pair arrays [(ity)*(ntype_pqeq+1)+jty] */
  int isPQEq, ntype_pqeq, *isPolarizable, *inxnpqeq;
  int isEfield, eFieldDir; double eFieldStrength;    /* --efield dir strength (cmdline.F90:131-137); used with PQEq (needs Zpqeq) */
  int pq_clean;   /* 0 = the reference's behaviour beyond the cutoff (outputs keep the previous pair's values); 1 = outputs zeroed */
  double *X0pqeq, *J0pqeq, *Zpqeq, *Rcpqeq, *Rspqeq, *Kspqeq, *alphacc, *alphasc, *alphass;
  double *TBL_Eclmb_pcc, *TBL_Eclmb_psc, *TBL_Eclmb_pss;   /* [inxn][i][0:1], i in 0..NTABLE+1 (i = 0 is outside the Fortran array) */
} Params;

#define T2(p, a, b) ((p)[(a) * (P->nso + 1) + (b)])
#define T3(p, a, b, c) ((p)[((a) * (P->nso + 1) + (b)) * (P->nso + 1) + (c)])
#define T4(p, a, b, c, d) ((p)[(((a) * (P->nso + 1) + (b)) * (P->nso + 1) + (c)) * (P->nso + 1) + (d)])

static double *dalloc(size_t n) { double *p = (double *)calloc(n ? n : 1, sizeof(double)); if (!p) { fprintf(stderr, "oracle: out of memory\n"); exit(2); } return p; }
static int *ialloc(size_t n) { int *p = (int *)calloc(n ? n : 1, sizeof(int)); if (!p) { fprintf(stderr, "oracle: out of memory\n"); exit(2); } return p; }

/* Fortran fixed-width numeric field (fw.d edit descriptor): blanks -> 0, no '.' -> implied decimals */
static double ffield_f(const char *line, int col0, int w, int d) {
  char buf[64]; int n = (int)strlen(line), k = 0, hasdot = 0, hasdig = 0;
  for (int c = col0; c < col0 + w && c < n; c++) {
    char ch = line[c];
    if (ch == '\n' || ch == '\r') break;
    if (ch == ' ') continue;
    if (ch == '.') hasdot = 1;
    if (isdigit((unsigned char)ch)) hasdig = 1;
    if (ch == 'd' || ch == 'D') ch = 'e';
    buf[k++] = ch;
  }
  buf[k] = 0;
  if (!hasdig) return 0.0;
  double v = strtod(buf, NULL);
  if (!hasdot && !strchr(buf, 'e')) v /= pow(10.0, d);
  return v;
}
static int ffield_i(const char *line, int col0, int w) {
  char buf[32]; int n = (int)strlen(line), k = 0;
  for (int c = col0; c < col0 + w && c < n; c++) { char ch = line[c]; if (ch == '\n' || ch == '\r') break; if (ch != ' ') buf[k++] = ch; }
  buf[k] = 0;
  return k ? atoi(buf) : 0;
}
static char *rdline(FILE *fp, char *buf, int n) { if (!fgets(buf, n, fp)) buf[0] = 0; return buf; }

/* restates GETPARAMS, src/param.F90:2-375 */
static int read_ffield(Params *P, const char *path) {
  FILE *fp = fopen(path, "r");
  char L[512];
  if (!fp) return -1;
  rdline(fp, L, 512);                               /* header, param.F90:40 */
  rdline(fp, L, 512); P->npar = atoi(L);            /* :42 list-directed */
  P->vpar = dalloc(P->npar + 40);
  for (int i = 1; i <= P->npar; i++) { rdline(fp, L, 512); P->vpar[i] = ffield_f(L, 0, 10, 4); } /* :46-48 fmt 1300 */
  P->pvdW1 = P->vpar[29]; P->pvdW1h = 0.5 * P->pvdW1; P->pvdW1inv = 1.0 / P->pvdW1; /* :51-53 */
  P->vpar30 = P->vpar[30];                           /* :56 */
  rdline(fp, L, 512); P->nso = ffield_i(L, 0, 3);   /* :59 */
  int nso = P->nso, n1 = nso + 1;
#define A1(x) P->x = dalloc(n1)
  A1(rat); A1(rapt); A1(vnq); A1(Val); A1(Valboc); A1(mass); A1(Vale); A1(plp1); A1(nlpopt); A1(plp2);
  A1(povun2); A1(povun3); A1(povun4); A1(povun5); A1(povun6); A1(povun7); A1(povun8);
  A1(pval3); A1(pval5); A1(Valangle); A1(Valval); A1(chi); A1(eta); A1(gam);
  A1(rvdw1); A1(eps); A1(alf); A1(vop); A1(bo131); A1(bo132); A1(bo133);
#undef A1
#define A2(x) P->x = dalloc(n1 * n1)
  A2(r0s); A2(r0p); A2(r0pp); A2(rvdW); A2(Dij); A2(alpij); A2(gamW); A2(gamij);
  if (P->isLG) { A2(C_lg); A2(rcore); A2(ecore); A2(acore); P->Re_lg = dalloc(n1); P->rcore2 = dalloc(n1); P->ecore2 = dalloc(n1); P->acore2 = dalloc(n1); }
  /* the reference leaves the C_lg pairs no off-diagonal row names unset (allocate without a fill, :83); zero here */
#undef A2
  P->inxn2 = ialloc(n1 * n1); P->inxn3 = ialloc(n1 * n1 * n1); P->inxn3hb = ialloc(n1 * n1 * n1); P->inxn4 = ialloc(n1 * n1 * n1 * n1);
  for (int i = 1; i <= nso; i++) {                   /* :90-95 */
    P->plp1[i] = P->vpar[16]; P->povun3[i] = P->vpar[33]; P->povun4[i] = P->vpar[32];
    P->povun6[i] = P->vpar[7]; P->povun7[i] = P->vpar[9]; P->povun8[i] = P->vpar[10];
  }
  rdline(fp, L, 512); rdline(fp, L, 512); rdline(fp, L, 512); /* :98-100 */
  for (int i = 1; i <= nso; i++) {                   /* :102-114, formats 1200/1250 */
    rdline(fp, L, 512);
    { int k = 0; for (int c = 1; c <= 2 && L[c] && L[c] != '\n'; c++) if (L[c] != ' ') P->name[i][k++] = L[c]; P->name[i][k] = 0; }
    P->rat[i] = ffield_f(L, 3, 9, 4); P->Val[i] = ffield_f(L, 12, 9, 4); P->mass[i] = ffield_f(L, 21, 9, 4);
    P->rvdw1[i] = ffield_f(L, 30, 9, 4); P->eps[i] = ffield_f(L, 39, 9, 4); P->gam[i] = ffield_f(L, 48, 9, 4);
    P->rapt[i] = ffield_f(L, 57, 9, 4); P->Vale[i] = ffield_f(L, 66, 9, 4);
    rdline(fp, L, 512);
    P->alf[i] = ffield_f(L, 3, 9, 4); P->vop[i] = ffield_f(L, 12, 9, 4); P->Valboc[i] = ffield_f(L, 21, 9, 4);
    P->povun5[i] = ffield_f(L, 30, 9, 4); P->chi[i] = ffield_f(L, 48, 9, 4); P->eta[i] = ffield_f(L, 57, 9, 4);
    rdline(fp, L, 512);
    P->vnq[i] = ffield_f(L, 3, 9, 4); P->plp2[i] = ffield_f(L, 12, 9, 4);
    P->bo131[i] = ffield_f(L, 30, 9, 4); P->bo132[i] = ffield_f(L, 39, 9, 4); P->bo133[i] = ffield_f(L, 48, 9, 4);
    rdline(fp, L, 512);
    P->povun2[i] = ffield_f(L, 3, 9, 4); P->pval3[i] = ffield_f(L, 12, 9, 4);
    P->Valval[i] = ffield_f(L, 30, 9, 4); P->pval5[i] = ffield_f(L, 39, 9, 4);
    if (P->isLG) {                                   /* :107-109: three more fields on line 4 and a fifth line */
      P->rcore2[i] = ffield_f(L, 48, 9, 4); P->ecore2[i] = ffield_f(L, 57, 9, 4); P->acore2[i] = ffield_f(L, 66, 9, 4);
      rdline(fp, L, 512);
      T2(P->C_lg, i, i) = ffield_f(L, 3, 9, 4); P->Re_lg[i] = ffield_f(L, 12, 9, 4);
    }
  }
  for (int i = 1; i <= nso; i++) if (P->mass[i] < 21.0 && P->Valboc[i] != P->Valval[i]) P->Valboc[i] = P->Valval[i]; /* :117-119 */
  for (int i = 1; i <= nso; i++) { P->nlpopt[i] = 0.5 * (P->Vale[i] - P->Val[i]); P->Valangle[i] = P->Valboc[i]; } /* :121-123 */
  for (int i = 1; i <= nso; i++) for (int j = 1; j <= nso; j++) {  /* :126-148 */
    T2(P->r0s, i, j) = 0.5 * (P->rat[i] + P->rat[j]);
    T2(P->r0p, i, j) = 0.5 * (P->rapt[i] + P->rapt[j]);
    T2(P->r0pp, i, j) = 0.5 * (P->vnq[i] + P->vnq[j]);
    T2(P->rvdW, i, j) = sqrt(4.0 * P->rvdw1[i] * P->rvdw1[j]);
    T2(P->Dij, i, j) = sqrt(P->eps[i] * P->eps[j]);
    T2(P->alpij, i, j) = sqrt(P->alf[i] * P->alf[j]);
    T2(P->gamW, i, j) = sqrt(P->vop[i] * P->vop[j]);
    T2(P->gamij, i, j) = pow(P->gam[i] * P->gam[j], -1.5);
    if (P->isLG) {                                   /* :140-145 */
      T2(P->rcore, i, j) = sqrt(P->rcore2[i] * P->rcore2[j]);
      T2(P->ecore, i, j) = sqrt(P->ecore2[i] * P->ecore2[j]);
      T2(P->acore, i, j) = sqrt(P->acore2[i] * P->acore2[j]);
    }
  }
  rdline(fp, L, 512); P->nboty = ffield_i(L, 0, 3);  /* :151 */
  int nb1 = P->nboty + 1;
#define AB(x) P->x = dalloc(nb1)
  AB(pbo1); AB(pbo2); AB(pbo3); AB(pbo4); AB(pbo5); AB(pbo6); AB(bom); AB(pboc1); AB(pboc2); AB(pboc3); AB(pboc4); AB(pboc5);
  AB(Desig); AB(Depi); AB(Depipi); AB(pbe1); AB(pbe2); AB(povun1); AB(ovc); AB(v13cor);
  AB(cBOp1); AB(cBOp3); AB(cBOp5); AB(pbo2h); AB(pbo4h); AB(pbo6h);
#undef AB
  P->swtch = dalloc(3 * nb1 + 3);
  rdline(fp, L, 512);                                /* :160 */
  for (int ih = 1; ih <= P->nboty; ih++) {           /* :164-170 formats 1400/1450 */
    rdline(fp, L, 512);
    int ta = ffield_i(L, 0, 3), tb = ffield_i(L, 3, 3);
    P->Desig[ih] = ffield_f(L, 6, 9, 4); P->Depi[ih] = ffield_f(L, 15, 9, 4); P->Depipi[ih] = ffield_f(L, 24, 9, 4);
    P->pbe1[ih] = ffield_f(L, 33, 9, 4); P->pbo5[ih] = ffield_f(L, 42, 9, 4); P->v13cor[ih] = ffield_f(L, 51, 9, 4);
    P->pbo6[ih] = ffield_f(L, 60, 9, 4); P->povun1[ih] = ffield_f(L, 69, 9, 4);
    rdline(fp, L, 512);
    P->pbe2[ih] = ffield_f(L, 6, 9, 4); P->pbo3[ih] = ffield_f(L, 15, 9, 4); P->pbo4[ih] = ffield_f(L, 24, 9, 4);
    P->bom[ih] = ffield_f(L, 33, 9, 4); P->pbo1[ih] = ffield_f(L, 42, 9, 4); P->pbo2[ih] = ffield_f(L, 51, 9, 4);
    P->ovc[ih] = ffield_f(L, 60, 9, 4);
    T2(P->inxn2, ta, tb) = ih; T2(P->inxn2, tb, ta) = ih;
  }
  for (int ih = 1; ih <= P->nboty; ih++) { P->pboc1[ih] = P->vpar[1]; P->pboc2[ih] = P->vpar[2]; } /* :174-175 */
  P->vpar1 = P->vpar[1]; P->vpar2 = P->vpar[2];      /* :178-179 */
  for (int i = 1; i <= nso; i++) for (int j = 1; j <= nso; j++) { /* :181-190 */
    int x = T2(P->inxn2, i, j);
    if (x) { P->pboc3[x] = sqrt(P->bo132[i] * P->bo132[j]); P->pboc4[x] = sqrt(P->bo131[i] * P->bo131[j]); P->pboc5[x] = sqrt(P->bo133[i] * P->bo133[j]); }
  }
  rdline(fp, L, 512); int nodmty = ffield_i(L, 0, 3); /* :194 */
  for (int k = 0; k < nodmty; k++) {                  /* :195-217 */
    rdline(fp, L, 512);
    int a = ffield_i(L, 0, 3), b = ffield_i(L, 3, 3);
    double deodmh = ffield_f(L, 6, 9, 4), rodmh = ffield_f(L, 15, 9, 4), godmh = ffield_f(L, 24, 9, 4);
    double rsig = ffield_f(L, 33, 9, 4), rpi = ffield_f(L, 42, 9, 4), rpi2 = ffield_f(L, 51, 9, 4);
    if (P->isLG) { double c = ffield_f(L, 60, 9, 4); T2(P->C_lg, a, b) = c; T2(P->C_lg, b, a) = c; } /* :197-200 */
    if (rsig > 0) { T2(P->r0s, a, b) = rsig; T2(P->r0s, b, a) = rsig; }
    if (rpi > 0) { T2(P->r0p, a, b) = rpi; T2(P->r0p, b, a) = rpi; }
    if (rpi2 > 0) { T2(P->r0pp, a, b) = rpi2; T2(P->r0pp, b, a) = rpi2; }
    if (rodmh > 0) { T2(P->rvdW, a, b) = 2.0 * rodmh; T2(P->rvdW, b, a) = 2.0 * rodmh; }
    if (deodmh > 0) { T2(P->Dij, a, b) = deodmh; T2(P->Dij, b, a) = deodmh; }
    if (godmh > 0) { T2(P->alpij, a, b) = godmh; T2(P->alpij, b, a) = godmh; }
  }
  for (int i = 1; i <= nso; i++) for (int j = 1; j <= nso; j++) { /* :226-261 */
    int x = T2(P->inxn2, i, j);
    if (!x) continue;
    if (P->rat[i] > 0 && P->rat[j] > 0) P->swtch[3 * x + 0] = 1;
    if (P->rapt[i] > 0 && P->rapt[j] > 0) P->swtch[3 * x + 1] = 1;
    if (P->vnq[i] > 0 && P->vnq[j] > 0) P->swtch[3 * x + 2] = 1;
    P->cBOp1[x] = (T2(P->r0s, i, j) <= 0) ? 0.0 : P->pbo1[x] / pow(T2(P->r0s, i, j), P->pbo2[x]);
    P->cBOp3[x] = (T2(P->r0p, i, j) <= 0) ? 0.0 : P->pbo3[x] / pow(T2(P->r0p, i, j), P->pbo4[x]);
    P->cBOp5[x] = (T2(P->r0pp, i, j) <= 0) ? 0.0 : P->pbo5[x] / pow(T2(P->r0pp, i, j), P->pbo6[x]);
    P->pbo2h[x] = 0.5 * P->pbo2[x]; P->pbo4h[x] = 0.5 * P->pbo4[x]; P->pbo6h[x] = 0.5 * P->pbo6[x];
  }
  rdline(fp, L, 512); P->nvaty = ffield_i(L, 0, 3);  /* :265 */
  int nv1 = P->nvaty + 1;
#define AV(x) P->x = dalloc(nv1)
  AV(pval1); AV(pval2); AV(pval4); AV(pval6); AV(pval7); AV(pval8); AV(pval9); AV(pval10); AV(theta00);
  AV(ppen1); AV(ppen2); AV(ppen3); AV(ppen4); AV(pcoa1); AV(pcoa2); AV(pcoa3); AV(pcoa4);
#undef AV
  for (int i = 1; i <= P->nvaty; i++) {              /* :273-277 fmt 1500 */
    rdline(fp, L, 512);
    int a = ffield_i(L, 0, 3), b = ffield_i(L, 3, 3), c = ffield_i(L, 6, 3);
    P->theta00[i] = ffield_f(L, 9, 9, 4); P->pval1[i] = ffield_f(L, 18, 9, 4); P->pval2[i] = ffield_f(L, 27, 9, 4);
    P->pcoa1[i] = ffield_f(L, 36, 9, 4); P->pval7[i] = ffield_f(L, 45, 9, 4); P->ppen1[i] = ffield_f(L, 54, 9, 4);
    P->pval4[i] = ffield_f(L, 63, 9, 4);
    T3(P->inxn3, a, b, c) = i; T3(P->inxn3, c, b, a) = i;
  }
  for (int i = 1; i <= P->nvaty; i++) {              /* :280-293 */
    P->pval6[i] = P->vpar[15]; P->pval8[i] = P->vpar[34]; P->pval9[i] = P->vpar[17]; P->pval10[i] = P->vpar[18];
    P->ppen2[i] = P->vpar[20]; P->ppen3[i] = P->vpar[21]; P->ppen4[i] = P->vpar[22];
    P->pcoa2[i] = P->vpar[3]; P->pcoa3[i] = P->vpar[39]; P->pcoa4[i] = P->vpar[31];
    P->theta00[i] = (PI_ / 180.0) * P->theta00[i];
  }
  rdline(fp, L, 512); P->ntoty = ffield_i(L, 0, 3);  /* :296 */
  int nt1 = P->ntoty + 1;
#define AT(x) P->x = dalloc(nt1)
  AT(ptor1); AT(ptor2); AT(ptor3); AT(ptor4); AT(V1); AT(V2); AT(V3); AT(pcot1); AT(pcot2);
#undef AT
  for (int i = 1; i <= P->ntoty; i++) {              /* :301-321 fmt 1600 */
    rdline(fp, L, 512);
    int i1 = ffield_i(L, 0, 3), i2 = ffield_i(L, 3, 3), i3 = ffield_i(L, 6, 3), i4 = ffield_i(L, 9, 3);
    P->V1[i] = ffield_f(L, 12, 9, 4); P->V2[i] = ffield_f(L, 21, 9, 4); P->V3[i] = ffield_f(L, 30, 9, 4);
    P->ptor1[i] = ffield_f(L, 39, 9, 4); P->pcot1[i] = ffield_f(L, 48, 9, 4);
    if (i1 == 0) {
      for (int a = 1; a <= nso; a++) for (int d = 1; d <= nso; d++)
        if (T4(P->inxn4, a, i2, i3, d) == 0 && T4(P->inxn4, a, i3, i2, d) == 0) {
          T4(P->inxn4, a, i2, i3, d) = i; T4(P->inxn4, d, i2, i3, a) = i;
          T4(P->inxn4, a, i3, i2, d) = i; T4(P->inxn4, d, i3, i2, a) = i;
        }
    } else {
      T4(P->inxn4, i1, i2, i3, i4) = i; T4(P->inxn4, i4, i2, i3, i1) = i;
      T4(P->inxn4, i1, i3, i2, i4) = i; T4(P->inxn4, i4, i3, i2, i1) = i;
    }
  }
  for (int i = 1; i <= P->ntoty; i++) { P->ptor2[i] = P->vpar[24]; P->ptor3[i] = P->vpar[25]; P->ptor4[i] = P->vpar[26]; P->pcot2[i] = P->vpar[28]; } /* :324-327 */
  rdline(fp, L, 512); P->nhbty = ffield_i(L, 0, 3);  /* :331 */
  int nh1 = P->nhbty + 1;
  P->phb1 = dalloc(nh1); P->phb2 = dalloc(nh1); P->phb3 = dalloc(nh1); P->r0hb = dalloc(nh1);
  for (int i = 1; i <= P->nhbty; i++) {              /* :334-337 */
    rdline(fp, L, 512);
    int a = ffield_i(L, 0, 3), b = ffield_i(L, 3, 3), c = ffield_i(L, 6, 3);
    P->r0hb[i] = ffield_f(L, 9, 9, 4); P->phb1[i] = ffield_f(L, 18, 9, 4); P->phb2[i] = ffield_f(L, 27, 9, 4); P->phb3[i] = ffield_f(L, 36, 9, 4);
    T3(P->inxn3hb, a, b, c) = i;
  }
  fclose(fp);
  for (int i = 1; i <= nso; i++) P->eta[i] *= 2.0;   /* :361 */
  return 0;
}

/* restates CUTOFFLENGTH, src/init.F90:363-418 */
static void cutofflength(Params *P, const long long *natoms_per_type) {
  P->cutoff_vpar30 = cutof2_bo * P->vpar30;
  P->rc = dalloc(P->nboty + 1); P->rc2 = dalloc(P->nboty + 1);
  for (int i = 1; i <= P->nso; i++) for (int j = i; j <= P->nso; j++) {
    int x = T2(P->inxn2, i, j);
    if (!x) continue;
    double dr = 1.0, BOsig = 1.0;
    while (BOsig > MINBOSIG) { dr = dr + 0.01; BOsig = exp(P->pbo1[x] * pow(dr / T2(P->r0s, i, j), P->pbo2[x])); }
    P->rc[x] = dr; P->rc2[x] = dr * dr;
  }
  for (int i = 1; i <= P->nso; i++) if (natoms_per_type[i] == 0)
    for (int j = 1; j <= P->nso; j++) { int x = T2(P->inxn2, i, j); if (x) P->rc[x] = 0.0; x = T2(P->inxn2, j, i); if (x) P->rc[x] = 0.0; }
  P->maxrc = 0.0;
  for (int x = 1; x <= P->nboty; x++) if (P->rc[x] > P->maxrc) P->maxrc = P->rc[x];
}

static double powi_lg(double a) { double a2 = a * a; return a2 * a2 * a2; }   /* dr_lg**6 as repeated squaring evaluates it: a^2, a^4, a^2*a^4 */
/* restates POTENTIALTABLE, src/init.F90:421-522 (the LG branch :496-514 included) */
static void potentialtable(Params *P) {
  size_t nb1 = P->nboty + 1;
  P->TBL_Eclmb = dalloc(nb1 * (NTABLE + 2) * 2); P->TBL_Evdw = dalloc(nb1 * (NTABLE + 2) * 2); P->TBL_Eclmb_QEq = dalloc(nb1 * (NTABLE + 2));
  P->UDR = P->rctap2 / NTABLE; P->UDRi = 1.0 / P->UDR;
  const double *CTap = P->CTap;
  for (int ity = 1; ity <= P->nso; ity++) for (int jty = ity; jty <= P->nso; jty++) {
    int x = T2(P->inxn2, ity, jty);
    if (!x) continue;
    for (int i = 1; i <= NTABLE; i++) {
      double dr2 = P->UDR * i, dr1 = sqrt(dr2);
      double gamWij = T2(P->gamW, ity, jty), alphaij = T2(P->alpij, ity, jty), Dij0 = T2(P->Dij, ity, jty), rvdW0 = T2(P->rvdW, ity, jty);
      double gamwinvp = pow(1.0 / gamWij, P->pvdW1);
      double dr3 = dr1 * dr2, dr4 = dr2 * dr2, dr5 = dr1 * dr2 * dr2, dr6 = dr2 * dr2 * dr2, dr7 = dr1 * dr2 * dr2 * dr2;
      double rij_vd1 = pow(dr2, P->pvdW1h);
      double Tap = CTap[7] * dr7 + CTap[6] * dr6 + CTap[5] * dr5 + CTap[4] * dr4 + CTap[0];
      double fn13 = pow(rij_vd1 + gamwinvp, P->pvdW1inv);
      double exp1 = exp(alphaij * (1.0 - fn13 / rvdW0)), exp2 = sqrt(exp1);
      double dr3gamij = pow(dr3 + T2(P->gamij, ity, jty), -1.0 / 3.0);
      size_t k = (size_t)x * (NTABLE + 2) + i;
      P->TBL_Evdw[2 * k] = Tap * Dij0 * (exp1 - 2.0 * exp2);
      P->TBL_Eclmb[2 * k] = Tap * Cclmb0 * dr3gamij;
      P->TBL_Eclmb_QEq[k] = Tap * Cclmb0_qeq * dr3gamij;
      double dTap = 7.0 * CTap[7] * dr5 + 6.0 * CTap[6] * dr4 + 5.0 * CTap[5] * dr3 + 4.0 * CTap[4] * dr2;
      double dfn13 = pow(rij_vd1 + gamwinvp, P->pvdW1inv - 1.0) * pow(dr2, P->pvdW1h - 1.0);
      P->TBL_Evdw[2 * k + 1] = Dij0 * (dTap * (exp1 - 2.0 * exp2) - Tap * (alphaij / rvdW0) * (exp1 - exp2) * dfn13);
      P->TBL_Eclmb[2 * k + 1] = Cclmb0 * dr3gamij * (dTap - (dr3gamij * dr3gamij * dr3gamij) * Tap * dr1);
      if (P->isLG && ity <= 4 && jty <= 4) {         /* :496-514: low-gradient dispersion + core repulsion, C H O N only (:499) */
        double dr_lg = 2 * sqrt(P->Re_lg[ity] * P->Re_lg[jty]);
        double dr6_lg = powi_lg(dr_lg);
        double Clg = T2(P->C_lg, ity, jty), rco = T2(P->rcore, ity, jty), eco = T2(P->ecore, ity, jty), aco = T2(P->acore, ity, jty);
        double Elg = -Clg / (dr6 + dr6_lg);
        double E_core = eco * exp(aco * (1.0 - (dr1 / rco)));
        double dElg = Clg * (6.0 * dr5) / ((dr6 + dr6_lg) * (dr6 + dr6_lg)) / dr1;
        double dE_core = -aco * E_core / rco / dr1;
        P->TBL_Evdw[2 * k] = P->TBL_Evdw[2 * k] + Tap * (Elg + E_core);
        P->TBL_Evdw[2 * k + 1] = P->TBL_Evdw[2 * k + 1] + dTap * Elg + Tap * dElg + dTap * E_core + Tap * dE_core;
      }
    }
  }
}

static double powi(double a, int b) { /* integer power as flang/compiler-rt evaluate x**n */
  double r = 1.0;
  while (1) { if (b & 1) r *= a; b /= 2; if (b == 0) break; a *= a; }
  return r;
}

/* ------------------------------------------------------------------ per-rank state (module atoms/base) */
typedef struct {
  int myid, vID[3], myparity[3], target_node[7];
  double OBOX[4];
  int NATOMS, NBUFFER, copyptr[7], maxn10;
  int *ity; long long *gid;            /* atype = type + gid*1e-13, kept split */
  double *pos, *v, *f;                 /* SoA: x[k*NBUFFER + i], i 1-based */
  double *q, *qs, *qt, *gs, *gt, *hs, *ht, *qsfp, *qsfv;
  double *spos, *fpqeq, *hshs, *hsht;  /* PQEq: shell displacement (SoA like pos), field term, H.h products */
  int *frcindx;
  int *header, *llist, *nacell, *nbheader, *nbllist, *nbnacell;
  int *nbrlist, *nbrindx;              /* [i*(MAXNEIGHBS+1)+k] */
  int *nbplist; double *hessian;       /* rows of maxn10+1 for residents */
  double *BO, *dln_BOp, *dBOp, *A0, *A1, *A2, *A3, *deltap, *delta, *nlp, *dDlp, *deltalp, *ccbnd, *cdbnd;
  double *ccused;   /* test tap only: ccbnd(i) as ForceBondedTerms consumes it (pot.F90:129-135), kept before it is zeroed (:138) */
  double PE[14], astr[6];
  double *sbuf; int ns, ne; double *rbuf; int nr; size_t sbuf_cap, rbuf_cap;
  char *commflag;
} Rank;

typedef struct {
  Params P;
  int vprocs[3], nprocs;
  double lata, latb, latc, lalpha, lbeta, lgamma, HH[3][3], HHi[3][3], MDBOX, LBOX[4];
  int cc[3], nbcc[3], nbnmesh, *nbmesh;
  double lcsize[3], nblcsize[3];
  int isQEq, NMAXQEq, nstep_qeq, maxn10; double QEq_tol, dt, Lex_fqs, Lex_w2, Lex_k;
  double *dthm, *hmas;
  long long GNATOMS;
  Rank *R;
  double *trace; int ntrace, trace_cap;      /* per-iteration (Est, Gnew1, Gnew2) of the last QEq call */
  int qeq_iters_total;
  long long pq_stale;                        /* PQEq lookups that fell outside the cutoff and left stale outputs behind (see pq_coulomb) */
  char err[256];
  int qstep, md_nstep;                           /* QEq every qstep MD steps; MD steps done (nstep of main.F90:64) */
} World;

#define POS(r, i, k) ((r)->pos[(size_t)(k) * (r)->NBUFFER + (i)])
#define VEL(r, i, k) ((r)->v[(size_t)(k) * (r)->NBUFFER + (i)])
#define FRC(r, i, k) ((r)->f[(size_t)(k) * (r)->NBUFFER + (i)])
#define SPOS(r, i, k) ((r)->spos[(size_t)(k) * (r)->NBUFFER + (i)])
#define NBR(r, i, k) ((r)->nbrlist[(size_t)(i) * (MAXNEIGHBS + 1) + (k)])
#define NBX(r, i, k) ((r)->nbrindx[(size_t)(i) * (MAXNEIGHBS + 1) + (k)])
#define SL(r, i, k) ((size_t)(i) * (MAXNEIGHBS + 1) + (k))
#define BOa(r, c, i, k) ((r)->BO[SL(r, i, k) * 4 + (c)])
#define DLN(r, c, i, k) ((r)->dln_BOp[SL(r, i, k) * 3 + (c) - 1])
#define NBP(r, i, k) ((r)->nbplist[(size_t)(i) * ((r)->maxn10 + 1) + (k)])
#define HES(r, i, k) ((r)->hessian[(size_t)(i) * ((r)->maxn10 + 1) + (k)])

/* restates GetBoxParams, src/init.F90:610-633 and matinv, src/main.F90:557-579 */
static void get_box(World *W) {
  double pi = atan(1.0) * 4.0;
  double lal = W->lalpha * pi / 180.0, lbe = W->lbeta * pi / 180.0, lga = W->lgamma * pi / 180.0;
  double la = W->lata, lb = W->latb, lc = W->latc;
  double hh1 = lc * (cos(lal) - cos(lbe) * cos(lga)) / sin(lga);
  double hh2 = lc * sqrt(1.0 - cos(lal) * cos(lal) - cos(lbe) * cos(lbe) - cos(lga) * cos(lga) + 2 * cos(lal) * cos(lbe) * cos(lga)) / sin(lga);
  double (*H)[3] = W->HH;
  H[0][0] = la; H[1][0] = 0; H[2][0] = 0;
  H[0][1] = lb * cos(lga); H[1][1] = lb * sin(lga); H[2][1] = 0;
  H[0][2] = lc * cos(lbe); H[1][2] = hh1; H[2][2] = hh2;
  double (*m1)[3] = W->HH, (*m2)[3] = W->HHi;
  m2[0][0] = m1[1][1] * m1[2][2] - m1[1][2] * m1[2][1];
  m2[0][1] = m1[0][2] * m1[2][1] - m1[0][1] * m1[2][2];
  m2[0][2] = m1[0][1] * m1[1][2] - m1[0][2] * m1[1][1];
  m2[1][0] = m1[1][2] * m1[2][0] - m1[1][0] * m1[2][2];
  m2[1][1] = m1[0][0] * m1[2][2] - m1[0][2] * m1[2][0];
  m2[1][2] = m1[0][2] * m1[1][0] - m1[0][0] * m1[1][2];
  m2[2][0] = m1[1][0] * m1[2][1] - m1[1][1] * m1[2][0];
  m2[2][1] = m1[0][1] * m1[2][0] - m1[0][0] * m1[2][1];
  m2[2][2] = m1[0][0] * m1[1][1] - m1[0][1] * m1[1][0];
  double detm = m1[0][0] * m1[1][1] * m1[2][2] + m1[0][1] * m1[1][2] * m1[2][0] + m1[0][2] * m1[1][0] * m1[2][1]
              - m1[0][2] * m1[1][1] * m1[2][0] - m1[0][1] * m1[1][0] * m1[2][2] - m1[0][0] * m1[1][2] * m1[2][1];
  for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) m2[a][b] = m2[a][b] / detm;
  W->MDBOX = H[0][0] * (H[1][1] * H[2][2] - H[2][1] * H[1][2]) + H[1][0] * (H[2][1] * H[0][2] - H[0][1] * H[2][2]) + H[2][0] * (H[0][1] * H[1][2] - H[1][1] * H[0][2]);
}

/* xu2xs_inplace / xs2xu_inplace, src/main.F90:619-681 */
static void xu2xs_inplace(const World *W, Rank *r, int nmax) {
  for (int i = 1; i <= nmax; i++) {
    double rr[3] = {POS(r, i, 0), POS(r, i, 1), POS(r, i, 2)};
    for (int a = 0; a < 3; a++) POS(r, i, a) = (W->HHi[a][0] * rr[0] + W->HHi[a][1] * rr[1] + W->HHi[a][2] * rr[2]) - r->OBOX[a + 1];
  }
}
static void xs2xu_inplace(const World *W, Rank *r, int nmax) {
  for (int i = 1; i <= nmax; i++) {
    double rr[3] = {POS(r, i, 0) + r->OBOX[1], POS(r, i, 1) + r->OBOX[2], POS(r, i, 2) + r->OBOX[3]};
    for (int a = 0; a < 3; a++) POS(r, i, a) = W->HH[a][0] * rr[0] + W->HH[a][1] * rr[1] + W->HH[a][2] * rr[2];
  }
}

/* ------------------------------------------------------------------ COPYATOMS, src/comm.F90:2-597 */
static const int dinv_[7] = {0, 2, 1, 4, 3, 6, 5}, cptridx_[7] = {0, 0, 0, 2, 2, 4, 4}, is_xyz_[7] = {0, 1, 1, 2, 2, 3, 3};

static int inBuffer(const World *W, int dflag, const double dr[3], double rr) { /* comm.F90:551-576 */
  switch (dflag) {
    case 1: return W->LBOX[1] - dr[0] < rr;
    case 2: return rr <= dr[0];
    case 3: return W->LBOX[2] - dr[1] < rr;
    case 4: return rr <= dr[1];
    case 5: return W->LBOX[3] - dr[2] < rr;
    case 6: return rr <= dr[2];
  }
  return 0;
}

static void ensure(double **buf, size_t *cap, size_t n) { if (n > *cap) { *cap = 2 * n + 64; *buf = (double *)realloc(*buf, *cap * sizeof(double)); } }

/* field lists per mode (comm.F90:118-220): positions first (shifted), then 1-d arrays */
static int mode_ne_(int imode) { /* atype travels as (type,gid): 11 and 13 instead of 10 and 12 */ return imode == MODE_COPY ? 11 : imode == MODE_MOVE ? 13 : imode == MODE_QCOPY1 ? 2 : imode == MODE_QCOPY2 ? 3 : 4; }
/* PQEq adds the (unshifted) shell displacement spos to the COPY and MOVE messages, comm.F90:122,129-131,153,165-167 */
#define mode_ne(imode) (mode_ne_(imode) + ((W->P.isPQEq && ((imode) == MODE_COPY || (imode) == MODE_MOVE)) ? 3 : 0))

static void store_atoms(World *W, Rank *r, int dflag, int imode, const double dr[3]) { /* comm.F90:273-287, 367-453 */
  int ne = mode_ne(imode);
  r->ne = ne; r->ns = 0;
  if (imode == MODE_CPBK) {
    int is = 7 - dflag;
    ensure(&r->sbuf, &r->sbuf_cap, (size_t)(r->copyptr[is] - r->copyptr[is - 1] + 1) * ne);
    for (int n = r->copyptr[is - 1] + 1; n <= r->copyptr[is]; n++) {
      r->sbuf[r->ns] = (double)r->frcindx[n];
      r->sbuf[r->ns + 1] = FRC(r, n, 0); r->sbuf[r->ns + 2] = FRC(r, n, 1); r->sbuf[r->ns + 3] = FRC(r, n, 2);
      r->ns += ne;
    }
    return;
  }
  int nscan = r->copyptr[cptridx_[dflag]];
  int ixyz = is_xyz_[dflag] - 1;
  ensure(&r->sbuf, &r->sbuf_cap, (size_t)nscan * ne + ne);
  double sft = (dflag % 2 == 1) ? -W->LBOX[is_xyz_[dflag]] : W->LBOX[is_xyz_[dflag]]; /* xshift, comm.F90:531-548 */
  for (int n = 1; n <= nscan; n++) r->commflag[n] = (char)inBuffer(W, dflag, dr, POS(r, n, ixyz));
  for (int n = 1; n <= nscan; n++) {
    if (!r->commflag[n]) continue;
    double *s = r->sbuf + r->ns;
    int o = 0;
    if (imode == MODE_COPY || imode == MODE_MOVE) {
      s[0] = POS(r, n, 0); s[1] = POS(r, n, 1); s[2] = POS(r, n, 2); s[ixyz] += sft; o = 3;
      if (imode == MODE_MOVE) { s[3] = VEL(r, n, 0); s[4] = VEL(r, n, 1); s[5] = VEL(r, n, 2); o = 6; }
      s[o++] = (double)r->ity[n]; s[o++] = (double)r->gid[n];   /* atype, split in two slots of the message */
      s[o++] = r->q[n]; s[o++] = r->qs[n]; s[o++] = r->qt[n];
      if (imode == MODE_COPY) { s[o++] = r->hs[n]; s[o++] = r->ht[n]; s[o++] = (double)n; }
      else { s[o++] = r->qsfp[n]; s[o++] = r->qsfv[n]; }
      if (W->P.isPQEq) { s[o++] = SPOS(r, n, 0); s[o++] = SPOS(r, n, 1); s[o++] = SPOS(r, n, 2); }
    } else if (imode == MODE_QCOPY1) { s[0] = r->qs[n]; s[1] = r->qt[n]; o = 2; }
    else { s[0] = r->hs[n]; s[1] = r->ht[n]; s[2] = r->q[n]; o = 3; }
    if (imode == MODE_MOVE) r->ity[n] = -1;          /* comm.F90:440 */
    r->ns += o;
  }
  r->ne = ne;
}

static int append_atoms(World *W, Rank *r, int dflag, int imode) { /* comm.F90:456-528 */
  int ne = mode_ne(imode);
  if (imode == MODE_CPBK) {
    for (int i = 0; i < r->nr / ne; i++) {
      const double *b = r->rbuf + (size_t)i * ne;
      int m = (int)lround(b[0]);
      FRC(r, m, 0) += b[1]; FRC(r, m, 1) += b[2]; FRC(r, m, 2) += b[3];
    }
    return 0;
  }
  int nrecv = r->nr / ne;
  if (r->copyptr[dflag - 1] + nrecv > r->NBUFFER - 1) return -1; /* over capacity trap, comm.F90:467-472 */
  r->copyptr[dflag] = r->copyptr[dflag - 1] + nrecv;
  for (int i = 0; i < nrecv; i++) {
    const double *b = r->rbuf + (size_t)i * ne;
    int m = r->copyptr[dflag - 1] + 1 + i, o = 0;
    if (imode == MODE_COPY || imode == MODE_MOVE) {
      POS(r, m, 0) = b[0]; POS(r, m, 1) = b[1]; POS(r, m, 2) = b[2]; o = 3;
      if (imode == MODE_MOVE) { VEL(r, m, 0) = b[3]; VEL(r, m, 1) = b[4]; VEL(r, m, 2) = b[5]; o = 6; }
      r->ity[m] = (int)lround(b[o]); r->gid[m] = (long long)llround(b[o + 1]); o += 2;
      r->q[m] = b[o++]; r->qs[m] = b[o++]; r->qt[m] = b[o++];
      if (imode == MODE_COPY) { r->hs[m] = b[o++]; r->ht[m] = b[o++]; r->frcindx[m] = (int)lround(b[o++]); }
      else { r->qsfp[m] = b[o++]; r->qsfv[m] = b[o++]; }
      if (W->P.isPQEq) { SPOS(r, m, 0) = b[o++]; SPOS(r, m, 1) = b[o++]; SPOS(r, m, 2) = b[o++]; }
    } else if (imode == MODE_QCOPY1) { r->qs[m] = b[0]; r->qt[m] = b[1]; }
    else { r->hs[m] = b[0]; r->ht[m] = b[1]; r->q[m] = b[2]; }
  }
  return 0;
}

static int COPYATOMS(World *W, int imode, const double dr[3]) { /* comm.F90:2-100 */
  int np = W->nprocs;
  for (int p = 0; p < np; p++) {                       /* initialize, comm.F90:104-229 */
    Rank *r = &W->R[p];
    r->copyptr[0] = r->NATOMS;
    if (imode == MODE_COPY) for (int a = 1; a <= r->NATOMS; a++) r->frcindx[a] = a;
    if (imode != MODE_CPBK) { int nm = r->copyptr[6] > r->NATOMS ? r->copyptr[6] : r->NATOMS; xu2xs_inplace(W, r, nm); }
  }
  for (int dflag = 1; dflag <= 6; dflag++) {
    for (int p = 0; p < np; p++) store_atoms(W, &W->R[p], dflag, imode, dr);
    for (int p = 0; p < np; p++) {                     /* send_recv: rank p receives what its tn2 sent to it */
      Rank *r = &W->R[p];
      int tn2 = (imode == MODE_CPBK) ? r->target_node[7 - dflag] : r->target_node[dinv_[dflag]];
      Rank *s = &W->R[tn2];
      ensure(&r->rbuf, &r->rbuf_cap, (size_t)s->ns + 1);
      memcpy(r->rbuf, s->sbuf, (size_t)s->ns * sizeof(double));
      r->nr = s->ns;
    }
    for (int p = 0; p < np; p++) if (append_atoms(W, &W->R[p], dflag, imode)) { snprintf(W->err, 256, "over capacity in append_atoms (NBUFFER=%d)", W->R[p].NBUFFER); return -1; }
  }
  for (int p = 0; p < np; p++) {                       /* finalize, comm.F90:232-270 */
    Rank *r = &W->R[p];
    if (imode == MODE_MOVE) {
      int ni = 0;
      for (int i = 1; i <= r->copyptr[6]; i++) if (r->ity[i] > 0) {
        ni++;
        for (int k = 0; k < 3; k++) { POS(r, ni, k) = POS(r, i, k); VEL(r, ni, k) = VEL(r, i, k); if (W->P.isPQEq) SPOS(r, ni, k) = SPOS(r, i, k); }
        r->ity[ni] = r->ity[i]; r->gid[ni] = r->gid[i]; r->q[ni] = r->q[i]; r->qs[ni] = r->qs[i]; r->qt[ni] = r->qt[i];
        r->qsfp[ni] = r->qsfp[i]; r->qsfv[ni] = r->qsfv[i];
      }
      r->NATOMS = ni;
    }
    if (imode != MODE_CPBK) xs2xu_inplace(W, r, r->copyptr[6]);
  }
  return 0;
}

/* ------------------------------------------------------------------ LINKEDLIST, src/main.F90:277-318 */
#define CIDX(c0, c1, c2, n, L) ((((size_t)((c0) + (L))) * ((n)[1] + 2 * (L)) + ((c1) + (L))) * ((n)[2] + 2 * (L)) + ((c2) + (L)))
static int LINKEDLIST(const World *W, Rank *r, const double cellDims[3], int *headAtom, int *atomList, int *NatomPerCell, const int Ncells[3], int NLAYERS) {
  size_t ncell = (size_t)(Ncells[0] + 2 * NLAYERS) * (Ncells[1] + 2 * NLAYERS) * (Ncells[2] + 2 * NLAYERS);
  for (size_t c = 0; c < ncell; c++) { headAtom[c] = -1; NatomPerCell[c] = 0; }
  for (int n = 0; n < r->NBUFFER; n++) atomList[n] = 0;
  for (int n = 1; n <= r->copyptr[6]; n++) {
    if (r->ity[n] == 0) continue;
    double rr[3] = {POS(r, n, 0), POS(r, n, 1), POS(r, n, 2)}, rn[3];
    int l[3];
    for (int a = 0; a < 3; a++) {                     /* xu2xs, main.F90:596-616 */
      rn[a] = (W->HHi[a][0] * rr[0] + W->HHi[a][1] * rr[1] + W->HHi[a][2] * rr[2]) - r->OBOX[a + 1];
      l[a] = (int)floor(rn[a] / cellDims[a]);
      if (l[a] < -NLAYERS || l[a] > Ncells[a] - 1 + NLAYERS) return -1;
    }
    size_t c = CIDX(l[0], l[1], l[2], Ncells, NLAYERS);
    atomList[n] = headAtom[c]; headAtom[c] = n; NatomPerCell[c]++;
  }
  return 0;
}

/* NEIGHBORLIST, src/main.F90:321-417 */
static int NEIGHBORLIST(World *W, Rank *r, int nlayer) {
  const Params *P = &W->P;
  const int *cc = W->cc;
  for (int i = 0; i < r->NBUFFER; i++) NBR(r, i, 0) = 0;
  for (int c1 = -nlayer; c1 <= cc[0] - 1 + nlayer; c1++) for (int c2 = -nlayer; c2 <= cc[1] - 1 + nlayer; c2++) for (int c3 = -nlayer; c3 <= cc[2] - 1 + nlayer; c3++) {
    size_t c = CIDX(c1, c2, c3, cc, MAXLAYERS);
    int m = r->header[c];
    for (int m1 = 1; m1 <= r->nacell[c]; m1++) {
      int mty = r->ity[m];
      for (int c4 = -1; c4 <= 1; c4++) for (int c5 = -1; c5 <= 1; c5++) for (int c6 = -1; c6 <= 1; c6++) {
        size_t cn = CIDX(c1 + c4, c2 + c5, c3 + c6, cc, MAXLAYERS);
        int n = r->header[cn];
        for (int nn = 1; nn <= r->nacell[cn]; nn++) {
          if (n != m) {
            int nty = r->ity[n], inxn = T2(P->inxn2, mty, nty);
            double d0 = POS(r, n, 0) - POS(r, m, 0), d1 = POS(r, n, 1) - POS(r, m, 1), d2 = POS(r, n, 2) - POS(r, m, 2);
            double dr2 = d0 * d0 + d1 * d1 + d2 * d2;
            if (inxn && dr2 < P->rc2[inxn]) {
              int k = ++NBR(r, m, 0);
              if (k > MAXNEIGHBS) { snprintf(W->err, 256, "overflow of max # in neighbor list"); return -1; }
              NBR(r, m, k) = n;
            }
          }
          n = r->llist[n];
        }
      }
      m = r->llist[m];
    }
  }
  for (int i = 1; i <= r->copyptr[6]; i++) for (int i1 = 1; i1 <= NBR(r, i, 0); i1++) { /* reverse index, :384-398 */
    int j = NBR(r, i, i1), found = 0;
    for (int j1 = 1; j1 <= NBR(r, j, 0); j1++) if (NBR(r, j, j1) == i) { NBX(r, i, i1) = j1; found = 1; }
    if (!found) { snprintf(W->err, 256, "inconsistency between nbrlist and nbrindx"); return -1; }
  }
  return 0;
}

/* the 10 A stencil walk shared by qeq_initialize (src/qeq.F90:183-268, float dr2, '<') and
 * GetNonbondingPairList (src/main.F90:420-477, double dr2, '<=') */
static int nb_pairlist(World *W, Rank *r, int with_hessian) {
  const Params *P = &W->P;
  const int *nbcc = W->nbcc;
  int fail = 0;
  if (!with_hessian) for (int i = 0; i <= r->NATOMS; i++) NBP(r, i, 0) = 0;
#pragma omp parallel for collapse(2) schedule(dynamic)
  for (int c1 = 0; c1 < nbcc[0]; c1++) for (int c2 = 0; c2 < nbcc[1]; c2++) for (int c3 = 0; c3 < nbcc[2]; c3++) {
    size_t c = CIDX(c1, c2, c3, nbcc, MAXLAYERS_NB);
    int i = r->nbheader[c];
    for (int m = 1; m <= r->nbnacell[c]; m++) {
      if (i > r->NATOMS) { i = r->nbllist[i]; continue; } /* ghosts never sit inside the domain; guard the row storage */
      int ity = r->ity[i], cnt = 0;
      NBP(r, i, 0) = 0;
      for (int mn = 0; mn < W->nbnmesh; mn++) {
        size_t cn = CIDX(c1 + W->nbmesh[3 * mn], c2 + W->nbmesh[3 * mn + 1], c3 + W->nbmesh[3 * mn + 2], nbcc, MAXLAYERS_NB);
        int j = r->nbheader[cn];
        for (int n = 1; n <= r->nbnacell[cn]; n++) {
          if (i != j) {
            double d0 = POS(r, i, 0) - POS(r, j, 0), d1 = POS(r, i, 1) - POS(r, j, 1), d2 = POS(r, i, 2) - POS(r, j, 2);
            double dr2d = d0 * d0 + d1 * d1 + d2 * d2;
            if (with_hessian) {
              float dr2 = (float)dr2d;                 /* real(4) :: dr2, qeq.F90:191,222 */
              if ((double)dr2 < P->rctap2) {
                if (cnt >= r->maxn10) { fail = 1; } else {
                  cnt++;
                  NBP(r, i, cnt) = j;
                  int itb = (int)((double)dr2 * P->UDRi);
                  double drtb = (double)dr2 - itb * P->UDR;
                  drtb = drtb * P->UDRi;
                  int inxn = T2(P->inxn2, ity, r->ity[j]);
                  const double *T = P->TBL_Eclmb_QEq + (size_t)inxn * (NTABLE + 2);
                  HES(r, i, cnt) = (1.0 - drtb) * T[itb] + drtb * T[itb + 1];
                }
              }
            } else if (dr2d <= P->rctap2) {
              if (cnt >= r->maxn10) { fail = 1; } else { cnt++; NBP(r, i, cnt) = j; }
            }
          }
          j = r->nbllist[j];
        }
      }
      NBP(r, i, 0) = cnt;
      i = r->nbllist[i];
    }
  }
  if (fail) { snprintf(W->err, 256, "nbplist greater than MAXNEIGHBS10=%d", r->maxn10); return -1; }
  return 0;
}

/* MPI_ALLREDUCE(SUM) as MPICH evaluates it for a power-of-two communicator (recursive doubling = pairwise tree);
 * for other sizes: rank order.  Only the rounding of 1e-16-level differs, but the CG exit iteration can feel it. */
static double allreduce_sum(const double *v, int n) {
  double t[64];
  if (n > 64 || (n & (n - 1))) { double s = 0; for (int i = 0; i < n; i++) s += v[i]; return s; }
  for (int i = 0; i < n; i++) t[i] = v[i];
  for (int w = 1; w < n; w <<= 1) for (int i = 0; i < n; i += 2 * w) t[i] = t[i] + t[i + w];
  return t[0];
}

/* ------------------------------------------------------------------ QEq, src/qeq.F90:2-178 */
static void get_gradient(World *W, double Gnew[2]) { /* qeq.F90:321-363 */
  const Params *P = &W->P;
  double ga[64], gb[64];
  for (int p = 0; p < W->nprocs; p++) {
    Rank *r = &W->R[p];
#pragma omp parallel for schedule(static)
    for (int i = 1; i <= r->NATOMS; i++) {
      double gssum = 0.0, gtsum = 0.0;
      for (int j1 = 1; j1 <= NBP(r, i, 0); j1++) {
        int j = NBP(r, i, j1);
        gssum = gssum + HES(r, i, j1) * r->qs[j];
        gtsum = gtsum + HES(r, i, j1) * r->qt[j];
      }
      double eta_ity = P->eta[r->ity[i]];
      r->gs[i] = -P->chi[r->ity[i]] - eta_ity * r->qs[i] - gssum;
      r->gt[i] = -1.0 - eta_ity * r->qt[i] - gtsum;
    }
    double a = 0, b = 0;
    for (int i = 1; i <= r->NATOMS; i++) { a += r->gs[i] * r->gs[i]; b += r->gt[i] * r->gt[i]; }
    ga[p] = a; gb[p] = b;                              /* MPI_ALLREDUCE, :357 */
  }
  Gnew[0] = allreduce_sum(ga, W->nprocs); Gnew[1] = allreduce_sum(gb, W->nprocs);
}

static void get_hsh(World *W, double *Est, double *hshs_sum, double *hsht_sum) { /* qeq.F90:271-318 */
  const Params *P = &W->P;
  double Ea[64], Sa[64], Ta[64];
  for (int p = 0; p < W->nprocs; p++) {
    Rank *r = &W->R[p];
    double e = 0.0, s = 0.0, t = 0.0;
    for (int i = 1; i <= r->NATOMS; i++) {
      int ity = r->ity[i];
      double eta_ity = P->eta[ity];
      double t_hshs = eta_ity * r->hs[i], t_hsht = eta_ity * r->ht[i];
      e = e + P->chi[ity] * r->q[i] + 0.5 * eta_ity * r->q[i] * r->q[i];
      for (int j1 = 1; j1 <= NBP(r, i, 0); j1++) {
        int j = NBP(r, i, j1);
        t_hshs = t_hshs + HES(r, i, j1) * r->hs[j];
        t_hsht = t_hsht + HES(r, i, j1) * r->ht[j];
        double Est1 = 0.5 * HES(r, i, j1) * r->q[i] * r->q[j];
        e = e + Est1;
        if (j <= r->NATOMS) e = e + Est1;
      }
      s = s + t_hshs * r->hs[i];
      t = t + t_hsht * r->ht[i];
    }
    Ea[p] = e; Sa[p] = s; Ta[p] = t;
  }
  *Est = allreduce_sum(Ea, W->nprocs); *hshs_sum = allreduce_sum(Sa, W->nprocs); *hsht_sum = allreduce_sum(Ta, W->nprocs);
}

static void trace_push(World *W, double a, double b, double c) {
  if (W->ntrace + 1 > W->trace_cap) { W->trace_cap = 2 * W->trace_cap + 64; W->trace = (double *)realloc(W->trace, sizeof(double) * 3 * W->trace_cap); }
  W->trace[3 * W->ntrace] = a; W->trace[3 * W->ntrace + 1] = b; W->trace[3 * W->ntrace + 2] = c; W->ntrace++;
}

static int QEq(World *W) {
  const Params *P = &W->P;
  int nmax;
  double QCopyDr[3] = {P->rctap / W->lata, P->rctap / W->latb, P->rctap / W->latc};
  for (int p = 0; p < W->nprocs; p++) {                /* qeq.F90:36-63 */
    Rank *r = &W->R[p];
    if (W->isQEq == 1) {
      for (int i = 1; i <= r->NATOMS; i++) { r->qsfp[i] = r->q[i]; r->qsfv[i] = 0.0; }
      for (int i = 0; i < r->NBUFFER; i++) { r->qs[i] = 0.0; r->qt[i] = 0.0; }
      for (int i = 1; i <= r->NATOMS; i++) r->qs[i] = r->q[i];
    } else if (W->isQEq == 2) {
      for (int i = 1; i <= r->NATOMS; i++) { r->qs[i] = W->Lex_fqs * r->qsfp[i] + (1.0 - W->Lex_fqs) * r->q[i]; r->qt[i] = 0.0; }
    }
  }
  if (W->isQEq == 1) nmax = W->NMAXQEq; else if (W->isQEq == 2) nmax = 1; else return 0;
  W->ntrace = 0;
  if (COPYATOMS(W, MODE_COPY, QCopyDr)) return -1;    /* :70 */
  for (int p = 0; p < W->nprocs; p++) {
    Rank *r = &W->R[p];
    if (LINKEDLIST(W, r, W->nblcsize, r->nbheader, r->nbllist, r->nbnacell, W->nbcc, MAXLAYERS_NB)) { snprintf(W->err, 256, "atom outside the cell grid (NB)"); return -1; }
    if (nb_pairlist(W, r, 1)) return -1;               /* qeq_initialize, :73 */
  }
  double Gnew[2], Gold[2];
  COPYATOMS(W, MODE_QCOPY1, QCopyDr);                  /* :86 */
  get_gradient(W, Gnew);
  for (int p = 0; p < W->nprocs; p++) { Rank *r = &W->R[p]; for (int i = 1; i <= r->NATOMS; i++) { r->hs[i] = r->gs[i]; r->ht[i] = r->gt[i]; } }
  COPYATOMS(W, MODE_QCOPY2, QCopyDr);                  /* :93 */
  double GEst2 = 1e99, GEst1 = 0;
  int it;
  for (it = 0; it <= nmax - 1; it++) {                 /* :96 */
    double Est, hshs_sum, hsht_sum;
    get_hsh(W, &Est, &hshs_sum, &hsht_sum);
    GEst1 = Est;
    trace_push(W, GEst1, Gnew[0], Gnew[1]);
    if (0.5 * (fabs(GEst2) + fabs(GEst1)) < W->QEq_tol) break;                       /* :114 */
    if (fabs(GEst2) > 0.0 && (fabs(GEst1 / GEst2 - 1.0) < W->QEq_tol)) break;        /* :115 */
    GEst2 = GEst1;
    double g_h[2], pa[64], pb[64];
    for (int p = 0; p < W->nprocs; p++) {              /* dot_product per rank then allreduce, :119-131 */
      Rank *r = &W->R[p];
      double a = 0, b = 0;
      for (int i = 1; i <= r->NATOMS; i++) { a += r->gs[i] * r->hs[i]; b += r->gt[i] * r->ht[i]; }
      pa[p] = a; pb[p] = b;
    }
    g_h[0] = allreduce_sum(pa, W->nprocs); g_h[1] = allreduce_sum(pb, W->nprocs);
    {  /* hshs/hsht travel in the same 4-element allreduce (qeq.F90:126-131): they were reduced once in get_hsh already in
          this restatement, which is what the reference does too (get_hsh returns local sums, the allreduce is here) */ }
    float lmin[2];                                     /* real(4) :: lmin(2), qeq.F90:23,133 */
    lmin[0] = (float)(g_h[0] / hshs_sum); lmin[1] = (float)(g_h[1] / hsht_sum);
    double ssum, tsum;
    for (int p = 0; p < W->nprocs; p++) {
      Rank *r = &W->R[p];
      double a = 0, b = 0;
      for (int i = 1; i <= r->NATOMS; i++) { r->qs[i] = r->qs[i] + (double)lmin[0] * r->hs[i]; r->qt[i] = r->qt[i] + (double)lmin[1] * r->ht[i]; }
      for (int i = 1; i <= r->NATOMS; i++) a += r->qs[i];
      for (int i = 1; i <= r->NATOMS; i++) b += r->qt[i];
      pa[p] = a; pb[p] = b;
    }
    ssum = allreduce_sum(pa, W->nprocs); tsum = allreduce_sum(pb, W->nprocs);
    double mu = ssum / tsum;                            /* :147 */
    for (int p = 0; p < W->nprocs; p++) { Rank *r = &W->R[p]; for (int i = 1; i <= r->NATOMS; i++) r->q[i] = r->qs[i] - mu * r->qt[i]; }
    COPYATOMS(W, MODE_QCOPY1, QCopyDr);                /* :153 */
    Gold[0] = Gnew[0]; Gold[1] = Gnew[1];
    get_gradient(W, Gnew);
    for (int p = 0; p < W->nprocs; p++) {
      Rank *r = &W->R[p];
      for (int i = 1; i <= r->NATOMS; i++) { r->hs[i] = r->gs[i] + (Gnew[0] / Gold[0]) * r->hs[i]; r->ht[i] = r->gt[i] + (Gnew[1] / Gold[1]) * r->ht[i]; }
    }
    COPYATOMS(W, MODE_QCOPY2, QCopyDr);                /* :164 */
  }
  W->nstep_qeq = it;                                    /* Fortran do-variable after exit / completion */
  W->qeq_iters_total += it;
  return 0;
}


/* ================================================================== PQEq, src/pqeq.F90 + module.F90:386-611 */
#define PQ2(a, i, j) ((a)[(i) * (P->ntype_pqeq + 1) + (j)])
#define PQT(T, inxn, i, d) ((T)[((size_t)(inxn) * (NTABLE + 2) + (i)) * 2 + (d)])
static const double Eev_kcal = 23.060538;                           /* module.F90:191 */
static const double lambda_pqeq = 0.462770;                         /* module.F90:298 */
static const double rctap0_pqeq = 12.5;                             /* module.F90:282 */

/* get_pqeq_parms, src/cmdline.F90:160-235: '#' lines skipped; "NPARMS n"; then per line  name flag X0 J0 Z Rc Rs Ks
 * (the flag token is read and ignored: every listed type is polarizable, :217) */
static int read_pqeq(Params *P, const char *path) {
  FILE *f = fopen(path, "r");
  if (!f) return -1;
  char line[512]; int n = 0;
  P->ntype_pqeq = 0;
  while (fgets(line, sizeof line, f)) {
    char *t = line; while (*t == ' ' || *t == '\t') t++;
    if (*t == '#' || *t == '\n' || *t == 0) continue;     /* cmdline.F90:185 tests the first column only; blank lines do not occur in the inputs */
    if (strstr(line, "NPARMS")) {
      int np = 0; sscanf(strstr(line, "NPARMS") + 6, "%d", &np);
      P->ntype_pqeq = np;
      size_t m = (size_t)np + 1;
      P->isPolarizable = ialloc(m); P->X0pqeq = dalloc(m); P->J0pqeq = dalloc(m); P->Zpqeq = dalloc(m); P->Rcpqeq = dalloc(m); P->Rspqeq = dalloc(m); P->Kspqeq = dalloc(m);
      P->alphacc = dalloc(m * m); P->alphasc = dalloc(m * m); P->alphass = dalloc(m * m); P->inxnpqeq = ialloc(m * m);
      continue;
    }
    if (!P->ntype_pqeq || n >= P->ntype_pqeq) continue;
    char nm[16]; int flag; double x0, j0, z, rc, rs, ks;
    if (sscanf(line, "%15s %d %lf %lf %lf %lf %lf %lf", nm, &flag, &x0, &j0, &z, &rc, &rs, &ks) != 8) continue;
    n++;
    P->isPolarizable[n] = 1; P->X0pqeq[n] = x0; P->J0pqeq[n] = j0; P->Zpqeq[n] = z; P->Rcpqeq[n] = rc; P->Rspqeq[n] = rs; P->Kspqeq[n] = ks;
  }
  fclose(f);
  return (P->ntype_pqeq > 0 && n == P->ntype_pqeq) ? 0 : -2;
}

/* set_alphaij_pqeq + initialize_pqeq, src/module.F90:448-611 (needs rctap, CTap) */
static void initialize_pqeq(Params *P) {
  int nt = P->ntype_pqeq;
  for (int ity = 1; ity <= nt; ity++) {
    double alpha_ci = 0.5 * lambda_pqeq / (P->Rcpqeq[ity] * P->Rcpqeq[ity]), alpha_si = 0.5 * lambda_pqeq / (P->Rspqeq[ity] * P->Rspqeq[ity]);
    for (int jty = 1; jty <= nt; jty++) {
      double alpha_cj = 0.5 * lambda_pqeq / (P->Rcpqeq[jty] * P->Rcpqeq[jty]), alpha_sj = 0.5 * lambda_pqeq / (P->Rspqeq[jty] * P->Rspqeq[jty]);
      PQ2(P->alphacc, ity, jty) = sqrt((alpha_ci * alpha_cj) / (alpha_ci + alpha_cj));
      if (P->isPolarizable[ity] && P->isPolarizable[jty]) PQ2(P->alphass, ity, jty) = sqrt((alpha_si * alpha_sj) / (alpha_si + alpha_sj));
      if (P->isPolarizable[ity]) PQ2(P->alphasc, ity, jty) = sqrt((alpha_si * alpha_cj) / (alpha_si + alpha_cj));
    }
  }
  for (int ity = 1; ity <= nt; ity++) {                 /* module.F90:501-519 */
    if (!P->isPolarizable[ity]) { P->Zpqeq[ity] = 0.0; P->Kspqeq[ity] = 0.0; }
    else if (ity <= P->nso) { P->chi[ity] = P->X0pqeq[ity]; P->eta[ity] = P->J0pqeq[ity]; }
  }
  for (int ity = 1; ity <= P->nso; ity++) P->eta[ity] = 2.0 * P->eta[ity];   /* :522, eta(:) = every ffield type: types the file does not list are doubled a second time */
  int icounter = 0;
  for (int ity = 1; ity <= nt; ity++) for (int jty = ity; jty <= nt; jty++) { icounter++; PQ2(P->inxnpqeq, ity, jty) = icounter; PQ2(P->inxnpqeq, jty, ity) = icounter; }
  size_t tsz = (size_t)(nt * nt + 1) * (NTABLE + 2) * 2;
  P->TBL_Eclmb_pcc = dalloc(tsz); P->TBL_Eclmb_psc = dalloc(tsz); P->TBL_Eclmb_pss = dalloc(tsz);
  const double sqrtpi_inv = 1.0 / sqrt(3.14159265358979);             /* module.F90:90-91 */
  double UDR = P->rctap2 / NTABLE;
  for (int ity = 1; ity <= nt; ity++) for (int jty = ity; jty <= nt; jty++) {
    double A[3] = {PQ2(P->alphacc, ity, jty), PQ2(P->alphasc, ity, jty), PQ2(P->alphass, ity, jty)};
    double *T[3] = {P->TBL_Eclmb_pcc, P->TBL_Eclmb_psc, P->TBL_Eclmb_pss};
    int inxn = PQ2(P->inxnpqeq, ity, jty);
    for (int i = 1; i <= NTABLE; i++) {
      double dr2 = UDR * i, dr1 = sqrt(dr2);
      double dr3 = dr1 * dr2, dr4 = dr2 * dr2, dr5 = dr1 * dr2 * dr2, dr6 = dr2 * dr2 * dr2, dr7 = dr1 * dr2 * dr2 * dr2;
      double Tap = P->CTap[7] * dr7 + P->CTap[6] * dr6 + P->CTap[5] * dr5 + P->CTap[4] * dr4 + P->CTap[0];
      double dTap = 7.0 * P->CTap[7] * dr5 + 6.0 * P->CTap[6] * dr4 + 5.0 * P->CTap[5] * dr3 + 4.0 * P->CTap[4] * dr2;
      double dr1i = 1.0 / dr1, clmb = dr1i, dclmb = -dr1i * dr1i * dr1i;
      for (int k = 0; k < 3; k++) {
        double screen = erf(A[k] * dr1), dscreen = 2.0 * A[k] * sqrtpi_inv * exp(-A[k] * A[k] * dr2) * dr1i;
        PQT(T[k], inxn, i, 0) = clmb * screen * Tap;
        PQT(T[k], inxn, i, 1) = dclmb * screen * Tap + clmb * dscreen * Tap + clmb * screen * dTap;
      }
    }
  }
}

/* get_coulomb_and_dcoulomb_pqeq, src/module.F90:386-445 (the live part, :401-418).  Beyond the cutoff the routine RETURNS
 * WITHOUT TOUCHING ITS OUTPUTS: callers that do not reset them see the previous pair's values (qeq_initialize,
 * update_shell_positions).  Kept: returns 0 in that case, 1 otherwise. */
static int pq_coulomb(const Params *P, const double rr[3], int inxn, const double *TBL, double *Eclmb, double ff[3]) {
  double dr2 = rr[0] * rr[0] + rr[1] * rr[1] + rr[2] * rr[2];
  if (dr2 > P->rctap2) { if (P->pq_clean) { *Eclmb = 0.0; ff[0] = ff[1] = ff[2] = 0.0; } return 0; }
  int itb = (int)(dr2 * P->UDRi), itb1 = itb + 1;
  double drtb = dr2 - itb * P->UDR; drtb = drtb * P->UDRi;
  double drtb1 = 1.0 - drtb;
  *Eclmb = drtb1 * PQT(TBL, inxn, itb, 0) + drtb * PQT(TBL, inxn, itb1, 0);
  double dEclmb = drtb1 * PQT(TBL, inxn, itb, 1) + drtb * PQT(TBL, inxn, itb1, 1);
  for (int k = 0; k < 3; k++) ff[k] = dEclmb * rr[k];
  return 1;
}

/* qeq_initialize of PQEq, src/pqeq.F90:262-353: list + core-core hessian + field term fpqeq (Eq. 30).  Serial on purpose:
 * pqeqc / pqeqs / ff are routine-scope variables that keep the previous pair's value when a lookup falls outside the cutoff. */
static int pq_initialize(World *W, Rank *r, long long *nstale) {
  const Params *P = &W->P; const int *nbcc = W->nbcc;
  double pqeqc = 0.0, pqeqs = 0.0, ff[3] = {0, 0, 0};
  for (int i = 0; i <= r->NATOMS; i++) NBP(r, i, 0) = 0;
  for (int c1 = 0; c1 < nbcc[0]; c1++) for (int c2 = 0; c2 < nbcc[1]; c2++) for (int c3 = 0; c3 < nbcc[2]; c3++) {
    size_t c = CIDX(c1, c2, c3, nbcc, MAXLAYERS_NB);
    int i = r->nbheader[c];
    for (int m = 1; m <= r->nbnacell[c]; m++) {
      if (i > r->NATOMS) { i = r->nbllist[i]; continue; }
      int ity = r->ity[i], cnt = 0;
      r->fpqeq[i] = 0.0;
      for (int mn = 0; mn < W->nbnmesh; mn++) {
        size_t cn = CIDX(c1 + W->nbmesh[3 * mn], c2 + W->nbmesh[3 * mn + 1], c3 + W->nbmesh[3 * mn + 2], nbcc, MAXLAYERS_NB);
        int j = r->nbheader[cn];
        for (int n = 1; n <= r->nbnacell[cn]; n++) {
          if (i != j) {
            double dr[3] = {POS(r, i, 0) - POS(r, j, 0), POS(r, i, 1) - POS(r, j, 1), POS(r, i, 2) - POS(r, j, 2)};
            float dr2 = (float)(dr[0] * dr[0] + dr[1] * dr[1] + dr[2] * dr[2]);      /* real(4) :: dr2, pqeq.F90:271,305 */
            if ((double)dr2 < P->rctap2) {
              int jty = r->ity[j];
              if (cnt >= r->maxn10) { snprintf(W->err, 256, "nbplist greater than MAXNEIGHBS10=%d", r->maxn10); return -1; }
              cnt++;
              NBP(r, i, cnt) = j;
              if (!pq_coulomb(P, dr, PQ2(P->inxnpqeq, ity, jty), P->TBL_Eclmb_pcc, &pqeqc, ff)) (*nstale)++;
              HES(r, i, cnt) = Cclmb0_qeq * pqeqc;
              r->fpqeq[i] = r->fpqeq[i] + Cclmb0_qeq * pqeqc * P->Zpqeq[jty];
              if (P->isPolarizable[jty]) {
                double d2[3] = {POS(r, i, 0) - POS(r, j, 0) - SPOS(r, j, 0), POS(r, i, 1) - POS(r, j, 1) - SPOS(r, j, 1), POS(r, i, 2) - POS(r, j, 2) - SPOS(r, j, 2)};
                if (!pq_coulomb(P, d2, PQ2(P->inxnpqeq, jty, ity), P->TBL_Eclmb_psc, &pqeqs, ff)) (*nstale)++;
                r->fpqeq[i] = r->fpqeq[i] - Cclmb0_qeq * pqeqs * P->Zpqeq[jty];
              }
            }
          }
          j = r->nbllist[j];
        }
      }
      NBP(r, i, 0) = cnt;
      i = r->nbllist[i];
    }
  }
  return 0;
}

static void pq_get_gradient(World *W, double Gnew[2]) { /* pqeq.F90:432-478 */
  const Params *P = &W->P;
  double ga[64], gb[64];
  for (int p = 0; p < W->nprocs; p++) {
    Rank *r = &W->R[p];
    for (int i = 1; i <= r->NATOMS; i++) {
      double gssum = 0.0, gtsum = 0.0;
      for (int j1 = 1; j1 <= NBP(r, i, 0); j1++) {
        int j = NBP(r, i, j1);
        gssum = gssum + HES(r, i, j1) * r->qs[j];
        gtsum = gtsum + HES(r, i, j1) * r->qt[j];
      }
      double eta_ity = P->eta[r->ity[i]];
      r->gs[i] = -P->chi[r->ity[i]] - eta_ity * r->qs[i] - gssum - r->fpqeq[i];
      r->gt[i] = -1.0 - eta_ity * r->qt[i] - gtsum;
    }
    double a = 0, b = 0;
    for (int i = 1; i <= r->NATOMS; i++) a += r->gs[i] * r->gs[i];
    for (int i = 1; i <= r->NATOMS; i++) b += r->gt[i] * r->gt[i];
    ga[p] = a; gb[p] = b;
  }
  Gnew[0] = allreduce_sum(ga, W->nprocs); Gnew[1] = allreduce_sum(gb, W->nprocs);
}

static double pq_get_hsh(World *W) { /* pqeq.F90:356-429 -- Est with the shell terms, no doubling for resident partners */
  const Params *P = &W->P;
  double Ea[64];
  for (int p = 0; p < W->nprocs; p++) {
    Rank *r = &W->R[p];
    double Est = 0.0;
    for (int i = 1; i <= r->NATOMS; i++) {
      int ity = r->ity[i];
      double eta_ity = P->eta[ity];
      r->hshs[i] = eta_ity * r->hs[i];
      r->hsht[i] = eta_ity * r->ht[i];
      double qic = r->q[i] + P->Zpqeq[ity];
      double shelli[3] = {POS(r, i, 0) + SPOS(r, i, 0), POS(r, i, 1) + SPOS(r, i, 1), POS(r, i, 2) + SPOS(r, i, 2)};
      Est = Est + P->chi[ity] * r->q[i] + 0.5 * eta_ity * r->q[i] * r->q[i];
      for (int j1 = 1; j1 <= NBP(r, i, 0); j1++) {
        int j = NBP(r, i, j1), jty = r->ity[j];
        double qjc = r->q[j] + P->Zpqeq[jty];
        double shellj[3] = {POS(r, j, 0) + SPOS(r, j, 0), POS(r, j, 1) + SPOS(r, j, 1), POS(r, j, 2) + SPOS(r, j, 2)};
        double Ccicj = 0.0, Csicj = 0.0, Csisj = 0.0, ff[3];
        Ccicj = HES(r, i, j1) * qic * qjc;
        if (P->isPolarizable[ity]) {
          double d[3] = {shelli[0] - POS(r, j, 0), shelli[1] - POS(r, j, 1), shelli[2] - POS(r, j, 2)};
          pq_coulomb(P, d, PQ2(P->inxnpqeq, ity, jty), P->TBL_Eclmb_psc, &Csicj, ff);
          Csicj = -Cclmb0_qeq * Csicj * qjc * P->Zpqeq[ity];
          if (P->isPolarizable[jty]) {
            double d2[3] = {shelli[0] - shellj[0], shelli[1] - shellj[1], shelli[2] - shellj[2]};
            pq_coulomb(P, d2, PQ2(P->inxnpqeq, ity, jty), P->TBL_Eclmb_pss, &Csisj, ff);
            Csisj = Cclmb0_qeq * Csisj * P->Zpqeq[ity] * P->Zpqeq[jty];
          }
        }
        r->hshs[i] = r->hshs[i] + HES(r, i, j1) * r->hs[j];
        r->hsht[i] = r->hsht[i] + HES(r, i, j1) * r->ht[j];
        double Est1 = 0.5 * (Ccicj + Csisj);
        Est = Est + Est1 + Csicj;
      }
    }
    Ea[p] = Est;
  }
  return allreduce_sum(Ea, W->nprocs);
}

/* update_shell_positions, src/pqeq.F90:184-259 (Eqs. 37-39); sf / Esc / Ess are routine-scope (stale beyond the cutoff) */
static void pq_update_shells(World *W, Rank *r, long long *nstale) {
  const Params *P = &W->P;
  const double MAX_SHELL_DISPLACEMENT = 1e-3;
  int n = r->NATOMS;
  double *sforce = dalloc(3 * (size_t)(n + 1));
  double sf[3] = {0, 0, 0}, Esc = 0.0, Ess = 0.0;
  for (int i = 1; i <= n; i++) {
    int ity = r->ity[i];
    if (!P->isPolarizable[ity]) continue;
    if (P->isEfield) sforce[3 * i + P->eFieldDir - 1] = sforce[3 * i + P->eFieldDir - 1] - P->Zpqeq[ity] * P->eFieldStrength * Eev_kcal;   /* pqeq.F90:205 */
    for (int k = 0; k < 3; k++) sforce[3 * i + k] = sforce[3 * i + k] - P->Kspqeq[ity] * SPOS(r, i, k);
    double shelli[3] = {POS(r, i, 0) + SPOS(r, i, 0), POS(r, i, 1) + SPOS(r, i, 1), POS(r, i, 2) + SPOS(r, i, 2)};
    for (int j1 = 1; j1 <= NBP(r, i, 0); j1++) {
      int j = NBP(r, i, j1), jty = r->ity[j];
      double qjc = r->q[j] + P->Zpqeq[jty];
      double shellj[3] = {POS(r, j, 0) + SPOS(r, j, 0), POS(r, j, 1) + SPOS(r, j, 1), POS(r, j, 2) + SPOS(r, j, 2)};
      double d[3] = {shelli[0] - POS(r, j, 0), shelli[1] - POS(r, j, 1), shelli[2] - POS(r, j, 2)};
      if (!pq_coulomb(P, d, PQ2(P->inxnpqeq, ity, jty), P->TBL_Eclmb_psc, &Esc, sf)) (*nstale)++;
      for (int k = 0; k < 3; k++) { double ff = -Cclmb0 * sf[k] * qjc * P->Zpqeq[ity]; sforce[3 * i + k] = sforce[3 * i + k] - ff; }
      if (P->isPolarizable[jty]) {
        double d2[3] = {shelli[0] - shellj[0], shelli[1] - shellj[1], shelli[2] - shellj[2]};
        if (!pq_coulomb(P, d2, PQ2(P->inxnpqeq, ity, jty), P->TBL_Eclmb_pss, &Ess, sf)) (*nstale)++;
        for (int k = 0; k < 3; k++) { double ff = Cclmb0 * sf[k] * P->Zpqeq[ity] * P->Zpqeq[jty]; sforce[3 * i + k] = sforce[3 * i + k] - ff; }
      }
    }
  }
  for (int i = 1; i <= n; i++) {
    int ity = r->ity[i];
    double dr[3] = {sforce[3 * i] / P->Kspqeq[ity], sforce[3 * i + 1] / P->Kspqeq[ity], sforce[3 * i + 2] / P->Kspqeq[ity]};
    double ddr = sqrt(dr[0] * dr[0] + dr[1] * dr[1] + dr[2] * dr[2]);
    if (ddr > MAX_SHELL_DISPLACEMENT) for (int k = 0; k < 3; k++) dr[k] = dr[k] / ddr * MAX_SHELL_DISPLACEMENT;
    if (P->isPolarizable[ity]) for (int k = 0; k < 3; k++) SPOS(r, i, k) = SPOS(r, i, k) + dr[k];
  }
  free(sforce);
}

static int PQEq(World *W) { /* pqeq.F90:2-178: the QEq driver with the PQEq list / products, then the shell update */
  const Params *P = &W->P;
  int nmax;
  double QCopyDr[3] = {P->rctap / W->lata, P->rctap / W->latb, P->rctap / W->latc};
  for (int p = 0; p < W->nprocs; p++) {
    Rank *r = &W->R[p];
    if (W->isQEq == 1) {
      for (int i = 1; i <= r->NATOMS; i++) { r->qsfp[i] = r->q[i]; r->qsfv[i] = 0.0; }
      for (int i = 0; i < r->NBUFFER; i++) { r->qs[i] = 0.0; r->qt[i] = 0.0; }
      for (int i = 1; i <= r->NATOMS; i++) r->qs[i] = r->q[i];
    } else if (W->isQEq == 2) {
      for (int i = 1; i <= r->NATOMS; i++) { r->qs[i] = W->Lex_fqs * r->qsfp[i] + (1.0 - W->Lex_fqs) * r->q[i]; r->qt[i] = 0.0; }
    }
  }
  if (W->isQEq == 1) nmax = W->NMAXQEq; else if (W->isQEq == 2) nmax = 1; else return 0;
  W->ntrace = 0;
  if (COPYATOMS(W, MODE_COPY, QCopyDr)) return -1;
  for (int p = 0; p < W->nprocs; p++) {
    Rank *r = &W->R[p];
    if (LINKEDLIST(W, r, W->nblcsize, r->nbheader, r->nbllist, r->nbnacell, W->nbcc, MAXLAYERS_NB)) { snprintf(W->err, 256, "atom outside the cell grid (NB)"); return -1; }
    if (pq_initialize(W, r, &W->pq_stale)) return -1;
  }
  double Gnew[2], Gold[2];
  COPYATOMS(W, MODE_QCOPY1, QCopyDr);
  pq_get_gradient(W, Gnew);
  for (int p = 0; p < W->nprocs; p++) { Rank *r = &W->R[p]; for (int i = 1; i <= r->NATOMS; i++) { r->hs[i] = r->gs[i]; r->ht[i] = r->gt[i]; } }
  COPYATOMS(W, MODE_QCOPY2, QCopyDr);
  double GEst2 = 1e99, GEst1 = 0;
  int it;
  for (it = 0; it <= nmax - 1; it++) {
    GEst1 = pq_get_hsh(W);
    trace_push(W, GEst1, Gnew[0], Gnew[1]);
    if (0.5 * (fabs(GEst2) + fabs(GEst1)) < W->QEq_tol) break;
    if (fabs(GEst2) > 0.0 && (fabs(GEst1 / GEst2 - 1.0) < W->QEq_tol)) break;
    GEst2 = GEst1;
    double pa[64], pb[64], pc[64], pd[64];
    for (int p = 0; p < W->nprocs; p++) {              /* four dot_products then one allreduce, pqeq.F90:118-131 */
      Rank *r = &W->R[p];
      double a = 0, b = 0, c = 0, d = 0;
      for (int i = 1; i <= r->NATOMS; i++) a += r->gs[i] * r->hs[i];
      for (int i = 1; i <= r->NATOMS; i++) c += r->hs[i] * r->hshs[i];
      for (int i = 1; i <= r->NATOMS; i++) b += r->gt[i] * r->ht[i];
      for (int i = 1; i <= r->NATOMS; i++) d += r->ht[i] * r->hsht[i];
      pa[p] = a; pb[p] = b; pc[p] = c; pd[p] = d;
    }
    double g_h[2] = {allreduce_sum(pa, W->nprocs), allreduce_sum(pb, W->nprocs)}, h_hsh[2] = {allreduce_sum(pc, W->nprocs), allreduce_sum(pd, W->nprocs)};
    float lmin[2];
    lmin[0] = (float)(g_h[0] / h_hsh[0]); lmin[1] = (float)(g_h[1] / h_hsh[1]);
    for (int p = 0; p < W->nprocs; p++) {
      Rank *r = &W->R[p];
      double a = 0, b = 0;
      for (int i = 1; i <= r->NATOMS; i++) { r->qs[i] = r->qs[i] + (double)lmin[0] * r->hs[i]; r->qt[i] = r->qt[i] + (double)lmin[1] * r->ht[i]; }
      for (int i = 1; i <= r->NATOMS; i++) a += r->qs[i];
      for (int i = 1; i <= r->NATOMS; i++) b += r->qt[i];
      pa[p] = a; pb[p] = b;
    }
    double ssum = allreduce_sum(pa, W->nprocs), tsum = allreduce_sum(pb, W->nprocs);
    double mu = ssum / tsum;
    for (int p = 0; p < W->nprocs; p++) { Rank *r = &W->R[p]; for (int i = 1; i <= r->NATOMS; i++) r->q[i] = r->qs[i] - mu * r->qt[i]; }
    COPYATOMS(W, MODE_QCOPY1, QCopyDr);
    Gold[0] = Gnew[0]; Gold[1] = Gnew[1];
    pq_get_gradient(W, Gnew);
    for (int p = 0; p < W->nprocs; p++) {
      Rank *r = &W->R[p];
      for (int i = 1; i <= r->NATOMS; i++) { r->hs[i] = r->gs[i] + (Gnew[0] / Gold[0]) * r->hs[i]; r->ht[i] = r->gt[i] + (Gnew[1] / Gold[1]) * r->ht[i]; }
    }
    COPYATOMS(W, MODE_QCOPY2, QCopyDr);
  }
  for (int p = 0; p < W->nprocs; p++) pq_update_shells(W, &W->R[p], &W->pq_stale);     /* pqeq.F90:169 */
  W->nstep_qeq = it;
  W->qeq_iters_total += it;
  return 0;
}

/* ------------------------------------------------------------------ BOCALC, src/bo.F90 */
static void BOPRIM(World *W, Rank *r) { /* bo.F90:28-118 */
  const Params *P = &W->P;
  int G = r->copyptr[6];
  for (int i = 1; i <= G; i++) r->deltap[2 * i] = -P->Val[r->ity[i]];
  for (int i = 1; i <= G; i++) {
    int ity = r->ity[i];
    for (int j1 = 1; j1 <= NBR(r, i, 0); j1++) {
      int j = NBR(r, i, j1);
      if (j >= i) continue;
      int jty = r->ity[j], inxn = T2(P->inxn2, ity, jty), i1 = NBX(r, i, j1);
      double d0 = POS(r, i, 0) - POS(r, j, 0), d1 = POS(r, i, 1) - POS(r, j, 1), d2 = POS(r, i, 2) - POS(r, j, 2);
      double dr2 = d0 * d0 + d1 * d1 + d2 * d2;
      if (dr2 <= P->rc2[inxn]) {
        double arg[3] = {P->cBOp1[inxn] * pow(dr2, P->pbo2h[inxn]), P->cBOp3[inxn] * pow(dr2, P->pbo4h[inxn]), P->cBOp5[inxn] * pow(dr2, P->pbo6h[inxn])};
        double bo[4];
        for (int k = 0; k < 3; k++) bo[k + 1] = P->swtch[3 * inxn + k] * exp(arg[k]);
        bo[1] = (1.0 + P->cutoff_vpar30) * bo[1];
        if (bo[1] + bo[2] + bo[3] > P->cutoff_vpar30) {
          double dl[3] = {P->swtch[3 * inxn] * P->pbo2[inxn] * arg[0], P->swtch[3 * inxn + 1] * P->pbo4[inxn] * arg[1], P->swtch[3 * inxn + 2] * P->pbo6[inxn] * arg[2]};
          for (int k = 0; k < 3; k++) { dl[k] = dl[k] / dr2; DLN(r, k + 1, i, j1) = dl[k]; DLN(r, k + 1, j, i1) = dl[k]; }
          double dB = bo[1] * dl[0] + bo[2] * dl[1] + bo[3] * dl[2];
          r->dBOp[SL(r, i, j1)] = dB; r->dBOp[SL(r, j, i1)] = dB;
          bo[1] = bo[1] - P->cutoff_vpar30;
          bo[0] = bo[1] + bo[2] + bo[3];
          for (int k = 0; k < 4; k++) { BOa(r, k, i, j1) = bo[k]; BOa(r, k, j, i1) = bo[k]; }
          r->deltap[2 * i] += bo[0]; r->deltap[2 * j] += bo[0];
        } else {
          r->dBOp[SL(r, i, j1)] = 0.0; r->dBOp[SL(r, j, i1)] = 0.0;
          for (int k = 0; k < 4; k++) { BOa(r, k, i, j1) = 0.0; BOa(r, k, j, i1) = 0.0; }
        }
      }
    }
  }
}

static void BOFULL(World *W, Rank *r) { /* bo.F90:121-298 */
  const Params *P = &W->P;
  int G = r->copyptr[6];
  double vpar1 = P->vpar1, vpar2 = P->vpar2;
  for (int i = 1; i <= G; i++) r->deltap[2 * i + 1] = r->deltap[2 * i] + P->Val[r->ity[i]] - P->Valval[r->ity[i]];
  for (int i = 1; i <= G; i++) {
    int ity = r->ity[i];
    double exppboc1i = exp(-vpar1 * r->deltap[2 * i]), exppboc2i = exp(-vpar2 * r->deltap[2 * i]);
    for (int j1 = 1; j1 <= NBR(r, i, 0); j1++) {
      int j = NBR(r, i, j1);
      if (j >= i) continue;
      int jty = r->ity[j];
      double exppboc1j = exp(-vpar1 * r->deltap[2 * j]), exppboc2j = exp(-vpar2 * r->deltap[2 * j]);
      int i1 = NBX(r, i, j1), inxn = T2(P->inxn2, ity, jty);
      double fn2 = exppboc1i + exppboc1j;
      double fn3 = (-1.0 / vpar2) * log(0.5 * (exppboc2i + exppboc2j));
      double fn23 = fn2 + fn3;
      double BOp0 = BOa(r, 0, i, j1);
      double fn1 = 0.5 * ((P->Val[ity] + fn2) / (P->Val[ity] + fn23) + (P->Val[jty] + fn2) / (P->Val[jty] + fn23));
      if (P->ovc[inxn] < 1e-3) fn1 = 1.0;
      double BOpsqr = BOa(r, 0, i, j1) * BOa(r, 0, i, j1);
      double fn4 = 1.0 / (1.0 + exp(-P->pboc3[inxn] * (P->pboc4[inxn] * BOpsqr - r->deltap[2 * i + 1]) + P->pboc5[inxn]));
      double fn5 = 1.0 / (1.0 + exp(-P->pboc3[inxn] * (P->pboc4[inxn] * BOpsqr - r->deltap[2 * j + 1]) + P->pboc5[inxn]));
      if (P->v13cor[inxn] < 1e-3) { fn4 = 1.0; fn5 = 1.0; }
      double fn45 = fn4 * fn5, fn145 = fn1 * fn45, fn1145 = fn1 * fn145;
      double B0 = BOa(r, 0, i, j1) * fn145, B2 = BOa(r, 2, i, j1) * fn1145, B3 = BOa(r, 3, i, j1) * fn1145;
      if (B0 < 1e-10) B0 = 0.0;
      if (B2 < 1e-10) B2 = 0.0;
      if (B3 < 1e-10) B3 = 0.0;
      double B1 = B0 - B2 - B3;
      BOa(r, 0, i, j1) = B0; BOa(r, 1, i, j1) = B1; BOa(r, 2, i, j1) = B2; BOa(r, 3, i, j1) = B3;
      BOa(r, 0, j, i1) = B0; BOa(r, 1, j, i1) = B1; BOa(r, 2, j, i1) = B2; BOa(r, 3, j, i1) = B3;
      double u1ij = P->Val[ity] + fn23, u1ji = P->Val[jty] + fn23;
      double u1ij_inv2 = 1.0 / (u1ij * u1ij), u1ji_inv2 = 1.0 / (u1ji * u1ji);
      double Cf1Aij = 0.5 * fn3 * (u1ij_inv2 + u1ji_inv2);
      double Cf1Bij = -0.5 * ((u1ij - fn3) * u1ij_inv2 + (u1ji - fn3) * u1ji_inv2);
      double exp_delt22 = exppboc2i + exppboc2j;
      double Cf1ij = (-Cf1Aij * P->pboc1[inxn] * exppboc1i) + (Cf1Bij * exppboc2i) / (exp_delt22);
      double Cf1ji = (-Cf1Aij * P->pboc1[inxn] * exppboc1j) + (Cf1Bij * exppboc2j) / (exp_delt22);
      double pboc34 = P->pboc3[inxn] * P->pboc4[inxn];
      double u45ij = P->pboc5[inxn] + P->pboc3[inxn] * r->deltap[2 * i + 1] - pboc34 * BOpsqr;
      double u45ji = P->pboc5[inxn] + P->pboc3[inxn] * r->deltap[2 * j + 1] - pboc34 * BOpsqr;
      double exph_45ij = exp(u45ij), exph_45ji = exp(u45ji);
      double exp1 = 1.0 / (1.0 + exph_45ij), exp2 = 1.0 / (1.0 + exph_45ji), exp12 = exp1 * exp2;
      double Cf45ij = -exph_45ij * exp12 * exp1, Cf45ji = -exph_45ji * exp12 * exp2;
      if (P->ovc[inxn] < 1e-3) { Cf1ij = 0.0; Cf1ji = 0.0; }
      if (P->v13cor[inxn] < 1e-3) { Cf45ij = 0.0; Cf45ji = 0.0; }
      double fn45_inv = 1.0 / fn45, Cf1ij_div1 = Cf1ij / fn1, Cf1ji_div1 = Cf1ji / fn1;
      size_t a = SL(r, i, j1), b = SL(r, j, i1);
      r->A0[a] = fn145;
      r->A1[a] = -2.0 * pboc34 * BOp0 * (Cf45ij + Cf45ji) * fn45_inv;
      r->A2[a] = Cf1ij_div1 + (P->pboc3[inxn] * Cf45ij * fn45_inv);
      r->A3[a] = r->A2[a] + Cf1ij_div1;
      r->A0[b] = r->A0[a]; r->A1[b] = r->A1[a];
      r->A2[b] = Cf1ji_div1 + (P->pboc3[inxn] * Cf45ji * fn45_inv);
      r->A3[b] = r->A2[b] + Cf1ji_div1;
    }
  }
  for (int i = 1; i <= G; i++) {
    double s = 0.0;
    for (int j1 = 1; j1 <= NBR(r, i, 0); j1++) s += BOa(r, 0, i, j1);
    r->delta[i] = -P->Val[r->ity[i]] + s;
  }
}

/* ------------------------------------------------------------------ force helpers, src/pot.F90:1230-1543 */
static void ForceD(Rank *r, int i, double coeff) { /* pot.F90:1230-1273 */
  for (int j1 = 1; j1 <= NBR(r, i, 0); j1++) {
    int j = NBR(r, i, j1), i1 = NBX(r, i, j1);
    size_t a = SL(r, i, j1), b = SL(r, j, i1);
    double Cb1 = coeff * (r->A0[a] + BOa(r, 0, i, j1) * r->A1[a]);
    for (int k = 0; k < 3; k++) { double ff = Cb1 * r->dBOp[a] * (POS(r, i, k) - POS(r, j, k)); FRC(r, i, k) -= ff; FRC(r, j, k) += ff; }
    r->ccbnd[i] += coeff * BOa(r, 0, i, j1) * r->A2[a];
    r->ccbnd[j] += coeff * BOa(r, 0, i, j1) * r->A2[b];
  }
}
static void ForceB(Rank *r, int i, int j1, int j, int i1, double coeff) { /* pot.F90:1276-1316 */
  size_t a = SL(r, i, j1), b = SL(r, j, i1);
  double Cb1 = coeff * (r->A0[a] + BOa(r, 0, i, j1) * r->A1[a]);
  for (int k = 0; k < 3; k++) { double ff = Cb1 * r->dBOp[a] * (POS(r, i, k) - POS(r, j, k)); FRC(r, i, k) -= ff; FRC(r, j, k) += ff; }
  r->ccbnd[i] += coeff * BOa(r, 0, i, j1) * r->A2[a];
  r->ccbnd[j] += coeff * BOa(r, 0, i, j1) * r->A2[b];
}
static void ForceBbo(Rank *r, int i, int j1, int j, int i1, const double coeff[3]) { /* pot.F90:1319-1365 */
  size_t a = SL(r, i, j1), b = SL(r, j, i1);
  double cf[3] = {coeff[0], coeff[1] - coeff[0], coeff[2] - coeff[0]};
  double Cb1 = cf[0] * (r->A0[a] + BOa(r, 0, i, j1) * r->A1[a]) * r->dBOp[a]
             + cf[1] * BOa(r, 2, i, j1) * (DLN(r, 2, i, j1) + r->A1[a] * r->dBOp[a])
             + cf[2] * BOa(r, 3, i, j1) * (DLN(r, 3, i, j1) + r->A1[a] * r->dBOp[a]);
  for (int k = 0; k < 3; k++) { double ff = Cb1 * (POS(r, i, k) - POS(r, j, k)); FRC(r, i, k) -= ff; FRC(r, j, k) += ff; }
  double cBO[3] = {cf[0] * BOa(r, 0, i, j1), cf[1] * BOa(r, 2, i, j1), cf[2] * BOa(r, 3, i, j1)};
  r->ccbnd[i] += cBO[0] * r->A2[a] + (cBO[1] + cBO[2]) * r->A3[a];
  r->ccbnd[j] += cBO[0] * r->A2[b] + (cBO[1] + cBO[2]) * r->A3[b];
}
static void ForceA3(Rank *r, double coeff, int i, int j, int k, const double da0[4], const double da1[4]) { /* pot.F90:1462-1521 */
  double C00 = da0[0] * da0[0], C01 = da0[1] * da1[1] + da0[2] * da1[2] + da0[3] * da1[3], C11 = da1[0] * da1[0];
  double CCisqr = 1.0 / (da0[0] * da1[0]), coCC = coeff * CCisqr;
  double Ci1 = -(C01 / C00), Ci2 = 1.0, Ck1 = -1.0, Ck2 = C01 / C11;
  for (int a = 0; a < 3; a++) {
    double fij = coCC * (Ci1 * da0[a + 1] + Ci2 * da1[a + 1]);
    double fjk = -coCC * (Ck1 * da0[a + 1] + Ck2 * da1[a + 1]);
    double fijjk = -fij + fjk;
    FRC(r, i, a) += fij; FRC(r, j, a) += fijjk; FRC(r, k, a) -= fjk;
  }
}
static void ForceA4(Rank *r, double coeff, int i, int j, int k, int l, const double da0[4], const double da1[4], const double da2[4]) { /* pot.F90:1369-1459 */
#define DOT3(x, y) ((x)[1] * (y)[1] + (x)[2] * (y)[2] + (x)[3] * (y)[3])
  double C00 = da0[0] * da0[0], C01 = DOT3(da0, da1), C02 = DOT3(da0, da2);
  double C11 = da1[0] * da1[0], C12 = DOT3(da1, da2), C22 = da2[0] * da2[0];
  double D0 = C00 * C11 - C01 * C01, Dm1 = C11 * C22 - C12 * C12;
  double DDisqr = 1.0 / sqrt(D0 * Dm1), coDD = coeff * DDisqr;
  double com = C01 * C12 - C02 * C11;
  double Cwi1 = C11 / D0 * com, Cwi2 = -(C12 + C01 / D0 * com), Cwi3 = C11;
  double Cwj1 = -(C12 + (C11 + C01) / D0 * com);
  double Cwj2 = -(-C12 - 2 * C02 - C22 / Dm1 * com - (C00 + C01) / D0 * com);
  double Cwj3 = -(C01 + C11 + C12 / Dm1 * com);
  double Cwl1 = -C11, Cwl2 = (C01 + C12 / Dm1 * com), Cwl3 = -(C11 / Dm1 * com);
  for (int a = 1; a <= 3; a++) {
    double fij = coDD * (Cwi1 * da0[a] + Cwi2 * da1[a] + Cwi3 * da2[a]);
    double fjk = coDD * ((Cwj1 + Cwi1) * da0[a] + (Cwj2 + Cwi2) * da1[a] + (Cwj3 + Cwi3) * da2[a]);
    double fkl = -coDD * (Cwl1 * da0[a] + Cwl2 * da1[a] + Cwl3 * da2[a]);
    FRC(r, i, a - 1) += fij; FRC(r, j, a - 1) += -fij + fjk; FRC(r, k, a - 1) += -fjk + fkl; FRC(r, l, a - 1) -= fkl;
  }
}
static void cross_product(const double d1[4], const double d2[4], double crs[4]) { /* pot.F90:1524-1543 */
  double n1[3] = {d1[1] / d1[0], d1[2] / d1[0], d1[3] / d1[0]}, n2[3] = {d2[1] / d2[0], d2[2] / d2[0], d2[3] / d2[0]};
  crs[1] = n1[1] * n2[2] - n1[2] * n2[1]; crs[2] = n1[2] * n2[0] - n1[0] * n2[2]; crs[3] = n1[0] * n2[1] - n1[1] * n2[0];
  crs[0] = sqrt(crs[1] * crs[1] + crs[2] * crs[2] + crs[3] * crs[3]);
  if (crs[0] < NSMALL) crs[0] = NSMALL;
}
static void vec(const Rank *r, int a, int b, double out[4]) { /* out(1:3)=pos(a)-pos(b), out(0)=norm */
  out[1] = POS(r, a, 0) - POS(r, b, 0); out[2] = POS(r, a, 1) - POS(r, b, 1); out[3] = POS(r, a, 2) - POS(r, b, 2);
  out[0] = sqrt(out[1] * out[1] + out[2] * out[2] + out[3] * out[3]);
}

/* ------------------------------------------------------------------ energy terms, src/pot.F90 */
static void ENbond(World *W, Rank *r) { /* pot.F90:676-781 */
  const Params *P = &W->P;
  for (int i = 1; i <= r->NATOMS; i++) {
    int ity = r->ity[i]; long long iid = r->gid[i];
    r->PE[13] += CEchrge * (P->chi[ity] * r->q[i] + 0.5 * P->eta[ity] * r->q[i] * r->q[i]);
    for (int j1 = 1; j1 <= NBP(r, i, 0); j1++) {
      int j = NBP(r, i, j1);
      if (r->gid[j] >= iid) continue;
      double d[3] = {POS(r, i, 0) - POS(r, j, 0), POS(r, i, 1) - POS(r, j, 1), POS(r, i, 2) - POS(r, j, 2)};
      double dr2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
      if (dr2 > P->rctap2) continue;
      int inxn = T2(P->inxn2, ity, r->ity[j]);
      int itb = (int)(dr2 * P->UDRi), itb1 = itb + 1;
      double drtb = dr2 - itb * P->UDR; drtb = drtb * P->UDRi;
      double drtb1 = 1.0 - drtb;
      const double *Tv = P->TBL_Evdw + (size_t)inxn * (NTABLE + 2) * 2, *Tc = P->TBL_Eclmb + (size_t)inxn * (NTABLE + 2) * 2;
      double PEvdw = drtb1 * Tv[2 * itb] + drtb * Tv[2 * itb1], CEvdw = drtb1 * Tv[2 * itb + 1] + drtb * Tv[2 * itb1 + 1];
      double qij = r->q[i] * r->q[j];
      double PEclmb = drtb1 * Tc[2 * itb] + drtb * Tc[2 * itb1]; PEclmb = PEclmb * qij;
      double CEclmb = drtb1 * Tc[2 * itb + 1] + drtb * Tc[2 * itb1 + 1]; CEclmb = CEclmb * qij;
      r->PE[11] += PEvdw; r->PE[12] += PEclmb;
      for (int k = 0; k < 3; k++) { double ff = (CEvdw + CEclmb) * d[k]; FRC(r, i, k) -= ff; FRC(r, j, k) += ff; }
    }
  }
}

static void ENbond_PQEq(World *W, Rank *r) { /* pot.F90:784-923 */
  const Params *P = &W->P;
  for (int i = 1; i <= r->NATOMS; i++) {
    int ity = r->ity[i]; long long iid = r->gid[i];
    double Eshell = 0.0;
    if (P->isPolarizable[ity]) {
      double dr2 = SPOS(r, i, 0) * SPOS(r, i, 0) + SPOS(r, i, 1) * SPOS(r, i, 1) + SPOS(r, i, 2) * SPOS(r, i, 2);
      Eshell = 0.5 * P->Kspqeq[ity] * dr2;
    }
    r->PE[13] = r->PE[13] + CEchrge * (P->chi[ity] * r->q[i] + 0.5 * P->eta[ity] * r->q[i] * r->q[i]) + Eshell;
    double qic = r->q[i] + P->Zpqeq[ity];
    for (int j1 = 1; j1 <= NBP(r, i, 0); j1++) {
      int j = NBP(r, i, j1);
      if (!(iid < r->gid[j])) continue;                 /* if(iid<jid), :830 -- the opposite half of ENbond's rule */
      double dr[3] = {POS(r, i, 0) - POS(r, j, 0), POS(r, i, 1) - POS(r, j, 1), POS(r, i, 2) - POS(r, j, 2)};
      double dr2 = dr[0] * dr[0] + dr[1] * dr[1] + dr[2] * dr[2];
      int jty = r->ity[j], inxn = T2(P->inxn2, ity, jty);
      int itb = (int)(dr2 * P->UDRi), itb1 = itb + 1;
      double drtb = dr2 - itb * P->UDR; drtb = drtb * P->UDRi;
      double drtb1 = 1.0 - drtb;
      const double *Tv = P->TBL_Evdw + (size_t)inxn * (NTABLE + 2) * 2;
      double PEvdw = drtb1 * Tv[2 * itb] + drtb * Tv[2 * itb1], CEvdw = drtb1 * Tv[2 * itb + 1] + drtb * Tv[2 * itb1 + 1];
      double qjc = r->q[j] + P->Zpqeq[jty], qij = qic * qjc;
      double Ecc = 0, Esc = 0, Ecs = 0, Ess = 0, fcc[3] = {0, 0, 0}, fsc[3] = {0, 0, 0}, fcs[3] = {0, 0, 0}, fss[3] = {0, 0, 0};
      pq_coulomb(P, dr, PQ2(P->inxnpqeq, ity, jty), P->TBL_Eclmb_pcc, &Ecc, fcc);
      for (int k = 0; k < 3; k++) fcc[k] = Cclmb0 * qij * fcc[k];
      Ecc = Cclmb0 * Ecc * qij;
      if (P->isPolarizable[ity]) {
        double d[3] = {dr[0] + SPOS(r, i, 0), dr[1] + SPOS(r, i, 1), dr[2] + SPOS(r, i, 2)};
        pq_coulomb(P, d, PQ2(P->inxnpqeq, ity, jty), P->TBL_Eclmb_psc, &Esc, fsc);
        for (int k = 0; k < 3; k++) fsc[k] = -Cclmb0 * P->Zpqeq[ity] * qjc * fsc[k];
        Esc = -Cclmb0 * Esc * P->Zpqeq[ity] * qjc;
      }
      if (P->isPolarizable[jty]) {
        double d[3] = {dr[0] - SPOS(r, j, 0), dr[1] - SPOS(r, j, 1), dr[2] - SPOS(r, j, 2)};
        pq_coulomb(P, d, PQ2(P->inxnpqeq, jty, ity), P->TBL_Eclmb_psc, &Ecs, fcs);
        for (int k = 0; k < 3; k++) fcs[k] = -Cclmb0 * P->Zpqeq[jty] * qic * fcs[k];
        Ecs = -Cclmb0 * Ecs * qic * P->Zpqeq[jty];
      }
      if (P->isPolarizable[ity] && P->isPolarizable[jty]) {
        double d[3] = {dr[0] + SPOS(r, i, 0) - SPOS(r, j, 0), dr[1] + SPOS(r, i, 1) - SPOS(r, j, 1), dr[2] + SPOS(r, i, 2) - SPOS(r, j, 2)};
        pq_coulomb(P, d, PQ2(P->inxnpqeq, ity, jty), P->TBL_Eclmb_pss, &Ess, fss);
        for (int k = 0; k < 3; k++) fss[k] = Cclmb0 * P->Zpqeq[ity] * P->Zpqeq[jty] * fss[k];
        Ess = Cclmb0 * Ess * P->Zpqeq[ity] * P->Zpqeq[jty];
      }
      double PEclmb = Ecc + Esc + Ecs + Ess;
      r->PE[11] += PEvdw; r->PE[12] += PEclmb;
      for (int k = 0; k < 3; k++) { double ff = CEvdw * dr[k] + fcc[k] + fcs[k] + fsc[k] + fss[k]; FRC(r, i, k) -= ff; FRC(r, j, k) += ff; }
    }
  }
}

static void Ebond(World *W, Rank *r) { /* pot.F90:926-977 */
  const Params *P = &W->P;
  for (int i = 1; i <= r->NATOMS; i++) {
    int ity = r->ity[i]; long long iid = r->gid[i];
    for (int j1 = 1; j1 <= NBR(r, i, 0); j1++) {
      int j = NBR(r, i, j1);
      if (r->gid[j] >= iid) continue;
      int inxn = T2(P->inxn2, ity, r->ity[j]);
      double B1 = BOa(r, 1, i, j1);
      double exp_be12 = exp(P->pbe1[inxn] * (1.0 - pow(B1, P->pbe2[inxn])));
      double PEbo = -P->Desig[inxn] * B1 * exp_be12 - P->Depi[inxn] * BOa(r, 2, i, j1) - P->Depipi[inxn] * BOa(r, 3, i, j1);
      r->PE[1] += PEbo;
      double CEbo = -P->Desig[inxn] * exp_be12 * (1.0 - P->pbe1[inxn] * P->pbe2[inxn] * pow(B1, P->pbe2[inxn]));
      double coeff[3] = {CEbo, -P->Depi[inxn], -P->Depipi[inxn]};
      ForceBbo(r, i, j1, j, NBX(r, i, j1), coeff);
    }
  }
}

static void Elnpr(World *W, Rank *r) { /* pot.F90:148-316 */
  const Params *P = &W->P;
  for (int i = 1; i <= r->copyptr[6]; i++) {          /* preparation, :183-209 */
    int ity = r->ity[i];
    if (ity == 0) continue;
    double deltaE = -P->Vale[ity] + P->Val[ity] + r->delta[i];
    double dEh = deltaE * 0.5;
    int idEh = is_idEh * (int)dEh;
    double u = 2.0 + deltaE - 2 * idEh;
    double explp1 = exp(-P->plp1[ity] * (u * u));
    double Clp = 2.0 * P->plp1[ity] * explp1 * (2.0 + deltaE - 2 * idEh);
    r->dDlp[i] = Clp;
    r->nlp[i] = explp1 - (double)idEh;
    r->deltalp[i] = P->nlpopt[ity] - r->nlp[i];
    if (P->mass[ity] > 21.0) r->deltalp[i] = 0.0;
  }
  for (int i = 1; i <= r->NATOMS; i++) {
    int ity = r->ity[i];
    double sum_ovun1 = 0.0, sum_ovun2 = 0.0;
    for (int j1 = 1; j1 <= NBR(r, i, 0); j1++) {
      int j = NBR(r, i, j1), inxn = T2(P->inxn2, ity, r->ity[j]);
      sum_ovun1 = sum_ovun1 + P->povun1[inxn] * P->Desig[inxn] * BOa(r, 0, i, j1);
      sum_ovun2 = sum_ovun2 + (r->delta[j] - r->deltalp[j]) * (BOa(r, 2, i, j1) + BOa(r, 3, i, j1));
    }
    double expvd2 = exp(-75.0 * r->deltalp[i]);
    double dElp = P->plp2[ity] * ((1.0 + expvd2) + 75.0 * r->deltalp[i] * expvd2) / ((1.0 + expvd2) * (1.0 + expvd2));
    double expovun1 = P->povun3[ity] * exp(P->povun4[ity] * sum_ovun2);
    double deltalpcorr = r->delta[i] - r->deltalp[i] / (1.0 + expovun1);
    double expovun2 = exp(P->povun2[ity] * deltalpcorr);
    double DlpV_i = 1.0 / (deltalpcorr + P->Val[ity] + 1e-8);
    double expovun2n = 1.0 / expovun2;
    double expovun6 = exp(P->povun6[ity] * deltalpcorr);
    double expovun8 = P->povun7[ity] * exp(P->povun8[ity] * sum_ovun2);
    double div_expovun1 = 1.0 / (1.0 + expovun1), div_expovun2 = 1.0 / (1.0 + expovun2);
    double div_expovun2n = 1.0 / (1.0 + expovun2n), div_expovun8 = 1.0 / (1.0 + expovun8);
    double PElp = P->plp2[ity] * r->deltalp[i] / (1.0 + expvd2);
    double PEover = sum_ovun1 * DlpV_i * deltalpcorr * div_expovun2;
    double PEunder = -P->povun5[ity] * (1.0 - expovun6) * div_expovun2n * div_expovun8;
    r->PE[2] += PElp; r->PE[3] += PEover; r->PE[4] += PEunder;
    double CElp1 = dElp * r->dDlp[i];
    double CEover[8], CEunder[7];
    CEover[1] = deltalpcorr * DlpV_i * div_expovun2;
    CEover[2] = sum_ovun1 * DlpV_i * div_expovun2 * (1.0 - deltalpcorr * DlpV_i - P->povun2[ity] * deltalpcorr * div_expovun2n);
    CEover[3] = CEover[2] * (1.0 - r->dDlp[i] * div_expovun1);
    CEover[4] = CEover[2] * r->deltalp[i] * P->povun4[ity] * expovun1 * (div_expovun1 * div_expovun1);
    CEunder[1] = (P->povun5[ity] * P->povun6[ity] * expovun6 * div_expovun8 + PEunder * P->povun2[ity] * expovun2n) * div_expovun2n;
    CEunder[2] = -PEunder * P->povun8[ity] * expovun8 * div_expovun8;
    CEunder[3] = CEunder[1] * (1.0 - r->dDlp[i] * div_expovun1);
    CEunder[4] = CEunder[1] * r->deltalp[i] * P->povun4[ity] * expovun1 * (div_expovun1 * div_expovun1) + CEunder[2];
    for (int j1 = 1; j1 <= NBR(r, i, 0); j1++) {
      int j = NBR(r, i, j1), inxn = T2(P->inxn2, ity, r->ity[j]);
      double bpp = BOa(r, 2, i, j1) + BOa(r, 3, i, j1);
      CEover[5] = CEover[1] * P->povun1[inxn] * P->Desig[inxn];
      CEover[6] = CEover[4] * (1.0 - r->dDlp[j]) * bpp;
      CEover[7] = CEover[4] * (r->delta[j] - r->deltalp[j]);
      CEunder[5] = CEunder[4] * (1.0 - r->dDlp[j]) * bpp;
      CEunder[6] = CEunder[4] * (r->delta[j] - r->deltalp[j]);
      double CElp_b = CElp1 + CEover[3] + CEover[5] + CEunder[3];
      double CElp_bpp = CEover[7] + CEunder[6];
      double coeff[3] = {CElp_b + 0.0, CElp_b + CElp_bpp, CElp_b + CElp_bpp};
      ForceBbo(r, i, j1, j, NBX(r, i, j1), coeff);
      r->cdbnd[j] += CEover[6] + CEunder[5];
    }
  }
}

static void E3b(World *W, Rank *r) { /* pot.F90:319-557 */
  const Params *P = &W->P;
  for (int j = 1; j <= r->NATOMS; j++) {
    int jty = r->ity[j], nj = NBR(r, j, 0);
    double sum_BO8 = 0.0, sum_SBO1 = 0.0;
    for (int n1 = 1; n1 <= nj; n1++) { sum_BO8 = sum_BO8 - pow(BOa(r, 0, j, n1), 8.0); sum_SBO1 = sum_SBO1 + BOa(r, 2, j, n1) + BOa(r, 3, j, n1); }
    double prod_SBO = exp(sum_BO8);
    double delta_ang = r->delta[j] + P->Val[jty] - P->Valangle[jty];
    for (int i1 = 1; i1 <= nj - 1; i1++) {
      double BOij = BOa(r, 0, j, i1) - cutof2_esub;
      if (!(BOij > 0.0)) continue;
      int i = NBR(r, j, i1), ity = r->ity[i];
      double rij[4]; vec(r, i, j, rij);
      for (int k1 = i1 + 1; k1 <= nj; k1++) {
        double BOjk = BOa(r, 0, j, k1) - cutof2_esub;
        if (!(BOjk > 0.0)) continue;
        if (!(BOa(r, 0, j, i1) * BOa(r, 0, j, k1) > cutof2_esub)) continue;
        int k = NBR(r, j, k1), kty = r->ity[k];
        double rjk[4]; vec(r, j, k, rjk);
        double cos_ijk = -(rij[1] * rjk[1] + rij[2] * rjk[2] + rij[3] * rjk[3]) / (rij[0] * rjk[0]);
        if (cos_ijk > MAXANGLE) cos_ijk = MAXANGLE;
        if (cos_ijk < MINANGLE) cos_ijk = MINANGLE;
        double theta_ijk = acos(cos_ijk), sin_ijk = sin(theta_ijk);
        int inxn = T3(P->inxn3, ity, jty, kty);
        if (inxn == 0) continue;
        double BOij_p4 = pow(BOij, P->pval4[inxn]), exp3ij = exp(-P->pval3[jty] * BOij_p4), fn7ij = 1.0 - exp3ij;
        double BOjk_p4 = pow(BOjk, P->pval4[inxn]), exp3jk = exp(-P->pval3[jty] * BOjk_p4), fn7jk = 1.0 - exp3jk;
        double exp6 = exp(P->pval6[inxn] * delta_ang), exp7 = exp(-P->pval7[inxn] * delta_ang), trm8 = 1.0 + exp6 + exp7;
        double fn8j = P->pval5[jty] - (P->pval5[jty] - 1.0) * (2.0 + exp6) / trm8;
        double SBO = sum_SBO1 + (1.0 - prod_SBO) * (-delta_ang - P->pval8[inxn] * r->nlp[j]), SBO2 = 0.0;
        if (SBO <= 0) SBO2 = 0.0;
        if (SBO > 0) SBO2 = pow(SBO, P->pval9[inxn]);
        if (SBO > 1) SBO2 = 2.0 - pow(2.0 - SBO, P->pval9[inxn]);
        if (SBO > 2) SBO2 = 2.0;
        double theta0 = PI_ - P->theta00[inxn] * (1.0 - exp(-P->pval10[inxn] * (2.0 - SBO2)));
        double theta_diff = theta0 - theta_ijk;
        double exp2 = exp(-P->pval2[inxn] * theta_diff * theta_diff);
        double PEval = fn7ij * fn7jk * fn8j * (P->pval1[inxn] - P->pval1[inxn] * exp2);
        double Cf7ij = P->pval3[jty] * P->pval4[inxn] * pow(BOij, P->pval4[inxn] - 1.0) * exp3ij;
        double Cf7jk = P->pval3[jty] * P->pval4[inxn] * pow(BOjk, P->pval4[inxn] - 1.0) * exp3jk;
        double Cf8j = (1.0 - P->pval5[jty]) / (trm8 * trm8) * (P->pval6[inxn] * exp6 * trm8 - (2.0 + exp6) * (P->pval6[inxn] * exp6 - P->pval7[inxn] * exp7));
        double Ctheta_diff = 2.0 * P->pval2[inxn] * theta_diff * exp2 / (1.0 - exp2); (void)Ctheta_diff;
        double Ctheta0 = P->pval10[inxn] * P->theta00[inxn] * exp(-P->pval10[inxn] * (2.0 - SBO2));
        double CSBO2 = 0.0;
        if (SBO <= 0 || SBO > 2) CSBO2 = 0.0;
        if (SBO > 0 && SBO <= 1) CSBO2 = P->pval9[inxn] * pow(SBO, P->pval9[inxn] - 1.0);
        if (SBO > 1 && SBO <= 2) CSBO2 = P->pval9[inxn] * pow(2.0 - SBO, P->pval9[inxn] - 1.0);
        double dSBO1 = -8.0 * prod_SBO * (delta_ang + P->pval8[inxn] * r->nlp[j]);
        double dSBO2 = (prod_SBO - 1.0) * (1.0 - P->pval8[inxn] * r->dDlp[j]);
        double CEval[9];
        CEval[1] = Cf7ij * fn7jk * fn8j * P->pval1[inxn] * (1.0 - exp2);
        CEval[2] = fn7ij * Cf7jk * fn8j * P->pval1[inxn] * (1.0 - exp2);
        CEval[3] = fn7ij * fn7jk * Cf8j * P->pval1[inxn] * (1.0 - exp2);
        CEval[4] = 2.0 * P->pval1[inxn] * P->pval2[inxn] * fn7ij * fn7jk * fn8j * exp2 * theta_diff;
        CEval[5] = CEval[4] * Ctheta0 * CSBO2;
        CEval[6] = CEval[5] * dSBO1;
        CEval[7] = CEval[5] * dSBO2;
        CEval[8] = CEval[4] / sin_ijk;
        double exp_pen3 = exp(-P->ppen3[inxn] * r->delta[j]), exp_pen4 = exp(P->ppen4[inxn] * r->delta[j]);
        double fn9 = (2.0 + exp_pen3) / (1.0 + exp_pen3 + exp_pen4);
        double exp_pen2ij = exp(-P->ppen2[inxn] * (BOij - 2.0) * (BOij - 2.0)), exp_pen2jk = exp(-P->ppen2[inxn] * (BOjk - 2.0) * (BOjk - 2.0));
        double PEpen = P->ppen1[inxn] * fn9 * exp_pen2ij * exp_pen2jk;
        double trm_pen34 = 1.0 + exp_pen3 + exp_pen4;
        double Cf9j = (-P->ppen3[inxn] * exp_pen3 * trm_pen34 - (2.0 + exp_pen3) * (-P->ppen3[inxn] * exp_pen3 + P->ppen4[inxn] * exp_pen4)) / (trm_pen34 * trm_pen34);
        double CEpen[4];
        CEpen[1] = Cf9j / fn9; CEpen[2] = -2.0 * P->ppen2[inxn] * (BOij - 2.0); CEpen[3] = -2.0 * P->ppen2[inxn] * (BOjk - 2.0);
        for (int a = 1; a <= 3; a++) CEpen[a] = CEpen[a] * PEpen;
        double sum_BOi = r->delta[i] + P->Val[ity], sum_BOk = r->delta[k] + P->Val[kty];
        double delta_val = r->delta[j] + P->Val[jty] - P->Valval[jty];
        double exp_coa2 = exp(P->pcoa2[inxn] * delta_val);
        double exp_coa3i = exp(-P->pcoa3[inxn] * ((-BOij + sum_BOi) * (-BOij + sum_BOi)));
        double exp_coa3k = exp(-P->pcoa3[inxn] * ((-BOjk + sum_BOk) * (-BOjk + sum_BOk)));
        double exp_coa4i = exp(-P->pcoa4[inxn] * ((BOij - 1.5) * (BOij - 1.5)));
        double exp_coa4k = exp(-P->pcoa4[inxn] * ((BOjk - 1.5) * (BOjk - 1.5)));
        double PEcoa = P->pcoa1[inxn] / (1.0 + exp_coa2) * exp_coa3i * exp_coa3k * exp_coa4i * exp_coa4k;
        double CEcoa[6];
        CEcoa[1] = -2.0 * P->pcoa4[inxn] * (BOij - 1.5); CEcoa[2] = -2.0 * P->pcoa4[inxn] * (BOjk - 1.5);
        CEcoa[3] = -P->pcoa2[inxn] * exp_coa2 / (1.0 + exp_coa2);
        CEcoa[4] = -2.0 * P->pcoa3[inxn] * (-BOij + sum_BOi); CEcoa[5] = -2.0 * P->pcoa3[inxn] * (-BOjk + sum_BOk);
        for (int a = 1; a <= 5; a++) CEcoa[a] = CEcoa[a] * PEcoa;
        r->PE[5] += PEval; r->PE[6] += PEpen; r->PE[7] += PEcoa;
        double CE3body_b1 = CEpen[2] + CEcoa[1] - CEcoa[4] + CEval[1];
        double CE3body_b2 = CEpen[3] + CEcoa[2] - CEcoa[5] + CEval[2];
        double CE3body_d1 = CEpen[1] + CEcoa[3] + CEval[3] + CEval[7];
        double CE3body_d2 = CEcoa[4], CE3body_d3 = CEcoa[5], CE3body_a = CEval[8];
        ForceB(r, i, NBX(r, j, i1), j, i1, CE3body_b1);
        ForceB(r, j, k1, k, NBX(r, j, k1), CE3body_b2);
        for (int n1 = 1; n1 <= nj; n1++) {
          double c0 = CE3body_d1 + CEval[6] * powi(BOa(r, 0, j, n1), 7);
          double coeff[3] = {c0 + 0.0, c0 + CEval[5], c0 + CEval[5]};
          ForceBbo(r, j, n1, NBR(r, j, n1), NBX(r, j, n1), coeff);
        }
        r->cdbnd[i] += CE3body_d2; r->cdbnd[k] += CE3body_d3;
        ForceA3(r, CE3body_a, i, j, k, rij, rjk);
      }
    }
  }
}

static void Ehb(World *W, Rank *r) { /* pot.F90:559-673 */
  const Params *P = &W->P;
  for (int i = 1; i <= r->NATOMS; i++) {
    int ity = r->ity[i];
    for (int j1 = 1; j1 <= NBR(r, i, 0); j1++) {
      int j = NBR(r, i, j1), jty = r->ity[j];
      if (!(jty == 2 && BOa(r, 0, i, j1) > MINBO0)) continue;   /* hydrogen hard-coded as type 2, :595 */
      for (int kk = 1; kk <= NBP(r, i, 0); kk++) {
        int k = NBP(r, i, kk), kty = r->ity[k];
        int inxnhb = T3(P->inxn3hb, ity, jty, kty);
        if (!(j != k && i != k && inxnhb != 0)) continue;
        double rik[3] = {POS(r, i, 0) - POS(r, k, 0), POS(r, i, 1) - POS(r, k, 1), POS(r, i, 2) - POS(r, k, 2)};
        double rik2 = rik[0] * rik[0] + rik[1] * rik[1] + rik[2] * rik[2];
        if (!(rik2 < rchb2)) continue;
        double rjk[4], rij[4]; vec(r, j, k, rjk); vec(r, i, j, rij);
        double cos_ijk = -(rij[1] * rjk[1] + rij[2] * rjk[2] + rij[3] * rjk[3]) / (rij[0] * rjk[0]);
        if (cos_ijk > MAXANGLE) cos_ijk = MAXANGLE;
        if (cos_ijk < MINANGLE) cos_ijk = MINANGLE;
        double theta_ijk = acos(cos_ijk);
        double sin_ijk_half = sin(0.5 * theta_ijk), sin_xhz4 = powi(sin_ijk_half, 4), cos_xhz1 = (1.0 - cos_ijk);
        double exp_hb2 = exp(-P->phb2[inxnhb] * BOa(r, 0, i, j1));
        double exp_hb3 = exp(-P->phb3[inxnhb] * (P->r0hb[inxnhb] / rjk[0] + rjk[0] / P->r0hb[inxnhb] - 2.0));
        double PEhb = P->phb1[inxnhb] * (1.0 - exp_hb2) * exp_hb3 * sin_xhz4;
        r->PE[10] += PEhb;
        double CEhb1 = P->phb1[inxnhb] * P->phb2[inxnhb] * exp_hb2 * exp_hb3 * sin_xhz4;
        double CEhb2 = -0.5 * P->phb1[inxnhb] * (1.0 - exp_hb2) * exp_hb3 * cos_xhz1;
        double CEhb3 = -PEhb * P->phb3[inxnhb] * (-P->r0hb[inxnhb] / (rjk[0] * rjk[0]) + 1.0 / P->r0hb[inxnhb]) * (1.0 / rjk[0]);
        ForceB(r, i, j1, j, NBX(r, i, j1), CEhb1);
        ForceA3(r, CEhb2, i, j, k, rij, rjk);
        for (int a = 0; a < 3; a++) { double ff = CEhb3 * rjk[a + 1]; FRC(r, j, a) -= ff; FRC(r, k, a) += ff; }
      }
    }
  }
}

static void E4b(World *W, Rank *r) { /* pot.F90:980-1227 */
  const Params *P = &W->P;
  for (int j = 1; j <= r->NATOMS; j++) {
    int jty = r->ity[j], nj = NBR(r, j, 0);
    double delta_ang_j = r->delta[j] + P->Val[jty] - P->Valangle[jty];
    long long jid = r->gid[j];
    for (int k1 = 1; k1 <= nj; k1++) {
      double BOjk = BOa(r, 0, j, k1) - cutof2_esub;
      if (!(BOa(r, 0, j, k1) > cutof2_esub)) continue;
      int k = NBR(r, j, k1);
      if (!(jid < r->gid[k])) continue;
      int kty = r->ity[k];
      double delta_ang_k = r->delta[k] + P->Val[kty] - P->Valangle[kty];
      double delta_ang_jk = delta_ang_j + delta_ang_k;
      double rjk[4]; vec(r, j, k, rjk);
      for (int i1 = 1; i1 <= nj; i1++) {
        double BOij = BOa(r, 0, j, i1) - cutof2_esub;
        if (!(BOa(r, 0, j, i1) > cutof2_esub && BOa(r, 0, j, i1) * BOa(r, 0, j, k1) > cutof2_esub)) continue;
        int i = NBR(r, j, i1);
        if (i == k) continue;
        int ity = r->ity[i];
        double rij[4]; vec(r, i, j, rij);
        double cos_ijk = -(rij[1] * rjk[1] + rij[2] * rjk[2] + rij[3] * rjk[3]) / (rij[0] * rjk[0]);
        if (cos_ijk > MAXANGLE) cos_ijk = MAXANGLE;
        if (cos_ijk < MINANGLE) cos_ijk = MINANGLE;
        double theta_ijk = acos(cos_ijk), sin_ijk = sin(theta_ijk), tan_ijk_i = 1.0 / tan(theta_ijk);
        double crs_ijk[4]; cross_product(rij, rjk, crs_ijk);
        for (int l1 = 1; l1 <= NBR(r, k, 0); l1++) {
          double BOkl = BOa(r, 0, k, l1) - cutof2_esub;
          if (!(BOa(r, 0, k, l1) > cutof2_esub && BOa(r, 0, j, k1) * BOa(r, 0, k, l1) > cutof2_esub)) continue;
          int l = NBR(r, k, l1), lty = r->ity[l];
          int inxn = T4(P->inxn4, ity, jty, kty, lty);
          if (!(inxn != 0 && i != l && j != l)) continue;
          if (!(BOa(r, 0, j, i1) * (BOa(r, 0, j, k1) * BOa(r, 0, j, k1)) * BOa(r, 0, k, l1) > MINBO0)) continue;
          double rkl[4]; vec(r, k, l, rkl);
          double exp_tor2[3] = {exp(-P->ptor2[inxn] * BOij), exp(-P->ptor2[inxn] * BOjk), exp(-P->ptor2[inxn] * BOkl)};
          double exp_tor3 = exp(-P->ptor3[inxn] * delta_ang_jk), exp_tor4 = exp(P->ptor4[inxn] * delta_ang_jk);
          double exp_tor34_i = 1.0 / (1.0 + exp_tor3 + exp_tor4);
          double fn10 = (1.0 - exp_tor2[0]) * (1.0 - exp_tor2[1]) * (1.0 - exp_tor2[2]);
          double fn11 = (2.0 + exp_tor3) / (1.0 + exp_tor3 + exp_tor4);
          double fn12 = exp(-P->pcot2[inxn] * ((BOij - 1.5) * (BOij - 1.5) + (BOjk - 1.5) * (BOjk - 1.5) + (BOkl - 1.5) * (BOkl - 1.5)));
          double btb2 = 2.0 - BOa(r, 2, j, k1) - fn11;
          double exp_tor1 = exp(P->ptor1[inxn] * (btb2 * btb2));
          double cos_jkl = -(rjk[1] * rkl[1] + rjk[2] * rkl[2] + rjk[3] * rkl[3]) / (rjk[0] * rkl[0]);
          if (cos_jkl > MAXANGLE) cos_jkl = MAXANGLE;
          if (cos_jkl < MINANGLE) cos_jkl = MINANGLE;
          double theta_jkl = acos(cos_jkl), sin_jkl = sin(theta_jkl), tan_jkl_i = 1.0 / tan(theta_jkl);
          double crs_jkl[4]; cross_product(rjk, rkl, crs_jkl);
          double c1 = (crs_ijk[1] * crs_jkl[1] + crs_ijk[2] * crs_jkl[2] + crs_ijk[3] * crs_jkl[3]) / (crs_ijk[0] * crs_jkl[0]);
          if (c1 > MAXANGLE) c1 = MAXANGLE;
          if (c1 < MINANGLE) c1 = MINANGLE;
          double omega_ijkl = acos(c1), cos_ijkl_sqr = c1 * c1, cos_2ijkl = cos(2.0 * omega_ijkl);
          double c2 = 1.0 - cos_2ijkl, c3 = 1.0 + cos(3.0 * omega_ijkl);
          double V1 = P->V1[inxn], V2 = P->V2[inxn], V3 = P->V3[inxn];
          double PEtors = 0.5 * fn10 * sin_ijk * sin_jkl * (V1 * (1.0 + c1) + V2 * exp_tor1 * c2 + V3 * c3);
          double PEconj = P->pcot1[inxn] * fn12 * (1.0 + (cos_ijkl_sqr - 1.0) * sin_ijk * sin_jkl);
          r->PE[8] += PEtors; r->PE[9] += PEconj;
          double CEtors[10], CEconj[7];
          CEtors[1] = 0.5 * sin_ijk * sin_jkl * (V1 * (1.0 + c1) + V2 * exp_tor1 * c2 + V3 * c3);
          CEtors[2] = -P->ptor1[inxn] * fn10 * sin_ijk * sin_jkl * V2 * exp_tor1 * btb2 * c2;
          double dfn11 = (-P->ptor3[inxn] * exp_tor3 + (P->ptor3[inxn] * exp_tor3 - P->ptor4[inxn] * exp_tor4) * (2.0 + exp_tor3) * exp_tor34_i) * exp_tor34_i;
          CEtors[3] = CEtors[2] * dfn11;
          CEtors[4] = CEtors[1] * P->ptor2[inxn] * exp_tor2[0] * (1.0 - exp_tor2[1]) * (1.0 - exp_tor2[2]);
          CEtors[5] = CEtors[1] * P->ptor2[inxn] * (1.0 - exp_tor2[0]) * exp_tor2[1] * (1.0 - exp_tor2[2]);
          CEtors[6] = CEtors[1] * P->ptor2[inxn] * (1.0 - exp_tor2[0]) * (1.0 - exp_tor2[1]) * exp_tor2[2];
          double cmn = -0.5 * fn10 * (V1 * (1.0 + c1) + V2 * exp_tor1 * c2 + V3 * c3);
          CEtors[7] = cmn * sin_jkl * tan_ijk_i;
          CEtors[8] = cmn * sin_ijk * tan_jkl_i;
          CEtors[9] = fn10 * sin_ijk * sin_jkl * (0.5 * V1 - 2.0 * V2 * exp_tor1 * c1 + 1.5 * V3 * (cos_2ijkl + 2.0 * cos_ijkl_sqr));
          double Cconj = -2.0 * P->pcot2[inxn] * PEconj;
          CEconj[1] = Cconj * (BOij - 1.5); CEconj[2] = Cconj * (BOjk - 1.5); CEconj[3] = Cconj * (BOkl - 1.5);
          CEconj[4] = -P->pcot1[inxn] * fn12 * (cos_ijkl_sqr - 1.0) * tan_ijk_i * sin_jkl;
          CEconj[5] = -P->pcot1[inxn] * fn12 * (cos_ijkl_sqr - 1.0) * sin_ijk * tan_jkl_i;
          CEconj[6] = 2.0 * P->pcot1[inxn] * fn12 * c1 * sin_ijk * sin_jkl;
          double C4b[3] = {CEconj[1] + CEtors[4], CEconj[2] + CEtors[5], CEconj[3] + CEtors[6]};
          double C4a[3] = {CEconj[4] + CEtors[7], CEconj[5] + CEtors[8], CEconj[6] + CEtors[9]};
          r->cdbnd[j] += CEtors[3]; r->cdbnd[k] += CEtors[3];
          ForceB(r, i, NBX(r, j, i1), j, i1, C4b[0]);
          double C4b_jk[3] = {C4b[1] + 0.0, C4b[1] + CEtors[2], C4b[1] + 0.0};
          ForceBbo(r, j, k1, k, NBX(r, j, k1), C4b_jk);
          ForceB(r, k, l1, l, NBX(r, k, l1), C4b[2]);
          ForceA3(r, C4a[0], i, j, k, rij, rjk);
          ForceA3(r, C4a[1], j, k, l, rjk, rkl);
          ForceA4(r, C4a[2], i, j, k, l, rij, rjk, rkl);
        }
      }
    }
  }
}

static void ForceBondedTerms(Rank *r) { /* pot.F90:113-144 -- index-ordered on purpose */
  for (int i = 1; i <= r->copyptr[6]; i++) {
    ForceD(r, i, r->cdbnd[i]);
    for (int j1 = 1; j1 <= NBR(r, i, 0); j1++) {
      int j = NBR(r, i, j1);
      for (int k = 0; k < 3; k++) { double ff = r->ccbnd[i] * r->dBOp[SL(r, i, j1)] * (POS(r, i, k) - POS(r, j, k)); FRC(r, i, k) -= ff; FRC(r, j, k) += ff; }
    }
    r->ccused[i] = r->ccbnd[i];
    r->ccbnd[i] = 0.0;
  }
}

static int FORCE(World *W) { /* pot.F90:2-90 */
  double dr[3] = {NMINCELL * W->lcsize[0], NMINCELL * W->lcsize[1], NMINCELL * W->lcsize[2]};
  for (int p = 0; p < W->nprocs; p++) {
    Rank *r = &W->R[p];
    for (int i = 0; i < r->NBUFFER; i++) { r->ccbnd[i] = 0.0; r->cdbnd[i] = 0.0; }
    for (size_t i = 0; i < (size_t)3 * r->NBUFFER; i++) r->f[i] = 0.0;
    for (int k = 0; k < 14; k++) r->PE[k] = 0.0;
  }
  if (COPYATOMS(W, MODE_COPY, dr)) return -1;
  for (int p = 0; p < W->nprocs; p++) {
    Rank *r = &W->R[p];
    if (LINKEDLIST(W, r, W->lcsize, r->header, r->llist, r->nacell, W->cc, MAXLAYERS)) { snprintf(W->err, 256, "atom outside the bonded cell grid"); return -1; }
    if (LINKEDLIST(W, r, W->nblcsize, r->nbheader, r->nbllist, r->nbnacell, W->nbcc, MAXLAYERS_NB)) { snprintf(W->err, 256, "atom outside the NB cell grid"); return -1; }
    if (NEIGHBORLIST(W, r, NMINCELL)) return -1;
    if (nb_pairlist(W, r, 0)) return -1;               /* GetNonbondingPairList */
    BOPRIM(W, r); BOFULL(W, r);
    if (W->P.isPQEq) ENbond_PQEq(W, r); else ENbond(W, r);      /* pot.F90:48-52 */
    Ebond(W, r); Elnpr(W, r); Ehb(W, r); E3b(W, r); E4b(W, r);
    if (W->P.isEfield)                                 /* EEfield, module.F90:359-383, pot.F90:61 */
      for (int i = 1; i <= r->NATOMS; i++) {
        double qic = r->q[i] + W->P.Zpqeq[r->ity[i]];
        double Eforce = -qic * W->P.eFieldStrength * Eev_kcal;
        if (W->P.pq_clean) { FRC(r, i, W->P.eFieldDir - 1) += Eforce; continue; }
        /* EEfield declares f(NATOMS,3) but receives f(NBUFFER,3) (module.F90:363 vs pot.F90:61): by sequence association its
         * f(i,dir) is element (dir-1)*NATOMS + i of the caller's storage, i.e. for dir > 1 the x component of a GHOST slot,
         * which the CPBK fold then adds to that ghost's owner.  Restated as it executes. */
        size_t lin = (size_t)(W->P.eFieldDir - 1) * r->NATOMS + (i - 1);          /* 0-based offset in f(NBUFFER,3) of the caller */
        size_t NBf = (size_t)r->NBUFFER;                                          /* the caller's first extent (create the oracle with the reference's NBUFFER, module.F90:80) */
        FRC(r, (int)(lin % NBf) + 1, (int)(lin / NBf)) += Eforce;
      }
    ForceBondedTerms(r);
    for (int i = 1; i < r->NBUFFER; i++) {             /* virial, pot.F90:65-72 */
      r->astr[0] += POS(r, i, 0) * FRC(r, i, 0); r->astr[1] += POS(r, i, 1) * FRC(r, i, 1); r->astr[2] += POS(r, i, 2) * FRC(r, i, 2);
      r->astr[3] += POS(r, i, 1) * FRC(r, i, 2); r->astr[4] += POS(r, i, 2) * FRC(r, i, 0); r->astr[5] += POS(r, i, 0) * FRC(r, i, 1);
    }
  }
  double z[3] = {0, 0, 0};
  return COPYATOMS(W, MODE_CPBK, z);
}

static void LinearMomentum(World *W);
/* one pass of the MD loop body, src/main.F90:64-98 (mdmode 1 = NVE; thermostats are out of the path) */
static int md_step(World *W) {
  const Params *P = &W->P;
  for (int p = 0; p < W->nprocs; p++) {
    Rank *r = &W->R[p];
    for (int i = 1; i <= r->NATOMS; i++) for (int k = 0; k < 3; k++) VEL(r, i, k) = VEL(r, i, k) + 1.0 * W->dthm[r->ity[i]] * FRC(r, i, k); /* vkick, :192-207 */
    for (int i = 1; i <= r->NATOMS; i++) r->qsfv[i] = r->qsfv[i] + 0.5 * W->dt * W->Lex_w2 * (r->q[i] - r->qsfp[i]);
    for (int i = 1; i <= r->NATOMS; i++) r->qsfp[i] = r->qsfp[i] + W->dt * r->qsfv[i];
  }
  if (P->isEfield) LinearMomentum(W);                  /* main.F90:70-71 */
  for (int p = 0; p < W->nprocs; p++) {
    Rank *r = &W->R[p];
    for (int i = 1; i <= r->NATOMS; i++) for (int k = 0; k < 3; k++) POS(r, i, k) = POS(r, i, k) + W->dt * VEL(r, i, k);
  }
  double z[3] = {0, 0, 0};
  if (COPYATOMS(W, MODE_MOVE, z)) return -1;
  if (W->md_nstep % (W->qstep > 0 ? W->qstep : 1) == 0)  /* if(mod(nstep,qstep)==0), main.F90:77-83; nstep counts from 0 */
    if (W->P.isPQEq ? PQEq(W) : QEq(W)) return -1;
  W->md_nstep++;
  if (FORCE(W)) return -1;
  for (int p = 0; p < W->nprocs; p++) {
    Rank *r = &W->R[p];
    for (int i = 1; i <= r->NATOMS; i++) {             /* kinetic stress, :86-94 */
      double m = P->mass[r->ity[i]];
      r->astr[0] += VEL(r, i, 0) * VEL(r, i, 0) * m; r->astr[1] += VEL(r, i, 1) * VEL(r, i, 1) * m; r->astr[2] += VEL(r, i, 2) * VEL(r, i, 2) * m;
      r->astr[3] += VEL(r, i, 1) * VEL(r, i, 2) * m; r->astr[4] += VEL(r, i, 2) * VEL(r, i, 0) * m; r->astr[5] += VEL(r, i, 0) * VEL(r, i, 1) * m;
    }
    for (int i = 1; i <= r->NATOMS; i++) for (int k = 0; k < 3; k++) VEL(r, i, k) = VEL(r, i, k) + 1.0 * W->dthm[r->ity[i]] * FRC(r, i, k);
    for (int i = 1; i <= r->NATOMS; i++) r->qsfv[i] = r->qsfv[i] + 0.5 * W->dt * W->Lex_w2 * (r->q[i] - r->qsfp[i]);
  }
  return 0;
}

/* ------------------------------------------------------------------ velocity scaling of the MD loop head, src/main.F90:45-61 */
static const double UTEMP0 = 503.398008, UTEMP = 503.398008 * 2.0 / 3.0;                 /* module.F90:198-199 */
static void LinearMomentum(World *W) { /* main.F90:766-797 */
  const Params *P = &W->P;
  double sb[4][64];
  for (int p = 0; p < W->nprocs; p++) {
    Rank *r = &W->R[p];
    double mm = 0, v0 = 0, v1 = 0, v2 = 0;
    for (int i = 1; i <= r->NATOMS; i++) { double m = P->mass[r->ity[i]]; v0 += m * VEL(r, i, 0); v1 += m * VEL(r, i, 1); v2 += m * VEL(r, i, 2); mm += m; }
    sb[0][p] = mm; sb[1][p] = v0; sb[2][p] = v1; sb[3][p] = v2;
  }
  double mm = allreduce_sum(sb[0], W->nprocs), vc[3] = {allreduce_sum(sb[1], W->nprocs), allreduce_sum(sb[2], W->nprocs), allreduce_sum(sb[3], W->nprocs)};
  for (int k = 0; k < 3; k++) vc[k] = vc[k] / mm;
  for (int p = 0; p < W->nprocs; p++) { Rank *r = &W->R[p]; for (int i = 1; i <= r->NATOMS; i++) for (int k = 0; k < 3; k++) VEL(r, i, k) = VEL(r, i, k) - vc[k]; }
}
/* what the loop head does when mod(nstep,sstep)==0: mdmode 4 (v *= vsfact), 5 (rescale to treq; gke = kinetic energy per atom of
 * the last PRINTE, main.F90:49), 7 (ScaleTemperature :722-763, per element), 8 (AdjustTemperature :684-719, only beyond 5 %).
 * mdmode 0/6 (INITVELOCITY, random) are not restated. */
int rxo_thermostat(void *w, int mdmode, double treq_K, double vsfact, double gke) {
  World *W = (World *)w; const Params *P = &W->P;
  const double treq = treq_K / UTEMP0;                 /* init.F90:72: the requested temperature is kept in energy units */
  if (mdmode == 4 || mdmode == 5) {
    double c = mdmode == 4 ? vsfact : sqrt((treq * UTEMP0) / (gke * UTEMP));
    for (int p = 0; p < W->nprocs; p++) { Rank *r = &W->R[p]; for (int i = 1; i <= r->NATOMS; i++) for (int k = 0; k < 3; k++) VEL(r, i, k) = c * VEL(r, i, k); }
    return 0;
  }
  if (mdmode == 8) {
    double ea[64];
    for (int p = 0; p < W->nprocs; p++) {
      Rank *r = &W->R[p]; double e = 0;
      for (int i = 1; i <= r->NATOMS; i++) e = e + 0.5 * P->mass[r->ity[i]] * (VEL(r, i, 0) * VEL(r, i, 0) + VEL(r, i, 1) * VEL(r, i, 1) + VEL(r, i, 2) * VEL(r, i, 2));
      ea[p] = e;
    }
    double Ek = allreduce_sum(ea, W->nprocs) / (double)W->GNATOMS;
    double c = sqrt((treq * UTEMP0) / (Ek * UTEMP));
    if (fabs(c - 1.0) > 0.05) {
      for (int p = 0; p < W->nprocs; p++) { Rank *r = &W->R[p]; for (int i = 1; i <= r->NATOMS; i++) for (int k = 0; k < 3; k++) VEL(r, i, k) = c * VEL(r, i, k); }
      LinearMomentum(W);
    }
    return 0;
  }
  if (mdmode == 7) {
    double cnt[21][64], ek[21][64], c[21];
    memset(cnt, 0, sizeof cnt); memset(ek, 0, sizeof ek);
    for (int p = 0; p < W->nprocs; p++) {
      Rank *r = &W->R[p];
      for (int i = 1; i <= r->NATOMS; i++) {
        int t = r->ity[i];
        cnt[t][p] += 1.0;
        ek[t][p] += 0.5 * P->mass[t] * (VEL(r, i, 0) * VEL(r, i, 0) + VEL(r, i, 1) * VEL(r, i, 1) + VEL(r, i, 2) * VEL(r, i, 2));
      }
    }
    for (int t = 1; t <= 20; t++) {
      double n = allreduce_sum(cnt[t], W->nprocs), e = allreduce_sum(ek[t], W->nprocs);
      if (n > 1.0) { c[t] = e / n; c[t] = sqrt((treq * UTEMP0) / (c[t] * UTEMP)); } else c[t] = 0.0;
    }
    for (int p = 0; p < W->nprocs; p++) { Rank *r = &W->R[p]; for (int i = 1; i <= r->NATOMS; i++) for (int k = 0; k < 3; k++) VEL(r, i, k) = c[r->ity[i]] * VEL(r, i, k); }
    LinearMomentum(W);
    return 0;
  }
  return -1;
}

/* ------------------------------------------------------------------ setup (INITSYSTEM, src/init.F90:7-288) + C API for the tests */
static void alloc_rank(World *W, Rank *r, int p, int NBUFFER) {
  memset(r, 0, sizeof(*r));
  r->myid = p; r->NBUFFER = NBUFFER; r->maxn10 = W->maxn10;
  r->vID[0] = p % W->vprocs[0]; r->vID[1] = (p / W->vprocs[0]) % W->vprocs[1]; r->vID[2] = p / (W->vprocs[0] * W->vprocs[1]); /* init.F90:75-80 */
  for (int a = 0; a < 3; a++) r->myparity[a] = r->vID[a] % 2;
  int k = 0;
  for (int i = 0; i < 3; i++) for (int j = 1; j >= -1; j -= 2) {      /* init.F90:82-100 */
    int l[3] = {r->vID[0], r->vID[1], r->vID[2]};
    l[i] = (r->vID[i] + j + W->vprocs[i]) % W->vprocs[i];
    r->target_node[++k] = l[0] + l[1] * W->vprocs[0] + l[2] * W->vprocs[0] * W->vprocs[1];
  }
  size_t NB = NBUFFER;
  r->ity = ialloc(NB); r->gid = (long long *)calloc(NB, sizeof(long long)); r->frcindx = ialloc(NB);
  r->pos = dalloc(3 * NB); r->v = dalloc(3 * NB); r->f = dalloc(3 * NB);
  r->q = dalloc(NB); r->qs = dalloc(NB); r->qt = dalloc(NB); r->gs = dalloc(NB); r->gt = dalloc(NB); r->hs = dalloc(NB); r->ht = dalloc(NB);
  r->qsfp = dalloc(NB); r->qsfv = dalloc(NB);
  r->spos = dalloc(3 * NB); r->fpqeq = dalloc(NB); r->hshs = dalloc(NB); r->hsht = dalloc(NB);   /* init.F90:117-120 */
  r->llist = ialloc(NB); r->nbllist = ialloc(NB);
  r->nbrlist = ialloc(NB * (MAXNEIGHBS + 1)); r->nbrindx = ialloc(NB * (MAXNEIGHBS + 1));
  size_t ns = NB * (MAXNEIGHBS + 1);
  r->BO = dalloc(ns * 4); r->dln_BOp = dalloc(ns * 3); r->dBOp = dalloc(ns); r->A0 = dalloc(ns); r->A1 = dalloc(ns); r->A2 = dalloc(ns); r->A3 = dalloc(ns);
  r->deltap = dalloc(2 * NB); r->delta = dalloc(NB); r->nlp = dalloc(NB); r->dDlp = dalloc(NB); r->deltalp = dalloc(NB); r->ccbnd = dalloc(NB); r->cdbnd = dalloc(NB); r->ccused = dalloc(NB);
  r->commflag = (char *)calloc(NB, 1);
}

/* --lg (cmdline.F90:148-151): a process-wide switch like the reference's module variable; set before rxo_create because it
 * changes the ffield FORMAT (param.F90:83-86,107-109,197-200). */
static int g_isLG = 0;
void rxo_global_lg(int on) { g_isLG = on; }

void *rxo_create(const char *ffield, const double lattice[6], const int vprocs[3], int isQEq, int NMAXQEq, double QEq_tol, double dt_fs, int NBUFFER, int maxn10) {
  World *W = (World *)calloc(1, sizeof(World));
  W->P.isLG = g_isLG;
  if (read_ffield(&W->P, ffield)) { free(W); return NULL; }
  W->lata = lattice[0]; W->latb = lattice[1]; W->latc = lattice[2]; W->lalpha = lattice[3]; W->lbeta = lattice[4]; W->lgamma = lattice[5];
  for (int a = 0; a < 3; a++) W->vprocs[a] = vprocs[a];
  W->nprocs = vprocs[0] * vprocs[1] * vprocs[2];
  W->isQEq = isQEq; W->NMAXQEq = NMAXQEq; W->QEq_tol = QEq_tol; W->maxn10 = maxn10 > 0 ? maxn10 : 1500;
  W->Lex_fqs = 1.0; W->Lex_k = 2.0;
  W->dt = dt_fs / UTIME;                               /* init.F90:66 */
  W->Lex_w2 = 2.0 * W->Lex_k / W->dt / W->dt;          /* init.F90:69 */
  Params *P = &W->P;
  P->rctap = 10.0; P->rctap2 = P->rctap * P->rctap;    /* init.F90:28-34 (no PQEq) */
  double rc = P->rctap;
  P->CTap[0] = 1.0; P->CTap[1] = P->CTap[2] = P->CTap[3] = 0.0;  /* init.F90:36-38 */
  P->CTap[4] = -35.0 / powi(rc, 4); P->CTap[5] = 84.0 / powi(rc, 5); P->CTap[6] = -70.0 / powi(rc, 6); P->CTap[7] = 20.0 / powi(rc, 7);
  get_box(W);
  for (int a = 0; a < 3; a++) W->LBOX[a + 1] = 1.0 / vprocs[a];
  W->dthm = dalloc(P->nso + 1); W->hmas = dalloc(P->nso + 1);
  for (int t = 1; t <= P->nso; t++) { W->dthm[t] = W->dt * 0.5 / P->mass[t]; W->hmas[t] = 0.5 * P->mass[t]; }
  W->R = (Rank *)calloc(W->nprocs, sizeof(Rank));
  for (int p = 0; p < W->nprocs; p++) {
    alloc_rank(W, &W->R[p], p, NBUFFER);
    for (int a = 0; a < 3; a++) W->R[p].OBOX[a + 1] = W->LBOX[a + 1] * W->R[p].vID[a];
  }
  return W;
}

/* --pqeq <path> / rxmd.in PQEqParm (cmdline.F90:112-128,291-293): call between rxo_create and rxo_init.
 * Switches the taper cutoff to rctap0_pqeq and replaces chi / eta (init.F90:28-43). */
int rxo_enable_pqeq(void *w, const char *pqeq_path) {
  World *W = (World *)w; Params *P = &W->P;
  int rc = read_pqeq(P, pqeq_path);
  if (rc) { snprintf(W->err, 256, "cannot read PQEq parameters from %s (%d)", pqeq_path, rc); return rc; }
  /* fewer rows than ffield types is legal (examples/3-reaxpq+): the loops of initialize_pqeq run over ntype_pqeq, module.F90:501-522 */
  P->isPQEq = 1;
  P->rctap = rctap0_pqeq; P->rctap2 = P->rctap * P->rctap;
  double rc_ = P->rctap;
  P->CTap[0] = 1.0; P->CTap[1] = P->CTap[2] = P->CTap[3] = 0.0;
  P->CTap[4] = -35.0 / powi(rc_, 4); P->CTap[5] = 84.0 / powi(rc_, 5); P->CTap[6] = -70.0 / powi(rc_, 6); P->CTap[7] = 20.0 / powi(rc_, 7);
  P->UDR = P->rctap2 / NTABLE; P->UDRi = 1.0 / P->UDR;
  initialize_pqeq(P);
  return 0;
}
/* --efield dir strength [V/A] (cmdline.F90:131-137); dir = 1,2,3 */
void rxo_set_efield(void *w, int dir, double strength) { World *W = (World *)w; W->P.isEfield = 1; W->P.eFieldDir = dir; W->P.eFieldStrength = strength; }
long long rxo_pqeq_stale(void *w) { return ((World *)w)->pq_stale; }
/* clean = 1: a lookup beyond the cutoff yields zero (what the HIP engine does) instead of the previous pair's values */
void rxo_set_pqeq_clean(void *w, int clean) { ((World *)w)->P.pq_clean = clean; }

/* rxff.bin record -> state (ReadBIN, src/fileio.F90:444-555): rnorm are normalised LOCAL coordinates */
int rxo_set_atoms(void *w, int rank, int n, const double *rnorm, const double *v, const double *q, const int *type, const long long *gid) {
  World *W = (World *)w; Rank *r = &W->R[rank];
  if (n >= r->NBUFFER) return -1;
  r->NATOMS = n;
  for (int i = 1; i <= n; i++) {
    double rr[3] = {rnorm[3 * (i - 1)] + r->OBOX[1], rnorm[3 * (i - 1) + 1] + r->OBOX[2], rnorm[3 * (i - 1) + 2] + r->OBOX[3]};
    for (int a = 0; a < 3; a++) POS(r, i, a) = W->HH[a][0] * rr[0] + W->HH[a][1] * rr[1] + W->HH[a][2] * rr[2]; /* xs2xu */
    for (int a = 0; a < 3; a++) VEL(r, i, a) = v ? v[3 * (i - 1) + a] : 0.0;
    r->q[i] = q ? q[i - 1] : 0.0; r->ity[i] = type[i - 1]; r->gid[i] = gid[i - 1];
    r->qsfp[i] = 0.0; r->qsfv[i] = 0.0;
  }
  return 0;
}

int rxo_init(void *w) { /* the rest of INITSYSTEM: cutoffs, cells, tables, 10 A mesh */
  World *W = (World *)w; Params *P = &W->P;
  long long *npt = (long long *)calloc(P->nso + 2, sizeof(long long));
  W->GNATOMS = 0;
  for (int p = 0; p < W->nprocs; p++) { Rank *r = &W->R[p]; W->GNATOMS += r->NATOMS; for (int i = 1; i <= r->NATOMS; i++) npt[r->ity[i]]++; }
  cutofflength(P, npt);
  free(npt);
  double lb[3] = {W->lata / W->vprocs[0], W->latb / W->vprocs[1], W->latc / W->vprocs[2]};
  for (int a = 0; a < 3; a++) { W->cc[a] = (int)(lb[a] / P->maxrc); W->lcsize[a] = W->LBOX[a + 1] / W->cc[a]; } /* UpdateBoxParams, init.F90:636-668 */
  potentialtable(P);
  /* GetNonbondingMesh, init.F90:525-607 */
  double nbl[3]; int imesh[3];
  for (int a = 0; a < 3; a++) { W->nbcc[a] = (int)(lb[a] / 3.0); nbl[a] = lb[a] / W->nbcc[a]; imesh[a] = (int)(P->rctap / nbl[a]) + 1; }
  for (int pass = 0; pass < 2; pass++) {
    int cnt = 0;
    for (int i = -imesh[0]; i <= imesh[0]; i++) for (int j = -imesh[1]; j <= imesh[1]; j++) for (int k = -imesh[2]; k <= imesh[2]; k++) {
      int ii[3] = {i, j, k};
      for (int a = 0; a < 3; a++) { if (ii[a] > 0) ii[a]--; else if (ii[a] < 0) ii[a]++; }
      double rr[3] = {ii[0] * nbl[0], ii[1] * nbl[1], ii[2] * nbl[2]};
      double dr2 = rr[0] * rr[0] + rr[1] * rr[1] + rr[2] * rr[2];
      if (dr2 <= P->rctap * P->rctap) { if (pass) { W->nbmesh[3 * cnt] = i; W->nbmesh[3 * cnt + 1] = j; W->nbmesh[3 * cnt + 2] = k; } cnt++; }
    }
    if (!pass) { W->nbnmesh = cnt; W->nbmesh = ialloc(3 * (size_t)cnt); }
  }
  double lat[3] = {W->lata, W->latb, W->latc};
  for (int a = 0; a < 3; a++) W->nblcsize[a] = nbl[a] / lat[a];
  for (int p = 0; p < W->nprocs; p++) {
    Rank *r = &W->R[p];
    size_t nc = (size_t)(W->cc[0] + 2 * MAXLAYERS) * (W->cc[1] + 2 * MAXLAYERS) * (W->cc[2] + 2 * MAXLAYERS);
    size_t nnb = (size_t)(W->nbcc[0] + 2 * MAXLAYERS_NB) * (W->nbcc[1] + 2 * MAXLAYERS_NB) * (W->nbcc[2] + 2 * MAXLAYERS_NB);
    r->header = ialloc(nc); r->nacell = ialloc(nc); r->nbheader = ialloc(nnb); r->nbnacell = ialloc(nnb);
    size_t rows = (size_t)r->NATOMS + (size_t)(r->NATOMS / 4) + 64;   /* residents may grow by migration */
    if (rows > (size_t)r->NBUFFER) rows = r->NBUFFER;
    r->nbplist = ialloc(rows * (r->maxn10 + 1)); r->hessian = dalloc(rows * (r->maxn10 + 1));
  }
  return 0;
}

int rxo_qeq(void *w) { World *W = (World *)w; return W->P.isPQEq ? PQEq(W) : QEq(W); }   /* main.F90:27-31 */
int rxo_force(void *w) { return FORCE((World *)w); }
/* rxmd.in `QEq <isQEq> <NMAXQEq> <QEq_tol> <qstep>`: charges are re-equilibrated every qstep-th MD step only */
void rxo_set_qstep(void *w, int qstep) { ((World *)w)->qstep = qstep; }
int rxo_step(void *w, int nsteps) { for (int s = 0; s < nsteps; s++) if (md_step((World *)w)) return -1; return 0; }
const char *rxo_error(void *w) { return ((World *)w)->err; }
int rxo_natoms(void *w, int rank) { return ((World *)w)->R[rank].NATOMS; }
int rxo_nghost_total(void *w, int rank) { return ((World *)w)->R[rank].copyptr[6]; }
int rxo_qeq_iters(void *w) { return ((World *)w)->nstep_qeq; }
int rxo_ntrace(void *w) { return ((World *)w)->ntrace; }
void rxo_get_trace(void *w, double *out) { World *W = (World *)w; memcpy(out, W->trace, sizeof(double) * 3 * W->ntrace); }
void rxo_get_info(void *w, double *out) { /* cutoffs & grid, for host-logic tests */
  World *W = (World *)w;
  out[0] = W->P.maxrc; for (int a = 0; a < 3; a++) { out[1 + a] = W->cc[a]; out[4 + a] = W->nbcc[a]; out[7 + a] = W->lcsize[a]; out[10 + a] = W->nblcsize[a]; }
  out[13] = W->nbnmesh; out[14] = W->P.nboty; out[15] = W->P.nso; out[16] = W->dt;
}
void rxo_get_rc(void *w, double *rc) { World *W = (World *)w; for (int x = 1; x <= W->P.nboty; x++) rc[x - 1] = W->P.rc[x]; }
/* tables, [inxn-1][i-1] i=1..NTABLE: which = 0 Evdw, 1 dEvdw, 2 Eclmb, 3 dEclmb, 4 Eclmb_QEq */
void rxo_get_table(void *w, int which, double *out) {
  World *W = (World *)w; Params *P = &W->P;
  for (int x = 1; x <= P->nboty; x++) for (int i = 1; i <= NTABLE; i++) {
    size_t k = (size_t)x * (NTABLE + 2) + i; double v;
    if (which == 0) v = P->TBL_Evdw[2 * k]; else if (which == 1) v = P->TBL_Evdw[2 * k + 1];
    else if (which == 2) v = P->TBL_Eclmb[2 * k]; else if (which == 3) v = P->TBL_Eclmb[2 * k + 1]; else v = P->TBL_Eclmb_QEq[k];
    out[(size_t)(x - 1) * NTABLE + (i - 1)] = v;
  }
}
void rxo_get_energy(void *w, double *pe14) { /* sum over ranks, PE(0) = sum(PE(1:13)) as PRINTE (main.F90:236) */
  World *W = (World *)w;
  for (int k = 0; k < 14; k++) pe14[k] = 0;
  for (int p = 0; p < W->nprocs; p++) for (int k = 1; k < 14; k++) pe14[k] += W->R[p].PE[k];
  for (int k = 1; k < 14; k++) pe14[0] += pe14[k];
}
/* astr(1:6) summed over ranks as PRINTE's allreduce (main.F90:241-246); reset != 0 zeroes the accumulators as PRINTE does (:270).
 * Printed pressure: ss = sum(astr(1:3))/3 / MDBOX * USTRS / pstep, USTRS = 6.94728103 (module.F90:200). */
void rxo_get_astr(void *w, double *out6, int reset) {
  World *W = (World *)w;
  for (int k = 0; k < 6; k++) out6[k] = 0.0;
  for (int p = 0; p < W->nprocs; p++) for (int k = 0; k < 6; k++) { out6[k] += W->R[p].astr[k]; if (reset) W->R[p].astr[k] = 0.0; }
}
double rxo_mdbox(void *w) { return ((World *)w)->MDBOX; }
double rxo_kinetic(void *w) { /* PRINTE, main.F90:225-229 */
  World *W = (World *)w; double KE = 0;
  for (int p = 0; p < W->nprocs; p++) { Rank *r = &W->R[p]; for (int i = 1; i <= r->NATOMS; i++) KE += W->hmas[r->ity[i]] * (VEL(r, i, 0) * VEL(r, i, 0) + VEL(r, i, 1) * VEL(r, i, 1) + VEL(r, i, 2) * VEL(r, i, 2)); }
  return KE;
}
/* per-atom arrays of residents (n) or residents+ghosts (what >= 100): 0 pos(3) 1 v(3) 2 f(3) 3 q 4 type 5 gid 6 qs 7 qt
 * 100 pos incl ghosts, 101 delta, 102 deltap1, 103 nbr count, 104 n10 count, 105 type incl ghosts, 106 gid incl ghosts, 107 cdbnd? (after force: zero) */
int rxo_get(void *w, int rank, int what, double *out) {
  World *W = (World *)w; Rank *r = &W->R[rank];
  int n = r->NATOMS, G = r->copyptr[6];
  switch (what) {
    case 0: for (int i = 1; i <= n; i++) for (int k = 0; k < 3; k++) out[3 * (i - 1) + k] = POS(r, i, k); return n;
    case 1: for (int i = 1; i <= n; i++) for (int k = 0; k < 3; k++) out[3 * (i - 1) + k] = VEL(r, i, k); return n;
    case 2: for (int i = 1; i <= n; i++) for (int k = 0; k < 3; k++) out[3 * (i - 1) + k] = FRC(r, i, k); return n;
    case 3: for (int i = 1; i <= n; i++) out[i - 1] = r->q[i]; return n;
    case 4: for (int i = 1; i <= n; i++) out[i - 1] = r->ity[i]; return n;
    case 5: for (int i = 1; i <= n; i++) out[i - 1] = (double)r->gid[i]; return n;
    case 6: for (int i = 1; i <= n; i++) out[i - 1] = r->qs[i]; return n;
    case 7: for (int i = 1; i <= n; i++) out[i - 1] = r->qt[i]; return n;
    case 8: for (int i = 1; i <= n; i++) for (int k = 0; k < 3; k++) out[3 * (i - 1) + k] = SPOS(r, i, k); return n;
    case 9: for (int i = 1; i <= n; i++) out[i - 1] = r->fpqeq[i]; return n;
    case 100: for (int i = 1; i <= G; i++) for (int k = 0; k < 3; k++) out[3 * (i - 1) + k] = POS(r, i, k); return G;
    case 101: for (int i = 1; i <= G; i++) out[i - 1] = r->delta[i]; return G;
    case 102: for (int i = 1; i <= G; i++) out[i - 1] = r->deltap[2 * i]; return G;
    case 103: for (int i = 1; i <= G; i++) out[i - 1] = NBR(r, i, 0); return G;
    case 104: for (int i = 1; i <= n; i++) out[i - 1] = NBP(r, i, 0); return n;
    case 105: for (int i = 1; i <= G; i++) out[i - 1] = r->ity[i]; return G;
    case 106: for (int i = 1; i <= G; i++) out[i - 1] = (double)r->gid[i]; return G;
    case 109: for (int i = 1; i <= G; i++) out[i - 1] = r->ccused[i]; return G;   /* ccbnd(i) at the moment pot.F90:129-135 uses it */
    case 110: for (int i = 1; i <= G; i++) out[i - 1] = r->cdbnd[i]; return G;    /* cdbnd(i) after the energy terms */
    case 108: for (int i = 1; i <= n; i++) { double s = 0; for (int k = 1; k <= NBP(r, i, 0); k++) s += HES(r, i, k); out[i - 1] = s; } return n;
  }
  return -1;
}
/* bonded neighbour table of atom range [1..G]: nbr index (1-based, 0 = empty) and BO(0), dense [G][MAXNEIGHBS] */
int rxo_get_bonds(void *w, int rank, int *nbr, double *bo0) {
  World *W = (World *)w; Rank *r = &W->R[rank]; int G = r->copyptr[6];
  for (int i = 1; i <= G; i++) for (int k = 1; k <= MAXNEIGHBS; k++) {
    size_t o = (size_t)(i - 1) * MAXNEIGHBS + (k - 1);
    if (k <= NBR(r, i, 0)) { nbr[o] = NBR(r, i, k); bo0[o] = BOa(r, 0, i, k); } else { nbr[o] = 0; bo0[o] = 0; }
  }
  return G;
}
/* qsfp / qsfv of a restart file (rxff.bin record columns 9-10, fileio.F90:536-537) */
void rxo_set_lex(void *w, int rank, const double *qsfp, const double *qsfv) { World *W = (World *)w; Rank *r = &W->R[rank]; for (int i = 1; i <= r->NATOMS; i++) { r->qsfp[i] = qsfp[i - 1]; r->qsfv[i] = qsfv[i - 1]; } }
void rxo_set_charges(void *w, int rank, const double *q) { World *W = (World *)w; Rank *r = &W->R[rank]; for (int i = 1; i <= r->NATOMS; i++) r->q[i] = q[i - 1]; }
void rxo_set_qeq(void *w, int isQEq, int NMAXQEq, double tol) { World *W = (World *)w; W->isQEq = isQEq; W->NMAXQEq = NMAXQEq; W->QEq_tol = tol; }
void rxo_destroy(void *w) { (void)w; /* test processes are short-lived; leak on purpose */ }
