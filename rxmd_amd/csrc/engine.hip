// engine.hip -- construction, device memory, atom I/O, ghost-atom build (COPYATOMS MODE_COPY),
// migration (MODE_MOVE), halo refresh (MODE_QCOPY1/2), ghost-force fold (MODE_CPBK), cell binning.
// Reference behaviour restated MI355X-first: src/comm.F90 (the six-direction staged exchange),
// src/main.F90:277-318 (LINKEDLIST), src/init.F90:7-288 (INITSYSTEM derived quantities).
#include "engine.h"
#include <cstdlib>

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstring>

namespace rxmd {

static const double UTIME = 1e3 / 20.455;  // reference src/module.F90:202
static const int NMINCELL = 4;             // reference src/module.F90:84
static const int cptridx_[7] = {0, 0, 0, 2, 2, 4, 4};  // comm.F90:61
static const int dinv_[7] = {0, 2, 1, 4, 3, 6, 5};     // comm.F90:60

// RXMD_POISON_ALLOC=1 (diagnostic; changes nothing that is computed): every buffer starts as 0xFF bytes -- a NaN for doubles, -1 for
// indices and counts -- instead of zeros, and the per-step scratch (ghost slots of the per-atom arrays, bonded tables, the 10 A list and its
// window form) is filled with the pattern again before every rebuild (Engine::poison_step_scratch), so that a kernel which reads an element
// nobody wrote this step shows as a NaN / an index trap instead of silently using a stale or zero value.  The reference's own allocator does not
// clear (module.F90:732-744); what it clears explicitly -- ccbnd, cdbnd, f, PE per FORCE call (pot.F90:20-26), spos and qtfp/qtfv at
// allocation (init.F90:117-131) -- the kernels here clear too.  Buffers whose ZERO is part of a protocol (arrival counters, error words,
// device scalars) are allocated with dzalloc.
static const Options g_opt = Options::from_env();      // (allocation helpers are free functions: the library-wide copy of the switches)
static const bool g_poison = g_opt.poison_alloc;
#ifdef RXMD_EXPERIMENTS
// RXMD_CONTIG_ALLOC=<bytes>: buffers of at most that many bytes (0: every buffer) come from hipExtMallocWithFlags(hipDeviceMallocContiguous) -- the
// configuration that failed 15 unrelated tests in round 3 (NOTES.md 3); with RXMD_POISON_ALLOC=1 a read of stale memory shows as a NaN
static const long long g_contig = g_opt.contig_alloc;
#endif
template <class T>
static void dmalloc(T *&p, size_t n) {
  const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
#ifdef RXMD_EXPERIMENTS
  if (g_contig == 0 || (g_contig > 0 && bytes <= static_cast<size_t>(g_contig))) RX_HIP(hipExtMallocWithFlags(reinterpret_cast<void **>(&p), bytes, hipDeviceMallocContiguous));
  else
#endif
  RX_HIP(hipMalloc(reinterpret_cast<void **>(&p), bytes));
  RX_HIP(hipMemset(p, g_poison ? 0xFF : 0, bytes));
}
template <class T>
static void dzalloc(T *&p, size_t n) {
  RX_HIP(hipMalloc(reinterpret_cast<void **>(&p), std::max<size_t>(n, 1) * sizeof(T)));
  RX_HIP(hipMemset(p, 0, std::max<size_t>(n, 1) * sizeof(T)));
}
template <class T>
static void dfree(T *&p) { if (p) { (void)hipFree(p); p = nullptr; } }

static void make_box(Box &b, const double lat[6], const int vprocs[3], const int vID[3]) {
  // GetBoxParams (reference src/init.F90:610-633)
  const double pi = std::atan(1.0) * 4.0;
  const double la = lat[0], lb = lat[1], lc = lat[2];
  const double lal = lat[3] * pi / 180.0, lbe = lat[4] * pi / 180.0, lga = lat[5] * pi / 180.0;
  const double hh1 = lc * (std::cos(lal) - std::cos(lbe) * std::cos(lga)) / std::sin(lga);
  const double hh2 = lc * std::sqrt(1.0 - std::cos(lal) * std::cos(lal) - std::cos(lbe) * std::cos(lbe) - std::cos(lga) * std::cos(lga) +
                                    2 * std::cos(lal) * std::cos(lbe) * std::cos(lga)) / std::sin(lga);
  double(*H)[3] = b.H;
  H[0][0] = la; H[1][0] = 0; H[2][0] = 0;
  H[0][1] = lb * std::cos(lga); H[1][1] = lb * std::sin(lga); H[2][1] = 0;
  H[0][2] = lc * std::cos(lbe); H[1][2] = hh1; H[2][2] = hh2;
  double(*m)[3] = b.Hi;  // matinv (src/main.F90:557-579)
  m[0][0] = H[1][1] * H[2][2] - H[1][2] * H[2][1]; m[0][1] = H[0][2] * H[2][1] - H[0][1] * H[2][2]; m[0][2] = H[0][1] * H[1][2] - H[0][2] * H[1][1];
  m[1][0] = H[1][2] * H[2][0] - H[1][0] * H[2][2]; m[1][1] = H[0][0] * H[2][2] - H[0][2] * H[2][0]; m[1][2] = H[0][2] * H[1][0] - H[0][0] * H[1][2];
  m[2][0] = H[1][0] * H[2][1] - H[1][1] * H[2][0]; m[2][1] = H[0][1] * H[2][0] - H[0][0] * H[2][1]; m[2][2] = H[0][0] * H[1][1] - H[0][1] * H[1][0];
  const double det = H[0][0] * H[1][1] * H[2][2] + H[0][1] * H[1][2] * H[2][0] + H[0][2] * H[1][0] * H[2][1] -
                     H[0][2] * H[1][1] * H[2][0] - H[0][1] * H[1][0] * H[2][2] - H[0][0] * H[1][2] * H[2][1];
  for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) m[a][c] = m[a][c] / det;
  b.volume = det;
  for (int a = 0; a < 6; ++a) b.lat[a] = lat[a];
  for (int a = 0; a < 3; ++a) { b.lbox[a] = 1.0 / vprocs[a]; b.obox[a] = b.lbox[a] * vID[a]; }
}

Engine::Engine(const rxmd_config &c) : cfg(c) {
  if (!c.ffield_path) throw EngineError(RXMD_E_ARG, "ffield_path is NULL");
  ffield_path = c.ffield_path;
  cfg.ffield_path = ffield_path.c_str();
  for (int a = 0; a < 3; ++a) if (cfg.vprocs[a] < 1) throw EngineError(RXMD_E_ARG, "vprocs must be >= 1");
  nprocs = cfg.vprocs[0] * cfg.vprocs[1] * cfg.vprocs[2];
  if (cfg.myid < 0 || cfg.myid >= nprocs) throw EngineError(RXMD_E_ARG, "myid outside the vprocs grid");
  {   // a box the H matrix of GetBoxParams (init.F90:610-633) cannot describe: non-positive edge, angle outside (0,180), zero volume
    const double *L = cfg.lattice, d2r = std::atan(1.0) / 45.0;
    const double ca = std::cos(L[3] * d2r), cb = std::cos(L[4] * d2r), cg = std::cos(L[5] * d2r);
    const double v2 = 1.0 - ca * ca - cb * cb - cg * cg + 2.0 * ca * cb * cg;
    bool ok = L[0] > 0.0 && L[1] > 0.0 && L[2] > 0.0 && v2 > 1e-12;
    for (int a = 3; a < 6; ++a) ok = ok && L[a] > 0.0 && L[a] < 180.0;
    if (!ok) throw EngineError(RXMD_E_ARG, "lattice does not span a box (edges must be positive, angles inside (0,180) and not coplanar)");
  }
  try { ff.parse(ffield_path, cfg.lg != 0); } catch (const std::exception &e) { throw EngineError(RXMD_E_FFIELD, e.what()); }
  // rank grid, reference src/init.F90:74-100
  vID[0] = cfg.myid % cfg.vprocs[0]; vID[1] = (cfg.myid / cfg.vprocs[0]) % cfg.vprocs[1]; vID[2] = cfg.myid / (cfg.vprocs[0] * cfg.vprocs[1]);
  int k = 0;
  for (int i = 0; i < 3; ++i)
    for (int j = 1; j >= -1; j -= 2) {
      int l[3] = {vID[0], vID[1], vID[2]};
      l[i] = (vID[i] + j + cfg.vprocs[i]) % cfg.vprocs[i];
      target_node[++k] = l[0] + l[1] * cfg.vprocs[0] + l[2] * cfg.vprocs[0] * cfg.vprocs[1];
    }
  make_box(box, cfg.lattice, cfg.vprocs, vID);
  dt = cfg.dt_fs / UTIME;                         // init.F90:66
  Lex_w2 = 2.0 * cfg.Lex_k / dt / dt;             // init.F90:69
  dthm.assign(ff.nso + 1, 0.0); hmas.assign(ff.nso + 1, 0.0);
  for (int t = 1; t <= ff.nso; ++t) { dthm[t] = dt * 0.5 / ff.atom[t].mass; hmas[t] = 0.5 * ff.atom[t].mass; }  // init.F90:105-108
  if (cfg.pqeq_path && cfg.pqeq_path[0]) {          // --pqeq: chi/eta replaced, taper cutoff 12.5 A (init.F90:28-43, module.F90:282)
    pqeq_path = cfg.pqeq_path; cfg.pqeq_path = pqeq_path.c_str();
    try { ff.parse_pqeq(pqeq_path); } catch (const std::exception &e) { throw EngineError(RXMD_E_FFIELD, e.what()); }
    ff.build_taper(12.5);
  } else {
    cfg.pqeq_path = nullptr;
    ff.build_taper(10.0);                         // rctap0, module.F90:281
  }
  if (cfg.efield_dir != 0 && (!ff.pqeq || cfg.efield_dir < 1 || cfg.efield_dir > 3))
    throw EngineError(RXMD_E_ARG, "efield needs a PQEq parameter file (core charges Z) and a direction 1..3");
  stage_pairs = !opt.no_stage_pairs;
  force_staged = opt.force_staged; force_remote = opt.force_remote;
  spin_wait = opt.spin_wait != 0;
  halo_direct = opt.halo_direct;
  comm_timeout_s = opt.comm_timeout_s;
  MAXNB = cfg.maxneighbs > 0 ? cfg.maxneighbs : 30;
  if (MAXNB > 31) throw EngineError(RXMD_E_ARG, "maxneighbs must be <= 31 (the wavefront-per-centre kernels stage the bond slots of two atoms in one 64-lane wavefront; the reference uses 30)");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) throw EngineError(RXMD_E_HIP, "no HIP device visible: this engine has no CPU path");
  RX_HIP(hipSetDevice(cfg.device));
  { hipDeviceProp_t pr; RX_HIP(hipGetDeviceProperties(&pr, cfg.device)); num_cu = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
    // the angle kernel and the window kernels ask for up to 78 KB of LDS per workgroup (gfx950: 160 KB per CU, 64 KB on every earlier arch): a
    // device that cannot give it is refused here -- a launch that fails later would leave energy terms out without a word
    if (pr.sharedMemPerBlock < size_t(80) * 1024) throw EngineError(RXMD_E_HIP, "device " + std::to_string(cfg.device) + " offers " + std::to_string(pr.sharedMemPerBlock) + " bytes of LDS per workgroup; the kernels are written for gfx950 (MI355X, 160 KB per CU) and need 80 KB"); }
  RX_HIP(hipStreamCreate(&stream));
  for (int k = 0; k < 64; ++k) { KtPair p; RX_HIP(hipEventCreate(&p.a)); RX_HIP(hipEventCreate(&p.b)); kt_free.push_back(p); }
  if (opt.single_stream) comm_stream = stream;   // diagnostic: the halo work queues on the main stream (no second hardware queue)
  else {                                         // highest priority: pack / send-recv / unpack kernels of a halo go ahead of the queued compute workgroups
    int lo = 0, hi = 0;
    RX_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    RX_HIP(hipStreamCreateWithPriority(&comm_stream, hipStreamDefault, hi));
  }
  if (!opt.single_stream && !opt.no_bond_overlap && opt.bond_overlap != 0) {   // the charge-free part of FORCE next to ENbond (engine.h: bond_stream); lowest priority.  Opt-in since round 6: with the torsion kernel at 2 ms it buys nothing any more (profiles/r06_ab_bond_overlap_after_e4b.txt)
    int lo = 0, hi = 0;
    RX_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    RX_HIP(hipStreamCreateWithPriority(&bond_stream, hipStreamDefault, lo));
  }
  RX_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming)); RX_HIP(hipEventCreateWithFlags(&ev_bond, hipEventDisableTiming));
  RX_HIP(hipEventCreateWithFlags(&ev_main, hipEventDisableTiming)); RX_HIP(hipEventCreateWithFlags(&ev_comm, hipEventDisableTiming));
  RX_HIP(hipEventCreateWithFlags(&ev_est, hipEventDisableTiming));
  RX_HIP(hipEventCreateWithFlags(&ev_spec[0], hipEventDisableTiming)); RX_HIP(hipEventCreateWithFlags(&ev_spec[1], hipEventDisableTiming));
  RX_HIP(hipEventCreateWithFlags(&ev_upd[0], hipEventDisableTiming)); RX_HIP(hipEventCreateWithFlags(&ev_upd[1], hipEventDisableTiming));
  for (auto &pr : ev_pass) for (auto &e2 : pr) RX_HIP(hipEventCreate(&e2));
  for (auto &e : ev) RX_HIP(hipEventCreate(&e));
}

Engine::~Engine() {
  rccl_destroy();
  free_device();
  for (auto &e : ev) if (e) (void)hipEventDestroy(e);
  for (auto *v : {&kt_free, &kt_pending}) for (auto &p : *v) { if (p.a) (void)hipEventDestroy(p.a); if (p.b) (void)hipEventDestroy(p.b); }
  if (ev_fork) (void)hipEventDestroy(ev_fork);
  if (ev_bond) (void)hipEventDestroy(ev_bond);
  if (bond_stream) (void)hipStreamDestroy(bond_stream);
  if (ev_main) (void)hipEventDestroy(ev_main);
  if (ev_comm) (void)hipEventDestroy(ev_comm);
  if (ev_est) (void)hipEventDestroy(ev_est);
  for (auto &e2 : ev_spec) if (e2) (void)hipEventDestroy(e2);
  for (auto &e2 : ev_upd) if (e2) (void)hipEventDestroy(e2);
  for (auto &pr : ev_pass) for (auto &e2 : pr) if (e2) (void)hipEventDestroy(e2);
  if (comm_stream && comm_stream != stream) (void)hipStreamDestroy(comm_stream);
  if (stream) (void)hipStreamDestroy(stream);
}

void Engine::allreduce_host(double *buf, int n) {
  if (nccl && n <= 8) {                       // the usual case: a scalar or two -- the tail of the device scalar block is the staging area
    double *d = scal + 64;
    RX_HIP(hipMemcpyAsync(d, buf, sizeof(double) * n, hipMemcpyHostToDevice, stream));
    rccl_allreduce_dev(d, n);
    RX_HIP(hipMemcpyAsync(buf, d, sizeof(double) * n, hipMemcpyDeviceToHost, stream));
    sync_stream();
    return;
  }
  if (nccl) {
    double *d = nullptr;
    RX_HIP(hipMalloc(reinterpret_cast<void **>(&d), sizeof(double) * n));
    RX_HIP(hipMemcpyAsync(d, buf, sizeof(double) * n, hipMemcpyHostToDevice, stream));
    rccl_allreduce_dev(d, n);
    RX_HIP(hipMemcpyAsync(buf, d, sizeof(double) * n, hipMemcpyDeviceToHost, stream));
    sync_stream();
    (void)hipFree(d);
    return;
  }
  if (!has_comm || !comm.allreduce_sum) throw EngineError(RXMD_E_COMM, "vprocs > 1 needs a transport: call rxmd_hip_set_comm or rxmd_hip_comm_init_rccl first");
  if (comm.allreduce_sum(comm.ctx, buf, n)) throw EngineError(RXMD_E_COMM, "allreduce callback failed");
}

void Engine::collect_timers() {
  size_t keep = 0;
  for (size_t i = 0; i < kt_pending.size(); ++i) {
    KtPair p = kt_pending[i];
    float ms = 0;
    if (hipEventQuery(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      if (p.dst) *p.dst += ms * p.scale;
      if (p.dst2) *p.dst2 += ms * p.scale;
      if (p.cnt) *p.cnt += 1;
      p.dst = p.dst2 = nullptr; p.cnt = nullptr;
      kt_free.push_back(p);
    } else kt_pending[keep++] = p;
  }
  kt_pending.resize(keep);
}

void Engine::fetch_device_error() {
  RX_HIP(hipMemcpyAsync(h_err, d_err, sizeof(int) * 16, hipMemcpyDeviceToHost, stream));
  sync_stream();
}
void Engine::check_device_error(const char *where, bool fetch) {
  if (fetch) fetch_device_error();
  const int e = h_err[0];
  if (e == DERR_NONE) return;
  RX_HIP(hipMemsetAsync(d_err, 0, sizeof(int) * 2, stream));   // [2] (longest bond list of the build) stays: a retry rebuilds the 10 A list only
  const std::string w = std::string(where) + ": ";
  if (e == DERR_MAXNB) throw EngineError(RXMD_E_MAXNEIGHBS, w + "overflow of max # in neighbor list (MAXNEIGHBS=" + std::to_string(MAXNB) + ", needed " + std::to_string(h_err[1]) + ")");
  if (e == DERR_MAXN10) throw EngineError(RXMD_E_MAXNEIGHBS10, w + "nbplist greater than MAXNEIGHBS10=" + std::to_string(S10) + " (needed " + std::to_string(h_err[1]) + ")");
  if (e == DERR_NBRINDX) throw EngineError(RXMD_E_STATE, w + "inconsistency between nbrlist and nbrindx");
  if (e == DERR_TYPE) throw EngineError(RXMD_E_ARG, w + "atom type outside the ffield");
  throw EngineError(RXMD_E_STATE, w + "device error " + std::to_string(e));
}

// ---------------------------------------------------------------------------------------------
// derived quantities that need the atoms (INITSYSTEM after ReadBIN, init.F90:141-213)
void Engine::setup_after_atoms(const std::vector<long long> &npt) {
  if (ff.pqeq)
    for (int t = ff.npq + 1; t <= ff.nso; ++t)
      if (npt[t] > 0) throw EngineError(RXMD_E_FFIELD, "PQEq: atoms of ffield type " + std::to_string(t) + " (" + ff.atom[t].name + ") have no row in the PQEq parameter file");
  ff.compute_cutoffs(npt);
  ff.build_tables();
  if (ff.pqeq) ff.build_pqeq_tables();
  const double lreal[3] = {box.lat[0] / cfg.vprocs[0], box.lat[1] / cfg.vprocs[1], box.lat[2] / cfg.vprocs[2]};
  for (int a = 0; a < 3; ++a) {
    cc[a] = static_cast<int>(lreal[a] / ff.maxrc);            // UpdateBoxParams, init.F90:656
    if (cc[a] < 1) throw EngineError(RXMD_E_ARG, "local box smaller than the bond cutoff");
    shell[a] = NMINCELL * (box.lbox[a] / cc[a]);               // dr of the FORCE ghost copy, pot.F90:28
  }
  // the engine's own grid: one cell >= max(rctap/2, maxrc) wide, measured PERPENDICULAR to its faces (the planes of constant normalised
  // coordinate a are 1 / |row a of Hi| apart per unit; with 90-degree angles that is the lattice constant); stencil +-2 covers the taper
  // cutoff, +-1 the bonds
  const double cw = std::max(0.5 * ff.rctap, ff.maxrc) * (1.0 + 1e-9);
  grid.ortho = (std::fabs(box.lat[3] - 90.0) < 1e-9 && std::fabs(box.lat[4] - 90.0) < 1e-9 && std::fabs(box.lat[5] - 90.0) < 1e-9) ? 1 : 0;
  grid.ncell = 1;
  for (int a = 0; a < 3; ++a) {
    grid.wid[a] = grid.ortho ? box.lat[a] : 1.0 / std::sqrt(box.Hi[a][0] * box.Hi[a][0] + box.Hi[a][1] * box.Hi[a][1] + box.Hi[a][2] * box.Hi[a][2]);
    const double wn = box.lbox[a] + 2.0 * shell[a];
    const double wreal = wn * grid.wid[a];
    grid.n[a] = std::max(1, static_cast<int>(wreal / cw));
    grid.org[a] = -shell[a];
    grid.inv[a] = grid.n[a] / wn;
    grid.ncell *= grid.n[a];
    grid.cw[a] = 1.0 / grid.inv[a];
  }
  grid.iwz = 1.0 / grid.wid[2];
  // the reference's meshes (RefMesh, engine.h)
  for (int a = 0; a < 3; ++a) {
    for (int c = 0; c < 3; ++c) rmesh.Hi[3 * a + c] = box.Hi[a][c];
    rmesh.obox[a] = box.obox[a];
    rmesh.lc[a] = box.lbox[a] / cc[a];                          // lcsize, init.F90:661
    const int nbcc = std::max(1, static_cast<int>(lreal[a] / 3.0));   // nblcsize = 3 A initial estimate, init.F90:538-545
    rmesh.nblr[a] = lreal[a] / nbcc;
    rmesh.nbl[a] = rmesh.nblr[a] / box.lat[a];                  // init.F90:605
    rmesh.qlo[a] = -ff.rctap / box.lat[a];                      // QCopyDr, qeq.F90:32
    rmesh.qhi[a] = box.lbox[a] + ff.rctap / box.lat[a];
  }
  // z-slices per cell: ~1/8 of a cell (0.6 A at the 5 A cell of a 10 A cutoff); bounded so that the slice ids fit the 31-bit sort key
  grid.fz = 8;
  while (grid.fz > 1 && static_cast<long long>(grid.ncell) * grid.fz > (1LL << 28)) grid.fz >>= 1;
  grid.nzf = grid.n[2] * grid.fz;
  grid.nfine = grid.n[0] * grid.n[1] * grid.nzf;
  tables_ready = true;
}

void Engine::upload_ff() {
  // one blob: atom | bond | angle | tors | hb | inxn2 | inxn3 | inxn3hb | inxn4 | tabNB | tabQEq
  const int n1 = ff.n1();
  std::vector<DevAtomP> a(ff.nso + 1);
  for (int t = 1; t <= ff.nso; ++t) {
    const auto &s = ff.atom[t];
    a[t] = {s.Val, s.Valboc, s.mass, s.Vale, s.nlpopt, s.plp2, s.povun2, s.povun5, s.pval3, s.pval5, s.Valangle, s.Valval, s.chi, s.eta};
  }
  std::vector<DevBondP> b(ff.nboty + 1);
  for (int r = 1; r <= ff.nboty; ++r) {
    const auto &s = ff.bond[r];
    b[r] = {s.Desig, s.Depi, s.Depipi, s.pbe1, s.pbe2, s.povun1, s.ovc, s.v13cor, s.pbo2, s.pbo4, s.pbo6, s.pboc3, s.pboc4, s.pboc5,
            s.cBOp1, s.cBOp3, s.cBOp5, s.pbo2h, s.pbo4h, s.pbo6h, s.sw[0], s.sw[1], s.sw[2], s.rc2};
  }
  std::vector<DevAngleP> an(ff.nvaty + 1);
  for (int r = 1; r <= ff.nvaty; ++r) { const auto &s = ff.angle[r]; an[r] = {s.theta00, s.pval1, s.pval2, s.pcoa1, s.pval7, s.ppen1, s.pval4}; }
  std::vector<DevTorsP> to(ff.ntoty + 1);
  for (int r = 1; r <= ff.ntoty; ++r) { const auto &s = ff.tors[r]; to[r] = {s.V1, s.V2, s.V3, s.ptor1, s.pcot1}; }
  std::vector<DevHbP> hb(ff.nhbty + 1);
  for (int r = 1; r <= ff.nhbty; ++r) { const auto &s = ff.hb[r]; hb[r] = {s.r0hb, s.phb1, s.phb2, s.phb3}; }
  const size_t stride = NTABLE + 2;
  std::vector<DevNBTab> nb((ff.nboty + 1) * stride);
  for (size_t i = 0; i + 1 < nb.size(); ++i)
    nb[i] = {ff.tblEvdw[i], ff.tblEvdw[i + 1] - ff.tblEvdw[i], ff.tbldEvdw[i], ff.tbldEvdw[i + 1] - ff.tbldEvdw[i],
             ff.tblEclmb[i], ff.tblEclmb[i + 1] - ff.tblEclmb[i], ff.tbldEclmb[i], ff.tbldEclmb[i + 1] - ff.tbldEclmb[i]};
  nb.back() = DevNBTab{};

  // PQEq tables as (E, dE, F, dF) nodes
  std::vector<double4> pt[3];
  std::vector<double> zk;
  if (ff.pqeq) {
    const std::vector<double> *T[3] = {&ff.tblPcc, &ff.tblPsc, &ff.tblPss};
    for (int k = 0; k < 3; ++k) {
      const size_t nn = T[k]->size() / 2;
      pt[k].assign(nn, make_double4(0, 0, 0, 0));
      for (size_t i = 0; i + 1 < nn; ++i)
        pt[k][i] = make_double4((*T[k])[2 * i], (*T[k])[2 * (i + 1)] - (*T[k])[2 * i], (*T[k])[2 * i + 1], (*T[k])[2 * (i + 1) + 1] - (*T[k])[2 * i + 1]);
    }
    zk.assign(2 * (ff.npq + 1), 0.0);
    for (int t = 1; t <= ff.npq; ++t) { zk[t] = ff.Zpq[t]; zk[ff.npq + 1 + t] = ff.Kspq[t]; }
  }
  // inxn4 followed by "is there a torsion row" as 4096 bits (ffields with at most 7 atom types)
  std::vector<int> inxn4x(ff.inxn4);
  inxn4x.resize(ff.inxn4.size() + 128, 0);
  if (n1 <= 8)
    for (size_t i = 0; i < ff.inxn4.size(); ++i)
      if (ff.inxn4[i] != 0) inxn4x[ff.inxn4.size() + (i >> 5)] |= static_cast<int>(1u << (i & 31));
  // the QEq table once more as pairs (T[i], T[i+1]): the interpolation of a list entry is ONE 16-byte load instead of two 8-byte gathers
  std::vector<double2> tq2(ff.tblQEq.size());
  for (size_t i = 0; i < tq2.size(); ++i) tq2[i] = make_double2(ff.tblQEq[i], i + 1 < tq2.size() ? ff.tblQEq[i + 1] : 0.0);
  auto al = [](size_t x) { return (x + 255) & ~size_t(255); };
  size_t off[13], tot = 0;
  const size_t sz[12] = {a.size() * sizeof(DevAtomP), b.size() * sizeof(DevBondP), an.size() * sizeof(DevAngleP), to.size() * sizeof(DevTorsP),
                         hb.size() * sizeof(DevHbP), ff.inxn2.size() * 4, ff.inxn3.size() * 4, ff.inxn3hb.size() * 4, inxn4x.size() * 4,
                         nb.size() * sizeof(DevNBTab), ff.tblQEq.size() * 8, tq2.size() * sizeof(double2)};
  const void *src[12] = {a.data(), b.data(), an.data(), to.data(), hb.data(), ff.inxn2.data(), ff.inxn3.data(), ff.inxn3hb.data(), inxn4x.data(), nb.data(), ff.tblQEq.data(), tq2.data()};
  for (int i = 0; i < 12; ++i) { off[i] = tot; tot += al(sz[i]); }
  if (ffblob) { (void)hipFree(ffblob); ffblob = nullptr; }
  RX_HIP(hipMalloc(&ffblob, tot));
  for (int i = 0; i < 12; ++i) RX_HIP(hipMemcpy(static_cast<char *>(ffblob) + off[i], src[i], sz[i], hipMemcpyHostToDevice));
  char *base = static_cast<char *>(ffblob);
  dff.nso = ff.nso; dff.n1 = n1; dff.nboty = ff.nboty; dff.ntoty = ff.ntoty; dff.nvaty = ff.nvaty;
  dff.atom = reinterpret_cast<DevAtomP *>(base + off[0]); dff.bond = reinterpret_cast<DevBondP *>(base + off[1]);
  dff.angle = reinterpret_cast<DevAngleP *>(base + off[2]); dff.tors = reinterpret_cast<DevTorsP *>(base + off[3]);
  dff.hb = reinterpret_cast<DevHbP *>(base + off[4]);
  dff.inxn2 = reinterpret_cast<int *>(base + off[5]); dff.inxn3 = reinterpret_cast<int *>(base + off[6]);
  dff.inxn3hb = reinterpret_cast<int *>(base + off[7]); dff.inxn4 = reinterpret_cast<int *>(base + off[8]);
  dff.tor_bits = reinterpret_cast<unsigned *>(base + off[8]) + ff.inxn4.size();
  dff.tabNB = reinterpret_cast<DevNBTab *>(base + off[9]); dff.tabQEq = reinterpret_cast<double *>(base + off[10]); dff.tabQEq2 = reinterpret_cast<double2 *>(base + off[11]);
  dff.rctap_pad = ff.rctap + 1e-6;
  dff.UDR = ff.UDR; dff.UDRi = ff.UDRi; dff.rctap2 = ff.rctap2; dff.cutoff_vpar30 = ff.cutoff_vpar30; dff.vpar1 = ff.vpar1; dff.vpar2 = ff.vpar2;
  dff.plp1 = ff.plp1; dff.povun3 = ff.povun3; dff.povun4 = ff.povun4; dff.povun6 = ff.povun6; dff.povun7 = ff.povun7; dff.povun8 = ff.povun8;
  dff.pval6 = ff.pval6; dff.pval8 = ff.pval8; dff.pval9 = ff.pval9; dff.pval10 = ff.pval10; dff.ppen2 = ff.ppen2; dff.ppen3 = ff.ppen3; dff.ppen4 = ff.ppen4;
  dff.pcoa2 = ff.pcoa2; dff.pcoa3 = ff.pcoa3; dff.pcoa4 = ff.pcoa4; dff.ptor2 = ff.ptor2; dff.ptor3 = ff.ptor3; dff.ptor4 = ff.ptor4; dff.pcot2 = ff.pcot2;
  dff.pqeq = ff.pqeq ? 1 : 0; dff.npq1 = ff.npq + 1;
  ehb_donor_types = 0u;                            // types X for which some hydrogen-bond row (X, 2, k) exists (hydrogen = type 2, pot.F90:595)
  if (ff.nso >= 2)
    for (int t = 1; t <= ff.nso && t < 32; ++t)
      for (int k = 1; k <= ff.nso; ++k)
        if (ff.inxn3hb[(t * n1 + 2) * n1 + k] != 0) ehb_donor_types |= 1u << t;
  if (ff.pqeq) {
    const size_t b0 = al(zk.size() * 8), b1 = al(ff.inxnpq.size() * 4), bt = al(pt[0].size() * sizeof(double4));
    if (pqblob) { (void)hipFree(pqblob); pqblob = nullptr; }
    RX_HIP(hipMalloc(&pqblob, b0 + b1 + 3 * bt));
    char *pb = static_cast<char *>(pqblob);
    RX_HIP(hipMemcpy(pb, zk.data(), zk.size() * 8, hipMemcpyHostToDevice));
    RX_HIP(hipMemcpy(pb + b0, ff.inxnpq.data(), ff.inxnpq.size() * 4, hipMemcpyHostToDevice));
    for (int k = 0; k < 3; ++k) RX_HIP(hipMemcpy(pb + b0 + b1 + k * bt, pt[k].data(), pt[k].size() * sizeof(double4), hipMemcpyHostToDevice));
    dff.Zpq = reinterpret_cast<double *>(pb); dff.Kspq = dff.Zpq + (ff.npq + 1);
    dff.inxnpq = reinterpret_cast<int *>(pb + b0);
    dff.tabPcc = reinterpret_cast<double4 *>(pb + b0 + b1); dff.tabPsc = reinterpret_cast<double4 *>(pb + b0 + b1 + bt); dff.tabPss = reinterpret_cast<double4 *>(pb + b0 + b1 + 2 * bt);
  }
}

void Engine::alloc_device() {
  const size_t nb = NB, ns = static_cast<size_t>(NB) * 32;      // staging of the bonded sweep: one 128-byte line of 32 slots per atom (lists.hip, BL_STRIDE; MAXNB <= 31)
  for (int a = 0; a < 3; ++a) { dmalloc(pos[a], nb); dmalloc(vel[a], nb); dmalloc(frc[a], nb); dmalloc(spos[a], nb); }
  dmalloc(q, nb); dmalloc(qsfp, nb); dmalloc(qsfv, nb); dmalloc(type, nb); dmalloc(gid, nb);
  if (ff.pqeq) {
    for (int a = 0; a < 3; ++a) dmalloc(shl[a], nb);
    dmalloc(sorted_shl, nb); dmalloc(hsc, static_cast<size_t>(rows10) * S10); dmalloc(pqrow, static_cast<size_t>(rows10));
  }
  dmalloc(qst, nb); dmalloc(hst, nb); dmalloc(gst, nb); dmalloc(hst2, nb); dzalloc(tickets, 16);
  { dmalloc(sall, static_cast<size_t>(rows10)); dmalloc(sgh, static_cast<size_t>(rows10)); dmalloc(wall, static_cast<size_t>(rows10)); dmalloc(wgh, static_cast<size_t>(rows10)); }
  dmalloc(gsrc, nb); dmalloc(groot, nb); dmalloc(gowner, nb); dmalloc(dh_ghost, nb); dmalloc(dh_keys, nb); dmalloc(dh_keys2, nb); dmalloc(dh_vals, nb); dmalloc(dh_off, 1100); dmalloc(sendidx, nb); dmalloc(rootperm, nb); dmalloc(invpos, nb); dmalloc(xs, nb); for (int a = 0; a < 3; ++a) { dmalloc(fnb[a], nb); dmalloc(fsort[a], nb); }
  dmalloc(cellid, nb); dmalloc(cellid_sorted, nb); dmalloc(perm, nb); dmalloc(perm_in, nb); dmalloc(cellstart, static_cast<size_t>(grid.nfine) + 2);
  dmalloc(sorted_xyzi, nb); dmalloc(sorted_type, nb); dmalloc(flags, nb + 1); dmalloc(scanout, nb + 1); dmalloc(flags2, nb + 1); dmalloc(scanout2, nb + 1);
  dmalloc(nbr_sm, ns); dmalloc(nbrcnt, nb + 1); dmalloc(boff, nb + 2);
  {   // 5.3 bonds per RDX atom, ~16 in SiC: 12 per atom slot to start with, grown when a build needs more (build_ghosts_and_lists).
      // RXMD_BOND_CAP=<bonds>: start smaller (the tests walk the growth path with it; a capacity, not a result)
    size_t cap = std::min<size_t>(ns, nb * 12);
    if (opt.bond_cap > 0) cap = static_cast<size_t>(opt.bond_cap);
    alloc_bond_tables(cap);
  }
  ehb_don_cap = static_cast<size_t>(rows10) + 64 * 256 + 256; dmalloc(ehb_don, ehb_don_cap); dzalloc(ehb_cnt, 72);   // 64 sub-lists (bonded.hip EHB_REGIONS) + debug words
  dmalloc(ecoef, 6 * nb); dmalloc(deltap, nb); dmalloc(delta, nb); dmalloc(nlp, nb); dmalloc(dDlp, nb); dmalloc(deltalp, nb); dmalloc(cds, nb); dmalloc(cd, nb); dmalloc(cc_, nb);
  dmalloc(nb10, static_cast<size_t>(rows10) * S10);
  dmalloc(rows_int, static_cast<size_t>(rows10)); dmalloc(rows_bnd, static_cast<size_t>(rows10));
  dmalloc(hess, static_cast<size_t>(rows10) * S10); dmalloc(n10, static_cast<size_t>(rows10));
  { const size_t ng = win_groups_bound(rows10) + 1;       // groups never straddle a cell column of the grid: up to one short group per column
    for (double2 **pp : {&r_qst, &r_hst, &r_hst2, &r_gst, &r_sall, &r_sgh, &r_wall, &r_wgh}) dmalloc(*pp, ng * WIN_ROWS);
    dmalloc(r_type, ng * WIN_ROWS); dmalloc(r_n10, ng * WIN_ROWS); dmalloc(r_xpos, ng * WIN_ROWS); dmalloc(rpos, nb); dmalloc(g_rrow, nb);
    dmalloc(rows_sorted, ng * WIN_ROWS); dmalloc(rowcols, ng * WIN_ROWS * 64); dmalloc(grp_base, ng * 32); dmalloc(win_flag, ng + 1); dmalloc(win_k, ng * WIN_MAXUNITS); dmalloc(win_cnt, ng); dmalloc(win_gint, ng); dmalloc(win_gbnd, ng); dmalloc(sl10, static_cast<size_t>(rows10) * S10); }
  partials_cap = std::max<size_t>(size_t(1) << 16, 4 * static_cast<size_t>(rows10) + 16384);   // up to one workgroup (4 partial sums) per row
  dmalloc(partials, partials_cap + 1024); dzalloc(scal, 192);   // + the 128 x 4 first-level sums of k_reduce_fused, behind the per-workgroup partials at a fixed offset
  RX_HIP(hipHostMalloc(reinterpret_cast<void **>(&h_scal), 320 * sizeof(double), hipHostMallocCoherent | hipHostMallocMapped));   // coherent: the update kernel's tail stores the CG snapshot into it and the host polls it      // [0,64): as before; [64,192): the two slots of the run-ahead CG loop (qeq.hip); [192,288): the per-type sums of a host transport
  dzalloc(tsum, 128); { double *sa_ = nullptr; dzalloc(sa_, 32); sargs = reinterpret_cast<ScaleArgs *>(sa_); }
  dzalloc(d_err, 16);
  RX_HIP(hipHostMalloc(reinterpret_cast<void **>(&h_err), 32 * sizeof(int)));
  h_cnt = h_err + 16;
  // hipcub scratch sized for the largest scan / sort we issue
  size_t b1 = 0, b2 = 0;
  hipcub::DeviceScan::ExclusiveSum(nullptr, b1, flags, scanout, NB + 1, stream);
  hipcub::DeviceRadixSort::SortPairs(nullptr, b2, cellid, cellid_sorted, perm_in, perm, NB, 0, 32, stream);
  cubtmp_bytes = std::max(b1, b2) + 256;
  RX_HIP(hipMalloc(&cubtmp, cubtmp_bytes));
}

void Engine::alloc_bond_tables(size_t cap) {
  bcap = std::max<size_t>(cap, 1024);
  dmalloc(nbr, bcap); dmalloc(brev, bcap); dmalloc(bown, bcap); dmalloc(btype, bcap);
  for (double **t : {&bo0, &bo1, &bo2, &bo3, &dln2, &dln3, &dBOp, &A0, &A1, &A2, &A3, &cf1, &cf2, &cf3, &cdn, &fnx, &fny, &fnz, &etor, &econ, &epen, &ecoa, &bt1, &bt2, &bt3}) dmalloc(*t, bcap);
}
void Engine::alloc_e4b_delivery(size_t entries) {
  dfree(e4b_t); dfree(e4b_flag);
  e4b_cap = (std::max<size_t>(entries, 1024) + 3) & ~static_cast<size_t>(3);
  dmalloc(e4b_t, e4b_cap); dzalloc(e4b_flag, e4b_cap);
  e4b_dirty = false;
}
void Engine::free_bond_tables() {
  dfree(nbr); dfree(brev); dfree(bown); dfree(btype);
  for (double **t : {&bo0, &bo1, &bo2, &bo3, &dln2, &dln3, &dBOp, &A0, &A1, &A2, &A3, &cf1, &cf2, &cf3, &cdn, &fnx, &fny, &fnz, &etor, &econ, &epen, &ecoa, &bt1, &bt2, &bt3}) dfree(*t);
  bcap = 0;
}

void Engine::free_device() {
  for (int a = 0; a < 3; ++a) { dfree(flags2); dfree(scanout2); dfree(pos[a]); dfree(vel[a]); dfree(frc[a]); dfree(spos[a]); }
  for (int a = 0; a < 3; ++a) dfree(shl[a]);
  dfree(sorted_shl); dfree(hsc); dfree(pqrow);
  if (pqblob) { (void)hipFree(pqblob); pqblob = nullptr; }
  dfree(q); dfree(qsfp); dfree(qsfv); dfree(type); dfree(gid); dfree(qst); dfree(hst); dfree(gst); dfree(hst2); dfree(tickets); dfree(sall); dfree(sgh); dfree(wall); dfree(wgh);
  dfree(gowner); dfree(dh_ghost); dfree(dh_keys); dfree(dh_keys2); dfree(dh_vals); dfree(dh_off); dfree(dh_serve);
  dfree(gsrc); dfree(groot); dfree(sendidx); dfree(rootperm); dfree(invpos); dfree(xs); for (int a = 0; a < 3; ++a) { dfree(fnb[a]); dfree(fsort[a]); } dfree(cellid); dfree(cellid_sorted); dfree(perm); dfree(perm_in); dfree(cellstart);
  dfree(sorted_xyzi); dfree(sorted_type); dfree(flags); dfree(scanout); dfree(nbr_sm); dfree(nbrcnt); dfree(boff);
  free_bond_tables();
  dfree(ehb_don); dfree(ehb_cnt); dfree(e4b_t); dfree(e4b_flag); e4b_cap = 0;
  dfree(ecoef); dfree(deltap); dfree(delta); dfree(nlp); dfree(dDlp); dfree(deltalp); dfree(cds); dfree(cd); dfree(cc_);
  for (double2 **pp : {&r_qst, &r_hst, &r_hst2, &r_gst, &r_sall, &r_sgh, &r_wall, &r_wgh}) dfree(*pp);
  dfree(r_type); dfree(r_n10); dfree(r_xpos); dfree(rpos); dfree(g_rrow);
  dfree(rows_int); dfree(rows_bnd); dfree(rows_sorted); dfree(rowcols); dfree(grp_base); dfree(win_flag); dfree(win_k); dfree(win_cnt); dfree(win_gint); dfree(win_gbnd); dfree(sl10);
  dfree(nb10); dfree(hess); dfree(n10); dfree(partials); dfree(scal); dfree(d_err); dfree(tsum); { double *sa_ = reinterpret_cast<double *>(sargs); dfree(sa_); sargs = nullptr; }
  if (xbuf_owned) { dfree(xbuf_send); dfree(xbuf_recv); }
  if (h_scal) { (void)hipHostFree(h_scal); h_scal = nullptr; }
  if (h_err) { (void)hipHostFree(h_err); h_err = nullptr; }
  dfree(seg_cnt); dfree(seg_tot); dfree(seg_code_); seg_blocks_cap = 0;
  if (h_seg) { (void)hipHostFree(h_seg); h_seg = nullptr; }
  if (h_pub) { (void)hipHostFree(h_pub); h_pub = nullptr; }
  if (cubtmp) { (void)hipFree(cubtmp); cubtmp = nullptr; }
  if (ffblob) { (void)hipFree(ffblob); ffblob = nullptr; }
}

// ReadBIN (reference src/fileio.F90:528-552): records -> real coordinates, split atype
void Engine::set_atoms_rxff(int natoms, const double *rec) {
  if (natoms < 0 || (natoms == 0 && nprocs == 1)) throw EngineError(RXMD_E_ARG, "natoms must be positive");   // a rank of a decomposed box may own nothing
  std::vector<double> hx[3], hv[3], hq(natoms), hp(natoms), hw(natoms);
  std::vector<int> ht(natoms);
  std::vector<long long> hg(natoms);
  for (int a = 0; a < 3; ++a) { hx[a].resize(natoms); hv[a].resize(natoms); }
  std::vector<long long> npt(ff.nso + 2, 0);
  int n14 = 0;
  for (int i = 0; i < natoms; ++i) {
    const double *r = rec + 10 * static_cast<size_t>(i);
    const double s[3] = {r[0] + box.obox[0], r[1] + box.obox[1], r[2] + box.obox[2]};
    for (int a = 0; a < 3; ++a) { hx[a][i] = box.H[a][0] * s[0] + box.H[a][1] * s[1] + box.H[a][2] * s[2]; hv[a][i] = r[3 + a]; }  // xs2xu
    hq[i] = r[6];
    const int t = static_cast<int>(std::lround(r[7]));
    if (t < 1 || t > ff.nso) throw EngineError(RXMD_E_ARG, "atom type outside the ffield");
    ht[i] = t; hg[i] = std::llround((r[7] - t) * 1e13);                 // l2g, main.F90:582-593
    hp[i] = r[8]; hw[i] = r[9];
    npt[t]++;
    if (r[7] == (static_cast<double>(t) + static_cast<double>(hg[i]) * 1e-13) + 1e-14) ++n14;
  }
  atype_resid = (natoms > 0 && n14 == natoms) ? 1e-14 : 0.0;
  if (!tables_ready) {
    if (nprocs > 1) {
      std::vector<double> tmp(ff.nso + 1);
      for (int t = 0; t <= ff.nso; ++t) tmp[t] = static_cast<double>(npt[t]);
      allreduce_host(tmp.data(), ff.nso + 1);
      for (int t = 0; t <= ff.nso; ++t) npt[t] = std::llround(tmp[t]);
    }
    setup_after_atoms(npt);
    // capacities: sized from this rank's atoms, but never below the mean share (a sparse or empty domain can fill up by migration)
    long long ntot = 0;
    for (int t = 1; t <= ff.nso; ++t) ntot += npt[t];
    const long long nsize = std::max<long long>(natoms, ntot / nprocs);
    double fac = 1.0;
    for (int a = 0; a < 3; ++a) fac *= 1.0 + 2.0 * std::min(shell[a] / box.lbox[a], 1.0);
    const long long want = static_cast<long long>(nsize * fac * 1.12) + 4096;
    NB = cfg.nbuffer > 0 ? cfg.nbuffer : static_cast<int>(std::min<long long>(want, 2000000000LL));
    if (NB <= natoms) throw EngineError(RXMD_E_NBUFFER, "nbuffer smaller than natoms");
    const double vloc = box.volume / nprocs;
    const double est10 = nsize / vloc * (4.0 / 3.0) * 3.14159265358979 * ff.rctap * ff.rctap2;
    int s10 = cfg.maxneighbs10 > 0 ? cfg.maxneighbs10 : static_cast<int>(est10 * 1.35 + 64);
    if (opt.s10 > 0 && cfg.maxneighbs10 <= 0) s10 = static_cast<int>(opt.s10);   // (experiments build: row stride of the 10 A list)
    S10 = (s10 + 63) / 64 * 64;
    rows10 = std::min<long long>(NB, nsize + nsize / 8 + 1024);
    if (ff.nso > 15) throw EngineError(RXMD_E_ARG, "more than 15 atom types do not fit the packed 10 A list entry");
    if (NB >= (1 << NB10_IDX_BITS)) throw EngineError(RXMD_E_NBUFFER, "more than 2^26 atoms+ghosts per GPU do not fit the packed 10 A list entry");
    alloc_device();
    upload_ff();
    st.n10_stride = S10; st.nbuffer = NB;
    for (int a = 0; a < 3; ++a) { st.cells10[a] = grid.n[a]; st.cells3[a] = cc[a]; }
  }
  if (natoms > rows10) throw EngineError(RXMD_E_NBUFFER, "more atoms than the engine was sized for");
  N = natoms; G = natoms;
  for (int a = 0; a < 3; ++a) {
    RX_HIP(hipMemcpy(pos[a], hx[a].data(), sizeof(double) * natoms, hipMemcpyHostToDevice));
    RX_HIP(hipMemcpy(vel[a], hv[a].data(), sizeof(double) * natoms, hipMemcpyHostToDevice));
    RX_HIP(hipMemset(frc[a], 0, sizeof(double) * NB));
  }
  if (ff.pqeq) for (int a = 0; a < 3; ++a) RX_HIP(hipMemset(shl[a], 0, sizeof(double) * NB));      // spos(:,:)=0, init.F90:117-120
  RX_HIP(hipMemcpy(q, hq.data(), sizeof(double) * natoms, hipMemcpyHostToDevice));
  RX_HIP(hipMemcpy(qsfp, hp.data(), sizeof(double) * natoms, hipMemcpyHostToDevice));
  RX_HIP(hipMemcpy(qsfv, hw.data(), sizeof(double) * natoms, hipMemcpyHostToDevice));
  RX_HIP(hipMemcpy(type, ht.data(), sizeof(int) * natoms, hipMemcpyHostToDevice));
  RX_HIP(hipMemcpy(gid, hg.data(), sizeof(long long) * natoms, hipMemcpyHostToDevice));
  atoms_set = true; lists_valid = false; ghosts_valid = false;
  st.natoms = N;
}

// atype = type + l2g * 1e-13 (main.F90:582-593) -> type, global id; a type outside the ffield raises the device error word
__global__ void k_split_atype(int n, int nso, const double *__restrict__ atype, int *__restrict__ type, long long *__restrict__ gid, int *err) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double a = atype[i];
  const int t = static_cast<int>(llround(a));
  if (t < 1 || t > nso) { atomicCAS(&err[0], DERR_NONE, DERR_TYPE); type[i] = 1; gid[i] = 0; return; }
  type[i] = t; gid[i] = llround((a - t) * 1e13);
}
void Engine::set_atoms_arrays(int natoms, const double *atype, const double *x, const double *y, const double *z, const double *qh, const double *lexp, const double *lexv) {
  if (!tables_ready) throw EngineError(RXMD_E_STATE, "set_atoms_arrays before the engine was sized");
  if (natoms < 0 || natoms > rows10 || natoms >= NB) throw EngineError(RXMD_E_NBUFFER, "more atoms than the engine was sized for");
  sync_stream();                                   // nothing of the previous call may still read what is overwritten
  const double *xyz[3] = {x, y, z};
  const size_t nbytes = sizeof(double) * static_cast<size_t>(natoms);
  if (natoms > 0) {
    RX_HIP(hipMemcpyAsync(cds, atype, nbytes, hipMemcpyHostToDevice, stream));     // cds: per-atom scratch of FORCE, free between calls
    k_split_atype<<<(natoms + 255) / 256, 256, 0, stream>>>(natoms, ff.nso, cds, type, gid, d_err);
    for (int a = 0; a < 3; ++a) {
      RX_HIP(hipMemcpyAsync(pos[a], xyz[a], nbytes, hipMemcpyHostToDevice, stream));
      RX_HIP(hipMemsetAsync(vel[a], 0, nbytes, stream));
    }
    if (qh) RX_HIP(hipMemcpyAsync(q, qh, nbytes, hipMemcpyHostToDevice, stream)); else RX_HIP(hipMemsetAsync(q, 0, nbytes, stream));
    if (lexp && lexv) { RX_HIP(hipMemcpyAsync(qsfp, lexp, nbytes, hipMemcpyHostToDevice, stream)); RX_HIP(hipMemcpyAsync(qsfv, lexv, nbytes, hipMemcpyHostToDevice, stream)); }
    else { RX_HIP(hipMemsetAsync(qsfp, 0, nbytes, stream)); RX_HIP(hipMemsetAsync(qsfv, 0, nbytes, stream)); }
  }
  for (int a = 0; a < 3; ++a) RX_HIP(hipMemsetAsync(frc[a], 0, sizeof(double) * NB, stream));
  if (ff.pqeq) for (int a = 0; a < 3; ++a) RX_HIP(hipMemsetAsync(shl[a], 0, sizeof(double) * NB, stream));
  N = natoms; G = natoms;
  atoms_set = true; lists_valid = false; ghosts_valid = false;
  atype_resid = 0.0;
  st.natoms = N;
  check_device_error("set_atoms_arrays");          // (synchronises: the caller's arrays are free again)
}

int Engine::get_atoms_rxff(double *rec, int capacity) {
  if (!atoms_set) throw EngineError(RXMD_E_STATE, "atoms were never set");
  if (!rec) return N;
  if (capacity < N) throw EngineError(RXMD_E_ARG, "capacity smaller than natoms");
  std::vector<double> hx[3], hv[3], hq(N), hp(N), hw(N);
  std::vector<int> ht(N);
  std::vector<long long> hg(N);
  sync_stream();
  for (int a = 0; a < 3; ++a) {
    hx[a].resize(N); hv[a].resize(N);
    RX_HIP(hipMemcpy(hx[a].data(), pos[a], sizeof(double) * N, hipMemcpyDeviceToHost));
    RX_HIP(hipMemcpy(hv[a].data(), vel[a], sizeof(double) * N, hipMemcpyDeviceToHost));
  }
  RX_HIP(hipMemcpy(hq.data(), q, sizeof(double) * N, hipMemcpyDeviceToHost));
  RX_HIP(hipMemcpy(hp.data(), qsfp, sizeof(double) * N, hipMemcpyDeviceToHost));
  RX_HIP(hipMemcpy(hw.data(), qsfv, sizeof(double) * N, hipMemcpyDeviceToHost));
  RX_HIP(hipMemcpy(ht.data(), type, sizeof(int) * N, hipMemcpyDeviceToHost));
  RX_HIP(hipMemcpy(hg.data(), gid, sizeof(long long) * N, hipMemcpyDeviceToHost));
  for (int i = 0; i < N; ++i) {
    double *r = rec + 10 * static_cast<size_t>(i);
    for (int a = 0; a < 3; ++a) r[a] = (box.Hi[a][0] * hx[0][i] + box.Hi[a][1] * hx[1][i] + box.Hi[a][2] * hx[2][i]) - box.obox[a];  // xu2xs
    for (int a = 0; a < 3; ++a) r[3 + a] = hv[a][i];
    r[6] = hq[i]; r[7] = (static_cast<double>(ht[i]) + static_cast<double>(hg[i]) * 1e-13) + atype_resid; r[8] = hp[i]; r[9] = hw[i];
  }
  return N;
}

// ---------------------------------------------------------------------------------------------
// kernels: coordinates, slab flags, append, halo
struct BoxDev { double H[9], Hi[9], obox[3], lbox[3]; };
static BoxDev boxdev(const Box &b) {
  BoxDev d;
  for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) { d.H[3 * a + c] = b.H[a][c]; d.Hi[3 * a + c] = b.Hi[a][c]; }
  for (int a = 0; a < 3; ++a) { d.obox[a] = b.obox[a]; d.lbox[a] = b.lbox[a]; }
  return d;
}

// xu2xs (reference src/main.F90:596-616): normalised local coordinates of atoms [i0,i1)
__global__ void k_to_normalised(BoxDev B, int i0, int i1, const double *x, const double *y, const double *z, double *sx, double *sy, double *sz) {
  const int i = i0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= i1) return;
  const double r0 = x[i], r1 = y[i], r2 = z[i];
  sx[i] = (B.Hi[0] * r0 + B.Hi[1] * r1 + B.Hi[2] * r2) - B.obox[0];
  sy[i] = (B.Hi[3] * r0 + B.Hi[4] * r1 + B.Hi[5] * r2) - B.obox[1];
  sz[i] = (B.Hi[6] * r0 + B.Hi[7] * r1 + B.Hi[8] * r2) - B.obox[2];
}
// xs2xu (reference src/main.F90:641-660)
__global__ void k_to_real(BoxDev B, int i0, int i1, const double *sx, const double *sy, const double *sz, double *x, double *y, double *z) {
  const int i = i0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= i1) return;
  const double r0 = sx[i] + B.obox[0], r1 = sy[i] + B.obox[1], r2 = sz[i] + B.obox[2];
  x[i] = B.H[0] * r0 + B.H[1] * r1 + B.H[2] * r2;
  y[i] = B.H[3] * r0 + B.H[4] * r1 + B.H[5] * r2;
  z[i] = B.H[6] * r0 + B.H[7] * r1 + B.H[8] * r2;
}

// inBuffer (reference src/comm.F90:551-576) over the scan range of one exchange stage
__global__ void k_slab_flags(int nscan, int dflag, double lbox, double dr, const double *s, const int *type, int skip_dead, int *flags) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n > nscan) return;
  int f = 0;
  if (n < nscan) {
    const double rr = s[n];
    f = (dflag & 1) ? (lbox - dr < rr) : (rr <= dr);
    if (skip_dead && type[n] <= 0) f = 0;
  }
  flags[n] = f;   // flags[nscan] = 0 so that scanout[nscan] is the total
}

// store_atoms + append_atoms for a self-exchange stage of MODE_COPY (reference src/comm.F90:367-453,456-528)
__global__ void k_append_ghosts(int nscan, int base, int N, int axis, double sft, const int *flags, const int *scanout,
                                double *sx, double *sy, double *sz, int *type, long long *gid, double *q, int *gsrc, int *groot, int *sendlist) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= nscan || !flags[n]) return;
  const int k = scanout[n], m = base + k;
  double a = sx[n], b = sy[n], c = sz[n];
  if (axis == 0) a += sft; else if (axis == 1) b += sft; else c += sft;
  sx[m] = a; sy[m] = b; sz[m] = c;
  type[m] = type[n]; gid[m] = gid[n]; q[m] = q[n];
  gsrc[m] = n; groot[m] = (n < N) ? n : groot[n];
  sendlist[k] = n;
}

__global__ void k_refresh2(int N, int G, const int *groot, double2 *v) {
  const int g = N + blockIdx.x * blockDim.x + threadIdx.x;
  if (g < G) v[g] = v[groot[g]];
}
__global__ void k_refresh1(int N, int G, const int *groot, double *v) {
  const int g = N + blockIdx.x * blockDim.x + threadIdx.x;
  if (g < G) v[g] = v[groot[g]];
}
// append_atoms for MODE_CPBK (reference src/comm.F90:474-482): one stage, sources are unique within a stage
__global__ void k_fold_stage(int g0, int g1, const int *gsrc, double *fx, double *fy, double *fz) {
  const int m = g0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= g1) return;
  const int n = gsrc[m];
  fx[n] += fx[m]; fy[n] += fy[m]; fz[n] += fz[m];
}

static inline int nblk(long long n, int b) { return n > 0 ? static_cast<int>((n + b - 1) / b) : 1; }   // an empty rank still launches (kernels guard their range)

void Engine::ghost_build() {
  if (multi()) { ghost_build_staged(); return; }
  if (stage_pairs) { ghost_build_fused(); return; }          // single rank: three kernels, one host wait (RXMD_NO_STAGE_PAIRS=1: the staged form below)
  const BoxDev B = boxdev(box);
  k_to_normalised<<<nblk(N, 256), 256, 0, stream>>>(B, 0, N, pos[0], pos[1], pos[2], spos[0], spos[1], spos[2]);
  copyptr[0] = N;
  sendoff[1] = 0;
  if (stage_pairs) {      // both stages of an axis scan the same atoms: one host wait per axis instead of one per stage
    for (int d0 = 1; d0 <= 5; d0 += 2) {
      const int d1 = d0 + 1, nscan = copyptr[cptridx_[d0]], axis = (d0 - 1) / 2;
      k_slab_flags<<<nblk(nscan + 1, 256), 256, 0, stream>>>(nscan, d0, box.lbox[axis], shell[axis], spos[axis], type, 0, flags);
      k_slab_flags<<<nblk(nscan + 1, 256), 256, 0, stream>>>(nscan, d1, box.lbox[axis], shell[axis], spos[axis], type, 0, flags2);
      size_t tb = cubtmp_bytes;
      RX_HIP(hipcub::DeviceScan::ExclusiveSum(cubtmp, tb, flags, scanout, nscan + 1, stream));
      tb = cubtmp_bytes;
      RX_HIP(hipcub::DeviceScan::ExclusiveSum(cubtmp, tb, flags2, scanout2, nscan + 1, stream));
      int t0 = 0, t1 = 0;
      RX_HIP(hipMemcpyAsync(h_cnt + 0, scanout + nscan, sizeof(int), hipMemcpyDeviceToHost, stream));
      RX_HIP(hipMemcpyAsync(h_cnt + 1, scanout2 + nscan, sizeof(int), hipMemcpyDeviceToHost, stream));
      sync_stream();
      t0 = h_cnt[0]; t1 = h_cnt[1];   // counts arrive in pinned host memory
      if (static_cast<long long>(copyptr[d0 - 1]) + t0 + t1 > NB || sendoff[d0] + t0 + t1 > NB)
        throw EngineError(RXMD_E_NBUFFER, "over capacity in append_atoms: residents+ghosts exceed NBUFFER=" + std::to_string(NB));
      sendoff[d0 + 1] = sendoff[d0] + t0; sendoff[d1 + 1] = sendoff[d1] + t1;
      copyptr[d0] = copyptr[d0 - 1] + t0; copyptr[d1] = copyptr[d0] + t1;
      if (t0 > 0)
        k_append_ghosts<<<nblk(nscan, 256), 256, 0, stream>>>(nscan, copyptr[d0 - 1], N, axis, -box.lbox[axis], flags, scanout, spos[0], spos[1], spos[2], type, gid, q, gsrc, groot, sendidx + sendoff[d0]);
      if (t1 > 0)
        k_append_ghosts<<<nblk(nscan, 256), 256, 0, stream>>>(nscan, copyptr[d0], N, axis, box.lbox[axis], flags2, scanout2, spos[0], spos[1], spos[2], type, gid, q, gsrc, groot, sendidx + sendoff[d1]);
    }
  } else
  for (int d = 1; d <= 6; ++d) {
    const int nscan = copyptr[cptridx_[d]], axis = (d - 1) / 2;
    const double sft = (d & 1) ? -box.lbox[axis] : box.lbox[axis];   // xshift, comm.F90:531-548
    k_slab_flags<<<nblk(nscan + 1, 256), 256, 0, stream>>>(nscan, d, box.lbox[axis], shell[axis], spos[axis], type, 0, flags);
    size_t tb = cubtmp_bytes;
    RX_HIP(hipcub::DeviceScan::ExclusiveSum(cubtmp, tb, flags, scanout, nscan + 1, stream));
    int total = 0;
    RX_HIP(hipMemcpyAsync(h_cnt + 0, scanout + nscan, sizeof(int), hipMemcpyDeviceToHost, stream));
    sync_stream();
    total = h_cnt[0];   // counts arrive in pinned host memory
    if (static_cast<long long>(copyptr[d - 1]) + total > NB || sendoff[d] + total > NB)
      throw EngineError(RXMD_E_NBUFFER, "over capacity in append_atoms: residents+ghosts exceed NBUFFER=" + std::to_string(NB));
    if (total > 0)
      k_append_ghosts<<<nblk(nscan, 256), 256, 0, stream>>>(nscan, copyptr[d - 1], N, axis, sft, flags, scanout, spos[0], spos[1], spos[2], type, gid, q, gsrc, groot, sendidx + sendoff[d]);
    copyptr[d] = copyptr[d - 1] + total;
    sendoff[d + 1] = sendoff[d] + total;
  }
  G = copyptr[6];
  if (G > N) k_to_real<<<nblk(G - N, 256), 256, 0, stream>>>(B, N, G, spos[0], spos[1], spos[2], pos[0], pos[1], pos[2]);
  ghosts_valid = true;
  st.nghost_force = G - N; st.nghost_qeq = G - N;
}

// ---------------------------------------------------------------------------------------------
// Single rank, round 5: COPYATOMS(MODE_COPY) and COPYATOMS(MODE_MOVE) without the per-axis host waits.
// The six-stage exchange of a rank that is its own neighbour on every axis (comm.F90:55-100 with self copies) is a fixed function of each
// RESIDENT's normalised coordinates: stage d scans everything the stages of the earlier axes appended, and an image inherits the
// coordinates of its source on the other axes, so whether "the x-image of atom n" is flagged by the y stage is a property of n.  The
// ghost array of the staged build is therefore 26 SEGMENTS laid end to end -- one per non-empty combination (ex, ey, ez), e in
// {none, U: near the upper face, shifted by -lbox, L: near the lower face, shifted by +lbox} -- in stage order and, inside a stage,
// in the order the stage scans its sources (residents, then the segments of the earlier stages in their order); inside a segment the
// residents keep their index order.  That is what the index-ordered force rule (pot.F90:113-144) needs, and it is exactly what
// flag -> scan -> append per stage produced (tests: ghost order against the CPU restatement of the reference, and against the staged path of this file,
// RXMD_NO_STAGE_PAIRS=1).  Three kernels: per-workgroup counts of the 26 predicates (+ normalised coordinates), one scan per
// segment over the workgroups, placement.  One host wait (the totals) instead of three; migration: none.
// MODE_MOVE is the same structure with EXCLUSIVE predicates (an atom leaves through at most one face per axis: its segment is the
// triple of faces it crossed) and dr = 0; a mover's final slot = (atoms that stay) + (movers of earlier segments) + (its rank).
__constant__ unsigned char c_seg_need[26] = {1, 2, 4, 5, 6, 8, 9, 10, 16, 17, 18, 20, 21, 22, 24, 25, 26, 32, 33, 34, 36, 37, 38, 40, 41, 42};   // bit 0/1: x U/L, 2/3: y, 4/5: z
__constant__ signed char c_seg_of[43] = {-1, 0, 1, -1, 2, 3, 4, -1, 5, 6, 7, -1, -1, -1, -1, -1, 8, 9, 10, -1, 11, 12, 13, -1, 14, 15, 16, -1, -1, -1, -1, -1, 17, 18, 19, -1, 20, 21, 22, -1, 23, 24, 25};
static const int seg_stage_first_[8] = {0, 0, 1, 2, 5, 8, 17, 26};     // first segment of stage d (1..6), [7] = end
struct SegGeom { double lbox[3], dr[3]; };
__device__ inline unsigned seg_code(const SegGeom &sg, double s0, double s1, double s2) {     // inBuffer, comm.F90:551-576, both faces of the three axes
  unsigned c = 0u;
  c |= (sg.lbox[0] - sg.dr[0] < s0) ? 1u : 0u;  c |= (s0 <= sg.dr[0]) ? 2u : 0u;
  c |= (sg.lbox[1] - sg.dr[1] < s1) ? 4u : 0u;  c |= (s1 <= sg.dr[1]) ? 8u : 0u;
  c |= (sg.lbox[2] - sg.dr[2] < s2) ? 16u : 0u; c |= (s2 <= sg.dr[2]) ? 32u : 0u;
  return c;
}
template <bool MOVE> __device__ inline bool seg_pred(unsigned code, unsigned need) { return MOVE ? code == need : (code & need) == need; }

// pass 1: normalised coordinates of the residents (xu2xs), their face code, and per workgroup how many of its atoms each segment takes
// (MOVE: slot 26 = the atoms that stay).  cnt[seg * nblocks + block].
template <bool MOVE>
__global__ void __launch_bounds__(256) k_seg_count(int N, BoxDev B, SegGeom sg, const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
                                                   double *__restrict__ sx, double *__restrict__ sy, double *__restrict__ sz, const int *__restrict__ type,
                                                   unsigned char *__restrict__ code_out, int *__restrict__ cnt) {
  __shared__ int s_w[27][4];
  const int n = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  unsigned code = 0u; bool live = false;
  if (n < N) {
    const double r0 = x[n], r1 = y[n], r2 = z[n];
    const double a = (B.Hi[0] * r0 + B.Hi[1] * r1 + B.Hi[2] * r2) - B.obox[0], b = (B.Hi[3] * r0 + B.Hi[4] * r1 + B.Hi[5] * r2) - B.obox[1],
                 c = (B.Hi[6] * r0 + B.Hi[7] * r1 + B.Hi[8] * r2) - B.obox[2];
    sx[n] = a; sy[n] = b; sz[n] = c;
    live = !MOVE || type[n] > 0;
    if (live) code = seg_code(sg, a, b, c);
    code_out[n] = static_cast<unsigned char>(code | (live ? 0u : 128u));
  }
  const unsigned long long any = __ballot(code != 0u);
  for (int s = 0; s < 26; ++s) {
    int c_ = 0;
    if (any) c_ = __popcll(__ballot(live && seg_pred<MOVE>(code, c_seg_need[s])));
    if (lane == 0) s_w[s][w] = c_;
  }
  if (MOVE) { const int c_ = __popcll(__ballot(live && code == 0u)); if (lane == 0) s_w[26][w] = c_; }
  __syncthreads();
  if (threadIdx.x < (MOVE ? 27 : 26)) cnt[static_cast<size_t>(threadIdx.x) * gridDim.x + blockIdx.x] = s_w[threadIdx.x][0] + s_w[threadIdx.x][1] + s_w[threadIdx.x][2] + s_w[threadIdx.x][3];
}
// pass 2: one workgroup per segment: exclusive prefix of its per-workgroup counts in place, its total -> tot[seg]
// hpub (the ghost build of one rank): the total also goes straight into pinned host memory, tagged with the build's sequence number in the upper half of
// the word -- the host polls these words instead of waiting for a copy behind the placement kernel (pinned_wait)
__global__ void __launch_bounds__(256) k_seg_scan(int nblocks, int *__restrict__ cnt, int *__restrict__ tot, unsigned long long *hpub = nullptr, unsigned seq = 0u) {
  __shared__ int s_p[256];
  int *c = cnt + static_cast<size_t>(blockIdx.x) * nblocks;
  const int per = (nblocks + 255) / 256, b0 = threadIdx.x * per, b1 = min(b0 + per, nblocks);
  int sum = 0;
  for (int b = b0; b < b1; ++b) sum += c[b];
  s_p[threadIdx.x] = sum;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const int v = threadIdx.x >= o ? s_p[threadIdx.x - o] : 0;
    __syncthreads();
    s_p[threadIdx.x] += v;
    __syncthreads();
  }
  int run = s_p[threadIdx.x] - sum;
  for (int b = b0; b < b1; ++b) { const int v = c[b]; c[b] = run; run += v; }
  if (threadIdx.x == 255) {
    tot[blockIdx.x] = s_p[255];
    if (hpub) __hip_atomic_store(hpub + blockIdx.x, (static_cast<unsigned long long>(seq) << 32) | static_cast<unsigned>(s_p[255]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// the error word and the counts of a list build (d_err[0..15]) into pinned host memory, each tagged with the build's sequence number
__global__ void k_publish_words(int n, const int *__restrict__ src, unsigned long long *hpub, unsigned seq) {
  if (threadIdx.x < n) __hip_atomic_store(hpub + threadIdx.x, (static_cast<unsigned long long>(seq) << 32) | static_cast<unsigned>(src[threadIdx.x]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// rank of this thread's atom among the atoms of its workgroup that segment `need` takes (lanes before it + wavefronts before it), from the
// per-wavefront counts in LDS
template <bool MOVE>
__device__ inline int seg_rank_in_block(const int (*s_w)[4], int s, bool pred, int lane, int w) {
  const unsigned long long m = __ballot(pred);
  int r = __popcll(m & ((1ULL << lane) - 1ULL));
  for (int q = 0; q < w; ++q) r += s_w[s][q];
  return r;
}
// pass 3 (MODE_COPY): every resident writes its images: shifted normalised and real coordinates (xshift comm.F90:531-548, xs2xu), type, gid,
// charge, its source in the stage scan (gsrc: the image of the same atom in the parent segment, or the resident), its root, the send list
__global__ void __launch_bounds__(256) k_seg_place_ghosts(int N, int NB, BoxDev B, SegGeom sg, const unsigned char *__restrict__ code_in, const int *__restrict__ cnt, const int *__restrict__ tot,
                                                          double *__restrict__ sx, double *__restrict__ sy, double *__restrict__ sz, double *__restrict__ x, double *__restrict__ y, double *__restrict__ z,
                                                          int *__restrict__ type, long long *__restrict__ gid, double *__restrict__ q, int *__restrict__ gsrc, int *__restrict__ groot, int *__restrict__ sendidx) {
  __shared__ int s_w[26][4], s_base[27];
  const int n = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const unsigned code = n < N ? code_in[n] : 0u;
  if (threadIdx.x == 0) { int o = N; for (int s = 0; s < 26; ++s) { s_base[s] = o; o += tot[s]; } s_base[26] = o; }
  const unsigned long long any = __ballot(code != 0u);
  if (__syncthreads_or(any != 0ULL) == 0) return;                       // an interior workgroup: nothing to place
  for (int s = 0; s < 26; ++s) {
    const int c_ = any ? __popcll(__ballot(seg_pred<false>(code, c_seg_need[s]))) : 0;
    if (lane == 0) s_w[s][w] = c_;
  }
  __syncthreads();
  if (!any) return;                                                      // (wave-uniform)
  double a0 = 0.0, b0 = 0.0, c0 = 0.0, qn = 0.0; int tn = 0; long long gn = 0;
  if (code) { a0 = sx[n]; b0 = sy[n]; c0 = sz[n]; qn = q[n]; tn = type[n]; gn = gid[n]; }
  for (int s = 0; s < 26; ++s) {
    const unsigned need = c_seg_need[s];
    const bool p = seg_pred<false>(code, need);
    const int r = seg_rank_in_block<false>(s_w, s, p, lane, w);
    const unsigned pneed = need >= 16u ? (need & 15u) : (need >= 4u ? (need & 3u) : 0u);      // the same atom one stage earlier
    int src = n;
    if (pneed) { const int ps = c_seg_of[pneed]; const int rp = seg_rank_in_block<false>(s_w, ps, seg_pred<false>(code, pneed), lane, w); src = s_base[ps] + cnt[static_cast<size_t>(ps) * gridDim.x + blockIdx.x] + rp; }
    if (!p) continue;
    const int m = s_base[s] + cnt[static_cast<size_t>(s) * gridDim.x + blockIdx.x] + r;
    if (m >= NB) continue;                                               // over capacity: the host sees the totals and raises the reference's trap
    double a = a0, b = b0, c = c0;
    if (need & 1u) a += -sg.lbox[0]; if (need & 2u) a += sg.lbox[0];
    if (need & 4u) b += -sg.lbox[1]; if (need & 8u) b += sg.lbox[1];
    if (need & 16u) c += -sg.lbox[2]; if (need & 32u) c += sg.lbox[2];
    sx[m] = a; sy[m] = b; sz[m] = c;
    const double r0 = a + B.obox[0], r1 = b + B.obox[1], r2 = c + B.obox[2];
    x[m] = B.H[0] * r0 + B.H[1] * r1 + B.H[2] * r2; y[m] = B.H[3] * r0 + B.H[4] * r1 + B.H[5] * r2; z[m] = B.H[6] * r0 + B.H[7] * r1 + B.H[8] * r2;
    type[m] = tn; gid[m] = gn; q[m] = qn;
    gsrc[m] = src; groot[m] = n; sendidx[m - N] = src;
  }
}

void Engine::ghost_build_fused() {
  const BoxDev B = boxdev(box);
  SegGeom sg;
  for (int a = 0; a < 3; ++a) { sg.lbox[a] = box.lbox[a]; sg.dr[a] = shell[a]; }
  const int nbk = nblk(N, 256);
  ensure_seg_buffers(nbk);
  k_seg_count<false><<<nbk, 256, 0, stream>>>(N, B, sg, pos[0], pos[1], pos[2], spos[0], spos[1], spos[2], type, seg_code_, seg_cnt);
  // The one host wait of the ghost build: the ghost count sizes every launch behind it.  The scan kernel hands the 26 totals to the host through pinned
  // memory (round 6, late): the host has them while the placement kernel still runs and queues the next kernels behind it -- until then a copy behind the
  // placement kernel and a stream synchronisation left the GPU idle for ~35 us per step.
  const unsigned seq = ++pub_seq;
  k_seg_scan<<<26, 256, 0, stream>>>(nbk, seg_cnt, seg_tot, h_pub, seq);
  k_seg_place_ghosts<<<nbk, 256, 0, stream>>>(N, NB, B, sg, seg_code_, seg_cnt, seg_tot, spos[0], spos[1], spos[2], pos[0], pos[1], pos[2], type, gid, q, gsrc, groot, sendidx);
  pinned_wait(26, seq, "ghost counts");
  for (int s = 0; s < 26; ++s) h_seg[s] = static_cast<int>(h_pub[s] & 0xffffffffull);
  copyptr[0] = N; sendoff[1] = 0;
  for (int d = 1; d <= 6; ++d) {
    long long t = 0;
    for (int s = seg_stage_first_[d]; s < seg_stage_first_[d + 1]; ++s) t += h_seg[s];
    if (static_cast<long long>(copyptr[d - 1]) + t > NB)
      throw EngineError(RXMD_E_NBUFFER, "over capacity in append_atoms: residents+ghosts exceed NBUFFER=" + std::to_string(NB));
    copyptr[d] = copyptr[d - 1] + static_cast<int>(t); sendoff[d + 1] = sendoff[d] + static_cast<int>(t);
  }
  G = copyptr[6];
  ghosts_valid = true;
  st.nghost_force = G - N; st.nghost_qeq = G - N;
}

// pass 3 (MODE_MOVE): every atom goes to its final slot of scratch copies (atoms that stay keep their order, movers follow segment by segment),
// shifted on the axes it crossed; pass 4 copies back and forms the real coordinates of everything (xs2xu).  Both leave at once when nobody moved.
struct MoveArrays { double *d[12]; double *t[12]; int nd; long long *gid, *gid_t; int *type, *type_t; };
__global__ void __launch_bounds__(256) k_seg_place_move(int N, SegGeom sg, const unsigned char *__restrict__ code_in, const int *__restrict__ cnt, const int *__restrict__ tot, MoveArrays A) {
  __shared__ int s_w[27][4], s_base[27];
  int movers = 0;
  for (int s = 0; s < 26; ++s) movers += tot[s];
  if (movers == 0) return;                                               // (uniform over the whole launch)
  const int n = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const unsigned cf = n < N ? code_in[n] : 128u;
  const bool live = (cf & 128u) == 0u; const unsigned code = cf & 63u;
  if (threadIdx.x == 0) { int o = tot[26]; for (int s = 0; s < 26; ++s) { s_base[s] = o; o += tot[s]; } s_base[26] = 0; }
  for (int s = 0; s < 27; ++s) {
    const int c_ = __popcll(__ballot(live && (s < 26 ? code == c_seg_need[s] : code == 0u)));
    if (lane == 0) s_w[s][w] = c_;
  }
  __syncthreads();
  if (!live) return;
  const int s = code ? c_seg_of[code] : 26;                              // (a code with both faces of an axis cannot occur: lbox < s and s <= 0 exclude each other)
  if (s < 0) return;
  // rank among the atoms of this workgroup with the same segment: every lane needs the ballot of ITS segment; segments are few, walk the ones present
  int r = 0;
  for (int t_ = 0; t_ < 27; ++t_) {
    const unsigned long long m = __ballot(s == t_);
    if (s == t_) { r = __popcll(m & ((1ULL << lane) - 1ULL)); for (int q = 0; q < w; ++q) r += s_w[t_][q]; }
  }
  const int m = s_base[s] + cnt[static_cast<size_t>(s) * gridDim.x + blockIdx.x] + r;
  for (int a = 0; a < A.nd; ++a) {
    double v = A.d[a][n];
    if (a < 3) { const unsigned up = 1u << (2 * a), lo = 2u << (2 * a); if (code & up) v += -sg.lbox[a]; if (code & lo) v += sg.lbox[a]; }   // d[0..2] = normalised x, y, z
    A.t[a][m] = v;
  }
  A.gid_t[m] = A.gid[n]; A.type_t[m] = A.type[n];
}
__global__ void __launch_bounds__(256) k_seg_finish_move(int N, BoxDev B, const int *__restrict__ tot, MoveArrays A, double *__restrict__ x, double *__restrict__ y, double *__restrict__ z) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  int movers = 0;
  for (int s = 0; s < 26; ++s) movers += tot[s];
  if (movers) {
    for (int a = 0; a < A.nd; ++a) A.d[a][n] = A.t[a][n];
    A.gid[n] = A.gid_t[n]; A.type[n] = A.type_t[n];
  }
  const double r0 = A.d[0][n] + B.obox[0], r1 = A.d[1][n] + B.obox[1], r2 = A.d[2][n] + B.obox[2];
  x[n] = B.H[0] * r0 + B.H[1] * r1 + B.H[2] * r2; y[n] = B.H[3] * r0 + B.H[4] * r1 + B.H[5] * r2; z[n] = B.H[6] * r0 + B.H[7] * r1 + B.H[8] * r2;
}

void Engine::migrate_fused() {
  // Single rank only: every resident is LIVE (type > 0) -- an atom that leaves through a face comes back in through the opposite one, nothing
  // is ever handed to another rank, so N does not change and there is no dead slot to compact out (the staged path, which several ranks run,
  // does both: k_pack_move marks type = -1, comm.F90:440, and the compaction behind the stages shrinks N).
  const BoxDev B = boxdev(box);
  SegGeom sg;
  for (int a = 0; a < 3; ++a) { sg.lbox[a] = box.lbox[a]; sg.dr[a] = 0.0; }
  const int nbk = nblk(N, 256);
  ensure_seg_buffers(nbk);
  k_seg_count<true><<<nbk, 256, 0, stream>>>(N, B, sg, pos[0], pos[1], pos[2], spos[0], spos[1], spos[2], type, seg_code_, seg_cnt);
  k_seg_scan<<<27, 256, 0, stream>>>(nbk, seg_cnt, seg_tot);
  // scratch: force + bonded scratch arrays as targets (they are recomputed every step), as the staged path does
  MoveArrays A{};
  double *src[12] = {spos[0], spos[1], spos[2], vel[0], vel[1], vel[2], q, qsfp, qsfv, shl[0], shl[1], shl[2]};
  double *tmp[12] = {frc[0], frc[1], frc[2], cds, cd, cc_, deltap, delta, nlp, A0, A1, A2};
  A.nd = ff.pqeq ? 12 : 9;
  for (int a = 0; a < 12; ++a) { A.d[a] = src[a]; A.t[a] = tmp[a]; }
  A.gid = gid; A.gid_t = reinterpret_cast<long long *>(dDlp); A.type = type; A.type_t = perm_in;
  k_seg_place_move<<<nbk, 256, 0, stream>>>(N, sg, seg_code_, seg_cnt, seg_tot, A);
  k_seg_finish_move<<<nbk, 256, 0, stream>>>(N, B, seg_tot, A, pos[0], pos[1], pos[2]);
  G = N;
  lists_valid = false; ghosts_valid = false;
  st.natoms = N;
}

void Engine::ensure_seg_buffers(int nbk) {
  if (nbk <= seg_blocks_cap) return;
  dfree(seg_cnt); dfree(seg_tot); dfree(seg_code_);
  seg_blocks_cap = nbk + nbk / 4 + 16;
  dmalloc(seg_cnt, static_cast<size_t>(27) * seg_blocks_cap); dzalloc(seg_tot, 32); dmalloc(seg_code_, static_cast<size_t>(seg_blocks_cap) * 256);   // (one face code per RESIDENT: sized with the workgroup count, not with the NB of the day)
  if (!h_seg) RX_HIP(hipHostMalloc(reinterpret_cast<void **>(&h_seg), 32 * sizeof(int)));
  if (!h_pub) { RX_HIP(hipHostMalloc(reinterpret_cast<void **>(&h_pub), 32 * sizeof(unsigned long long), hipHostMallocCoherent | hipHostMallocMapped)); for (int k = 0; k < 32; ++k) h_pub[k] = 0ull; }
}

void Engine::halo_refresh(double2 *v2, double *v1) {
  if (multi()) { if (v2) halo_staged(reinterpret_cast<double *>(v2), 2); if (v1) halo_staged(v1, 1); return; }
  if (G <= N) return;
  if (v2) k_refresh2<<<nblk(G - N, 256), 256, 0, stream>>>(N, G, groot, v2);
  if (v1) k_refresh1<<<nblk(G - N, 256), 256, 0, stream>>>(N, G, groot, v1);
}

// ---------------------------------------------------------------------------------------------
// multi-rank: the same six stages with pack -> send_recv -> unpack (reference src/comm.F90:68-86).  A stage whose
// partner is this rank (vprocs(axis) == 1) is a device copy; otherwise the host-supplied transport moves the bytes.
__global__ void k_pack_ghosts(int nscan, int axis, double sft, const int *flags, const int *scanout, const double *sx, const double *sy, const double *sz,
                              const int *type, const long long *gid, const double *q, double *buf, int *sendlist, int nres, int myid, const long long *gowner) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= nscan || !flags[n]) return;
  const int k = scanout[n];
  double a = sx[n], b = sy[n], c = sz[n];
  if (axis == 0) a += sft; else if (axis == 1) b += sft; else c += sft;
  double *o = buf + 6 * static_cast<size_t>(k);
  // the type word also carries who owns the atom: (owner's local index * 1024 + owner rank) * 64 + type, exact in a double.  A resident is its
  // own owner; a ghost that is forwarded keeps the owner it arrived with (direct vector halo, engine.h)
  const long long own = (n < nres) ? (static_cast<long long>(n) * 1024 + myid) : gowner[n];
  o[0] = a; o[1] = b; o[2] = c; o[3] = static_cast<double>(own * 64 + type[n]); o[4] = static_cast<double>(gid[n]); o[5] = q[n];
  sendlist[k] = n;
}
__global__ void k_unpack_ghosts(int cnt, int base, const double *buf, double *sx, double *sy, double *sz, int *type, long long *gid, double *q, int *gsrc, long long *gowner) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= cnt) return;
  const double *o = buf + 6 * static_cast<size_t>(k);
  const int m = base + k;
  const long long w = llrint(o[3]);
  sx[m] = o[0]; sy[m] = o[1]; sz[m] = o[2]; type[m] = static_cast<int>(w & 63); gid[m] = llrint(o[4]); q[m] = o[5];
  gowner[m] = w >> 6;
  gsrc[m] = -1;
}
__global__ void k_pack_vec(int cnt, int ncomp, const int *sendlist, const double *v, double *buf) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= cnt) return;
  const int n = sendlist[k];
  for (int c = 0; c < ncomp; ++c) buf[static_cast<size_t>(k) * ncomp + c] = v[static_cast<size_t>(n) * ncomp + c];
}
__global__ void k_unpack_vec(int cnt, int ncomp, int base, const double *buf, double *v) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= cnt) return;
  for (int c = 0; c < ncomp; ++c) v[static_cast<size_t>(base + k) * ncomp + c] = buf[static_cast<size_t>(k) * ncomp + c];
}
__global__ void k_pack_force(int g0, int cnt, const double *fx, const double *fy, const double *fz, double *buf) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= cnt) return;
  buf[3 * static_cast<size_t>(k)] = fx[g0 + k]; buf[3 * static_cast<size_t>(k) + 1] = fy[g0 + k]; buf[3 * static_cast<size_t>(k) + 2] = fz[g0 + k];
}
__global__ void k_add_force(int cnt, const int *sendlist, const double *buf, double *fx, double *fy, double *fz) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= cnt) return;
  const int n = sendlist[k];      // unique within a stage
  fx[n] += buf[3 * static_cast<size_t>(k)]; fy[n] += buf[3 * static_cast<size_t>(k) + 1]; fz[n] += buf[3 * static_cast<size_t>(k) + 2];
}

// The migration sizes its message buffers from what THIS rank sends.  The RCCL transport learns the incoming size first and grows the
// buffers (grow_xbuf_keep_send); a callback transport (rxmd_comm_ops: torch.distributed, the host-staged MPI binding) has one call
// per message and would have to fail after its size exchange, leaving the peer inside its payload exchange.  With callbacks and the
// engine's own buffers the receive side is therefore sized for the worst case the ghost build already allocates (NBUFFER x 6 doubles:
// more migrants than a rank can hold); host-supplied buffers keep their contract (the callback gets the capacity).
size_t Engine::migrate_xbuf_doubles(size_t from_send_count) const {
  if (nccl || !multi() || (!xbuf_owned && xbuf_doubles > 0)) return from_send_count;
  return std::max(from_send_count, static_cast<size_t>(NB) * 6);
}

void Engine::ensure_xbuf(size_t doubles) {
  if (doubles <= xbuf_doubles) return;
  if (!xbuf_owned && xbuf_doubles > 0) throw EngineError(RXMD_E_COMM, "host-supplied exchange buffers are too small");
  dfree(xbuf_send); dfree(xbuf_recv);
  xbuf_doubles = doubles + doubles / 4 + 4096;
  dmalloc(xbuf_send, xbuf_doubles); dmalloc(xbuf_recv, xbuf_doubles);
  xbuf_owned = true;
}

// a message announced by its size message does not fit: the engine's own buffers grow with the packed send data kept (a caller that
// sized them from its send count only, as the migration does, needs no worst case); host-supplied buffers cannot grow
void Engine::grow_xbuf_keep_send(size_t need, size_t keep) {
  if (!xbuf_owned) throw EngineError(RXMD_E_NBUFFER, "incoming message larger than the host-supplied exchange buffers");
  double *ns = nullptr, *nr = nullptr;
  const size_t cap = need + need / 4 + 4096;
  dmalloc(ns, cap); dmalloc(nr, cap);
  if (keep > 0) RX_HIP(hipMemcpyAsync(ns, xbuf_send, sizeof(double) * keep, hipMemcpyDeviceToDevice, stream));
  sync_stream();
  dfree(xbuf_send); dfree(xbuf_recv);
  xbuf_send = ns; xbuf_recv = nr; xbuf_doubles = cap;
}

long long Engine::exchange_stage(int d, bool reverse, long long nsend, long long known_nrecv) {
  const int to = reverse ? target_node[dinv_[d]] : target_node[d];
  const int from = reverse ? target_node[d] : target_node[dinv_[d]];
  if (to == cfg.myid && from == cfg.myid && !(force_remote && nccl)) {                    // comm.F90:305-315
    if (nsend > 0) RX_HIP(hipMemcpyAsync(xbuf_recv, xbuf_send, sizeof(double) * nsend, hipMemcpyDeviceToDevice, stream));
    return nsend;
  }
  if (nccl) return rccl_exchange(to, from, nsend, known_nrecv);       // native: stays in stream order
  if (!has_comm || !comm.exchange) throw EngineError(RXMD_E_COMM, "vprocs > 1 needs a transport: call rxmd_hip_set_comm first");
  sync_stream();                        // the message must be packed before the transport reads it
  const long long nr = (known_nrecv >= 0 && comm.exchange_known)
                           ? comm.exchange_known(comm.ctx, to, xbuf_send, nsend, from, xbuf_recv, known_nrecv)
                           : comm.exchange(comm.ctx, to, xbuf_send, nsend, from, xbuf_recv, static_cast<long long>(xbuf_doubles));
  if (nr < 0) throw EngineError(RXMD_E_COMM, "exchange callback failed");
  return nr;
}

void Engine::ghost_build_staged() {
  const BoxDev B = boxdev(box);
  // own buffers: sized once for the worst case; host-supplied buffers (rxmd_hip_set_exchange_buffers): each axis asks for what its
  // two messages need, the receive side is bounded by the transport (the callbacks get the capacity, the RCCL path checks it)
  const bool xb_fixed = !xbuf_owned && xbuf_doubles > 0;
  if (!xb_fixed) ensure_xbuf(static_cast<size_t>(NB) * 6);
  k_to_normalised<<<nblk(N, 256), 256, 0, stream>>>(B, 0, N, pos[0], pos[1], pos[2], spos[0], spos[1], spos[2]);
  copyptr[0] = N;
  sendoff[1] = 0;
  if (stage_pairs) {
    // the + and - stage of an axis scan the same atoms (residents and the ghosts of earlier axes): both selections, both size
    // messages and both payloads go together -- three rounds and six host waits per ghost build instead of six and twelve
    for (int d0 = 1; d0 <= 5; d0 += 2) {
      const int d1 = d0 + 1, nscan = copyptr[cptridx_[d0]], axis = (d0 - 1) / 2;
      k_slab_flags<<<nblk(nscan + 1, 256), 256, 0, stream>>>(nscan, d0, box.lbox[axis], shell[axis], spos[axis], type, 0, flags);
      k_slab_flags<<<nblk(nscan + 1, 256), 256, 0, stream>>>(nscan, d1, box.lbox[axis], shell[axis], spos[axis], type, 0, flags2);
      size_t tb = cubtmp_bytes;
      RX_HIP(hipcub::DeviceScan::ExclusiveSum(cubtmp, tb, flags, scanout, nscan + 1, stream));
      tb = cubtmp_bytes;
      RX_HIP(hipcub::DeviceScan::ExclusiveSum(cubtmp, tb, flags2, scanout2, nscan + 1, stream));
      int t0 = 0, t1 = 0;
      RX_HIP(hipMemcpyAsync(h_cnt + 0, scanout + nscan, sizeof(int), hipMemcpyDeviceToHost, stream));
      RX_HIP(hipMemcpyAsync(h_cnt + 1, scanout2 + nscan, sizeof(int), hipMemcpyDeviceToHost, stream));
      sync_stream();
      t0 = h_cnt[0]; t1 = h_cnt[1];   // counts arrive in pinned host memory
      if (sendoff[d0] + t0 + t1 > NB) throw EngineError(RXMD_E_NBUFFER, "over capacity in store_atoms (send list)");
      if (xb_fixed) ensure_xbuf(6 * (static_cast<size_t>(t0) + t1));
      sendoff[d0 + 1] = sendoff[d0] + t0; sendoff[d1 + 1] = sendoff[d1] + t1;
      if (t0 > 0) k_pack_ghosts<<<nblk(nscan, 256), 256, 0, stream>>>(nscan, axis, -box.lbox[axis], flags, scanout, spos[0], spos[1], spos[2], type, gid, q, xbuf_send, sendidx + sendoff[d0], N, cfg.myid, gowner);
      if (t1 > 0) k_pack_ghosts<<<nblk(nscan, 256), 256, 0, stream>>>(nscan, axis, box.lbox[axis], flags2, scanout2, spos[0], spos[1], spos[2], type, gid, q, xbuf_send + 6LL * t0, sendidx + sendoff[d1], N, cfg.myid, gowner);
      long long r0 = 0, r1 = 0;
      exchange_pair_sized(d0, 6LL * t0, 6LL * t1, r0, r1);
      const int c0 = static_cast<int>(r0 / 6), c1 = static_cast<int>(r1 / 6);
      if (static_cast<long long>(copyptr[d0 - 1]) + c0 + c1 > NB)
        throw EngineError(RXMD_E_NBUFFER, "over capacity in append_atoms: residents+ghosts exceed NBUFFER=" + std::to_string(NB));
      if (c0 + c1 > 0) k_unpack_ghosts<<<nblk(c0 + c1, 256), 256, 0, stream>>>(c0 + c1, copyptr[d0 - 1], xbuf_recv, spos[0], spos[1], spos[2], type, gid, q, gsrc, gowner);
      copyptr[d0] = copyptr[d0 - 1] + c0; copyptr[d1] = copyptr[d0] + c1;
    }
  } else
  for (int d = 1; d <= 6; ++d) {
    const int nscan = copyptr[cptridx_[d]], axis = (d - 1) / 2;
    const double sft = (d & 1) ? -box.lbox[axis] : box.lbox[axis];
    k_slab_flags<<<nblk(nscan + 1, 256), 256, 0, stream>>>(nscan, d, box.lbox[axis], shell[axis], spos[axis], type, 0, flags);
    size_t tb = cubtmp_bytes;
    RX_HIP(hipcub::DeviceScan::ExclusiveSum(cubtmp, tb, flags, scanout, nscan + 1, stream));
    int total = 0;
    RX_HIP(hipMemcpyAsync(h_cnt + 0, scanout + nscan, sizeof(int), hipMemcpyDeviceToHost, stream));
    sync_stream();
    total = h_cnt[0];   // counts arrive in pinned host memory
    if (sendoff[d] + total > NB) throw EngineError(RXMD_E_NBUFFER, "over capacity in store_atoms (send list)");
    if (xb_fixed) ensure_xbuf(6 * static_cast<size_t>(total));
    if (total > 0)
      k_pack_ghosts<<<nblk(nscan, 256), 256, 0, stream>>>(nscan, axis, sft, flags, scanout, spos[0], spos[1], spos[2], type, gid, q, xbuf_send, sendidx + sendoff[d], N, cfg.myid, gowner);
    sendoff[d + 1] = sendoff[d] + total;
    const long long nr = exchange_stage(d, false, 6LL * total);
    const int cnt = static_cast<int>(nr / 6);
    if (static_cast<long long>(copyptr[d - 1]) + cnt > NB)
      throw EngineError(RXMD_E_NBUFFER, "over capacity in append_atoms: residents+ghosts exceed NBUFFER=" + std::to_string(NB));
    if (cnt > 0) k_unpack_ghosts<<<nblk(cnt, 256), 256, 0, stream>>>(cnt, copyptr[d - 1], xbuf_recv, spos[0], spos[1], spos[2], type, gid, q, gsrc, gowner);
    copyptr[d] = copyptr[d - 1] + cnt;
  }
  G = copyptr[6];
  if (G > N) k_to_real<<<nblk(G - N, 256), 256, 0, stream>>>(B, N, G, spos[0], spos[1], spos[2], pos[0], pos[1], pos[2]);
  ghosts_valid = true;
  st.nghost_force = G - N; st.nghost_qeq = G - N;
  if (halo_direct) direct_halo_setup();             // owners and index lists of this build's ghosts (engine.h)
}

// The + and - stage of one axis are independent (both scan residents and the ghosts of EARLIER axes only, comm.F90:55-66), and
// their send lists, ghost slots and messages lie back to back: one pack, one exchange round with both messages in flight, one
// unpack -- three rounds per halo instead of six.  xbuf_send = [message of d0 | message of d0+1], xbuf_recv likewise.
void Engine::exchange_pair(int d0, bool reverse, long long n0, long long r0, long long n1, long long r1) {
  const int d1 = d0 + 1;
  const int to0 = reverse ? target_node[dinv_[d0]] : target_node[d0], from0 = reverse ? target_node[d0] : target_node[dinv_[d0]];
  const int to1 = reverse ? target_node[dinv_[d1]] : target_node[d1], from1 = reverse ? target_node[d1] : target_node[dinv_[d1]];
  if (to0 == cfg.myid && from0 == cfg.myid && !(force_remote && nccl)) {                  // the axis is not split: both partners are this rank
    if (n0 != r0 || n1 != r1) throw EngineError(RXMD_E_COMM, "self exchange with unequal send and receive counts");
    if (n0 + n1 > 0) RX_HIP(hipMemcpyAsync(xbuf_recv, xbuf_send, sizeof(double) * (n0 + n1), hipMemcpyDeviceToDevice, stream));
    return;
  }
  if (nccl) { rccl_exchange_pair(to0, from0, n0, r0, to1, from1, n1, r1); return; }
  if (!has_comm || !comm.exchange_known) throw EngineError(RXMD_E_COMM, "vprocs > 1 needs a transport: call rxmd_hip_set_comm first");
  sync_stream();
  if (comm.exchange_known(comm.ctx, to0, xbuf_send, n0, from0, xbuf_recv, r0) != r0 ||
      comm.exchange_known(comm.ctx, to1, xbuf_send + n0, n1, from1, xbuf_recv + r0, r1) != r1)
    throw EngineError(RXMD_E_COMM, "halo size changed between the ghost build and a vector exchange");
}

// the same round when the receive counts are not known yet (ghost build): the two size messages travel in one group, one host
// wait, then the two payloads in one group
void Engine::exchange_pair_sized(int d0, long long n0, long long n1, long long &r0, long long &r1) {
  const int d1 = d0 + 1;
  const int to0 = target_node[d0], from0 = target_node[dinv_[d0]], to1 = target_node[d1], from1 = target_node[dinv_[d1]];
  if (to0 == cfg.myid && from0 == cfg.myid && !(force_remote && nccl)) {
    r0 = n0; r1 = n1;
    if (n0 + n1 > 0) RX_HIP(hipMemcpyAsync(xbuf_recv, xbuf_send, sizeof(double) * (n0 + n1), hipMemcpyDeviceToDevice, stream));
    return;
  }
  if (nccl) { rccl_exchange_pair_sized(to0, from0, n0, r0, to1, from1, n1, r1); return; }
  if (!has_comm || !comm.exchange) throw EngineError(RXMD_E_COMM, "vprocs > 1 needs a transport: call rxmd_hip_set_comm first");
  sync_stream();
  r0 = comm.exchange(comm.ctx, to0, xbuf_send, n0, from0, xbuf_recv, static_cast<long long>(xbuf_doubles));
  if (r0 < 0) throw EngineError(RXMD_E_COMM, "exchange callback failed");
  r1 = comm.exchange(comm.ctx, to1, xbuf_send + n0, n1, from1, xbuf_recv + r0, static_cast<long long>(xbuf_doubles) - r0);
  if (r1 < 0) throw EngineError(RXMD_E_COMM, "exchange callback failed");
}

// ---- direct vector halo (RXMD_HALO_DIRECT=1; engine.h) -----------------------------------------------------------------------------
__global__ void k_dh_keys(int N, int G, const long long *__restrict__ gowner, int *__restrict__ keys, int *__restrict__ vals) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= G - N) return;
  keys[t] = static_cast<int>(gowner[N + t] & 1023); vals[t] = N + t;
}
__global__ void k_dh_offsets(int np, int n, const int *__restrict__ keys_sorted, int *__restrict__ off) {     // off[p] = first position whose owner rank is >= p
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p > np) return;
  int lo = 0, hi = n;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys_sorted[mid] < p) lo = mid + 1; else hi = mid; }
  off[p] = lo;
}
__global__ void k_dh_requests(int n, const int *__restrict__ ghost, const long long *__restrict__ gowner, double *__restrict__ out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) out[k] = static_cast<double>(gowner[ghost[k]] >> 10);
}
__global__ void k_dh_to_int(int n, const double *__restrict__ in, int *__restrict__ out, int nres, int *err) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const long long i = llrint(in[k]);
  if (i < 0 || i >= nres) { atomicCAS(&err[0], DERR_NONE, DERR_GRID); out[k] = 0; return; }   // a request for an atom this rank does not own
  out[k] = static_cast<int>(i);
}
__global__ void k_dh_unpack(int cnt, int ncomp, const int *__restrict__ ghost, const double *__restrict__ buf, double *__restrict__ v) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= cnt) return;
  const int m = ghost[k];
  for (int c = 0; c < ncomp; ++c) v[static_cast<size_t>(m) * ncomp + c] = buf[static_cast<size_t>(k) * ncomp + c];
}

// segment p of xbuf_send goes to rank p, segment p of xbuf_recv comes from rank p (offsets in atoms, ncomp doubles each): RCCL -- every
// peer in ONE group; callbacks -- np - 1 shifted send_recv rounds (to = me + r, from = me - r: every rank is in the same round at the same
// time, and every rank always posts both halves, as the transports expect); the rank's own segment is a device copy
void Engine::exchange_many(const std::vector<long long> &soff, const std::vector<long long> &roff, int ncomp) {
  const int me = cfg.myid, np = nprocs;
  const bool self_remote = force_remote && nccl;
  if (!self_remote) {
    const long long sc = (soff[me + 1] - soff[me]) * ncomp, rc = (roff[me + 1] - roff[me]) * ncomp;
    if (sc != rc) throw EngineError(RXMD_E_COMM, "direct halo: a rank disagrees with itself about its own images");
    if (sc > 0) RX_HIP(hipMemcpyAsync(xbuf_recv + roff[me] * ncomp, xbuf_send + soff[me] * ncomp, sizeof(double) * sc, hipMemcpyDeviceToDevice, stream));
  }
  if (nccl) { rccl_exchange_many(soff, roff, ncomp); return; }
  if (np == 1) return;
  if (!has_comm || !comm.exchange) throw EngineError(RXMD_E_COMM, "vprocs > 1 needs a transport: call rxmd_hip_set_comm first");
  sync_stream();
  for (int r = 1; r < np; ++r) {
    const int to = (me + r) % np, from = (me - r + np) % np;
    const long long sc = (soff[to + 1] - soff[to]) * ncomp, rc = (roff[from + 1] - roff[from]) * ncomp;
    const long long got = comm.exchange_known ? comm.exchange_known(comm.ctx, to, xbuf_send + soff[to] * ncomp, sc, from, xbuf_recv + roff[from] * ncomp, rc)
                                              : comm.exchange(comm.ctx, to, xbuf_send + soff[to] * ncomp, sc, from, xbuf_recv + roff[from] * ncomp, static_cast<long long>(xbuf_doubles) - roff[from] * ncomp);
    if (got != rc) throw EngineError(RXMD_E_COMM, "direct halo: message size differs from what the request phase announced");
  }
}

// after a ghost build: group the ghosts by owner rank, tell every rank how many of its atoms each other rank needs (one all-reduce of an
// np x np table), send the owners their index lists
void Engine::direct_halo_setup() {
  dh_ready = false;
  const int np = nprocs, me = cfg.myid, ng = G - N;
  if (np > 1000) throw EngineError(RXMD_E_ARG, "direct halo: more than 1000 ranks");
  dh_need_off.assign(np + 1, 0); dh_serve_off.assign(np + 1, 0);
  if (ng > 0) {
    k_dh_keys<<<nblk(ng, 256), 256, 0, stream>>>(N, G, gowner, dh_keys, dh_vals);
    size_t tb = cubtmp_bytes;
    RX_HIP(hipcub::DeviceRadixSort::SortPairs(cubtmp, tb, dh_keys, dh_keys2, dh_vals, dh_ghost, ng, 0, 10, stream));
  }
  k_dh_offsets<<<nblk(np + 1, 256), 256, 0, stream>>>(np, ng, dh_keys2, dh_off);
  std::vector<int> off(np + 1);
  RX_HIP(hipMemcpyAsync(off.data(), dh_off, sizeof(int) * (np + 1), hipMemcpyDeviceToHost, stream));
  sync_stream();
  for (int p = 0; p <= np; ++p) dh_need_off[p] = off[p];
  std::vector<double> table(static_cast<size_t>(np) * np, 0.0);                 // table[a * np + b] = atoms of rank b that rank a holds as ghosts
  for (int p = 0; p < np; ++p) table[static_cast<size_t>(me) * np + p] = static_cast<double>(off[p + 1] - off[p]);
  if (np > 1) allreduce_host(table.data(), np * np);
  for (int p = 0; p < np; ++p) dh_serve_off[p + 1] = dh_serve_off[p] + static_cast<long long>(table[static_cast<size_t>(p) * np + me]);
  const long long nserve = dh_serve_off[np];
  ensure_xbuf(static_cast<size_t>(std::max<long long>(std::max<long long>(ng, nserve), 1)) * 3);      // up to three components per atom (PQEq shells)
  if (nserve > dh_serve_cap) { dfree(dh_serve); dh_serve_cap = static_cast<int>(nserve + nserve / 4 + 1024); dmalloc(dh_serve, dh_serve_cap); }
  if (ng > 0) k_dh_requests<<<nblk(ng, 256), 256, 0, stream>>>(ng, dh_ghost, gowner, xbuf_send);
  exchange_many(dh_need_off, dh_serve_off, 1);                                   // my requests out, the other ranks' requests in
  if (nserve > 0) k_dh_to_int<<<nblk(nserve, 256), 256, 0, stream>>>(static_cast<int>(nserve), xbuf_recv, dh_serve, N, d_err);
  dh_ready = true;
}

void Engine::halo_direct_exchange(double *v, int ncomp) {
  const int ng = G - N;
  const long long nserve = dh_serve_off[nprocs];
  if (nserve > 0) k_pack_vec<<<nblk(nserve, 256), 256, 0, stream>>>(static_cast<int>(nserve), ncomp, dh_serve, v, xbuf_send);
  exchange_many(dh_serve_off, dh_need_off, ncomp);
  if (ng > 0) k_dh_unpack<<<nblk(ng, 256), 256, 0, stream>>>(ng, ncomp, dh_ghost, xbuf_recv, v);
}

// MODE_QCOPY1 / MODE_QCOPY2 (comm.F90:187-212): ghost slots of an ncomp-interleaved vector, axis by axis
void Engine::halo_staged(double *v, int ncomp) {
  // QCOPY1 / QCOPY2 (comm.F90:2-100 with MODE_QCOPY*): on the second stream the part the main stream waits for is measured at the join
  const bool kt = kt_begin(&st.ms_halo, in_comm_region ? nullptr : &st.ms_halo_exposed, &st.halo_calls, 1);
  struct End { Engine *e; bool kt; ~End() { e->kt_end(kt); } } end_{this, kt};
  if (halo_direct && dh_ready) { halo_direct_exchange(v, ncomp); return; }
  if (stage_pairs && (nccl || (has_comm && comm.exchange_known))) {
    for (int d0 = 1; d0 <= 5; d0 += 2) {
      const int ns0 = sendoff[d0 + 1] - sendoff[d0], ns1 = sendoff[d0 + 2] - sendoff[d0 + 1];
      const int cnt0 = copyptr[d0] - copyptr[d0 - 1], cnt1 = copyptr[d0 + 1] - copyptr[d0];
      if (ns0 + ns1 > 0) k_pack_vec<<<nblk(ns0 + ns1, 256), 256, 0, stream>>>(ns0 + ns1, ncomp, sendidx + sendoff[d0], v, xbuf_send);
      exchange_pair(d0, false, static_cast<long long>(ns0) * ncomp, static_cast<long long>(cnt0) * ncomp, static_cast<long long>(ns1) * ncomp, static_cast<long long>(cnt1) * ncomp);
      if (cnt0 + cnt1 > 0) k_unpack_vec<<<nblk(cnt0 + cnt1, 256), 256, 0, stream>>>(cnt0 + cnt1, ncomp, copyptr[d0 - 1], xbuf_recv, v);
    }
    return;
  }
  for (int d = 1; d <= 6; ++d) {
    const int ns = sendoff[d + 1] - sendoff[d];
    if (ns > 0) k_pack_vec<<<nblk(ns, 256), 256, 0, stream>>>(ns, ncomp, sendidx + sendoff[d], v, xbuf_send);
    const int cnt = copyptr[d] - copyptr[d - 1];
    const long long nr = exchange_stage(d, false, static_cast<long long>(ns) * ncomp, static_cast<long long>(cnt) * ncomp);
    if (nr != static_cast<long long>(cnt) * ncomp) throw EngineError(RXMD_E_COMM, "halo size changed between the ghost build and a vector exchange");
    if (cnt > 0) k_unpack_vec<<<nblk(cnt, 256), 256, 0, stream>>>(cnt, ncomp, copyptr[d - 1], xbuf_recv, v);
  }
}

void Engine::fold_ghost_forces() {
  if (multi() && stage_pairs && (nccl || (has_comm && comm.exchange_known))) {   // MODE_CPBK by axis, z first; within an axis the + stage's sums land first (6 then 5)
    for (int d0 = 5; d0 >= 1; d0 -= 2) {
      const int g0 = copyptr[d0 - 1], cnt0 = copyptr[d0] - copyptr[d0 - 1], cnt1 = copyptr[d0 + 1] - copyptr[d0];
      const int ns0 = sendoff[d0 + 1] - sendoff[d0], ns1 = sendoff[d0 + 2] - sendoff[d0 + 1];
      if (cnt0 + cnt1 > 0) k_pack_force<<<nblk(cnt0 + cnt1, 256), 256, 0, stream>>>(g0, cnt0 + cnt1, frc[0], frc[1], frc[2], xbuf_send);
      exchange_pair(d0, true, 3LL * cnt0, 3LL * ns0, 3LL * cnt1, 3LL * ns1);
      if (ns1 > 0) k_add_force<<<nblk(ns1, 256), 256, 0, stream>>>(ns1, sendidx + sendoff[d0 + 1], xbuf_recv + 3LL * ns0, frc[0], frc[1], frc[2]);
      if (ns0 > 0) k_add_force<<<nblk(ns0, 256), 256, 0, stream>>>(ns0, sendidx + sendoff[d0], xbuf_recv, frc[0], frc[1], frc[2]);
    }
    return;
  }
  if (multi()) {                                               // MODE_CPBK, reversed stage order (comm.F90:74-78,385-396,474-482)
    for (int d = 6; d >= 1; --d) {
      const int g0 = copyptr[d - 1], cnt = copyptr[d] - copyptr[d - 1];
      if (cnt > 0) k_pack_force<<<nblk(cnt, 256), 256, 0, stream>>>(g0, cnt, frc[0], frc[1], frc[2], xbuf_send);
      const int ns = sendoff[d + 1] - sendoff[d];
      const long long nr = exchange_stage(d, true, 3LL * cnt, 3LL * ns);
      if (nr != 3LL * ns) throw EngineError(RXMD_E_COMM, "returned force count does not match the stage send list");
      if (ns > 0) k_add_force<<<nblk(ns, 256), 256, 0, stream>>>(ns, sendidx + sendoff[d], xbuf_recv, frc[0], frc[1], frc[2]);
    }
    return;
  }
  for (int d = 6; d >= 1; --d) {
    const int g0 = copyptr[d - 1], g1 = copyptr[d];
    if (g1 > g0) k_fold_stage<<<nblk(g1 - g0, 256), 256, 0, stream>>>(g0, g1, gsrc, frc[0], frc[1], frc[2]);
  }
}

// ---------------------------------------------------------------------------------------------
// COPYATOMS(MODE_MOVE) (reference src/comm.F90 with dr = 0): atoms that left [0,lbox) re-enter through
// the periodic image (or go to the neighbour rank) and are appended after the residents; survivors
// are compacted in order (comm.F90:238-257).  All positions take the normalise -> real round trip.
__global__ void k_move_append(int nscan, int base, int axis, double sft, const int *flags, const int *scanout,
                              double *sx, double *sy, double *sz, double *vx, double *vy, double *vz,
                              int *type, long long *gid, double *q, double *qsfp, double *qsfv) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= nscan || !flags[n]) return;
  const int m = base + scanout[n];
  double a = sx[n], b = sy[n], c = sz[n];
  if (axis == 0) a += sft; else if (axis == 1) b += sft; else c += sft;
  sx[m] = a; sy[m] = b; sz[m] = c; vx[m] = vx[n]; vy[m] = vy[n]; vz[m] = vz[n];
  type[m] = type[n]; gid[m] = gid[n]; q[m] = q[n]; qsfp[m] = qsfp[n]; qsfv[m] = qsfv[n];
  type[n] = -1;   // comm.F90:440
}
// the same append for one more per-atom array (PQEq shell displacement, unshifted: comm.F90:165-167); runs before k_move_append
__global__ void k_move_append_extra(int nscan, int base, const int *flags, const int *scanout, double *a) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= nscan || !flags[n]) return;
  a[base + scanout[n]] = a[n];
}
__global__ void k_pack_extra3(int nscan, const int *flags, const int *scanout, int W, int o, const double *a0, const double *a1, const double *a2, double *buf) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= nscan || !flags[n]) return;
  double *p = buf + static_cast<size_t>(W) * scanout[n] + o;
  p[0] = a0[n]; p[1] = a1[n]; p[2] = a2[n];
}
__global__ void k_unpack_extra3(int cnt, int base, int W, int o, const double *buf, double *a0, double *a1, double *a2) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= cnt) return;
  const double *p = buf + static_cast<size_t>(W) * k + o;
  a0[base + k] = p[0]; a1[base + k] = p[1]; a2[base + k] = p[2];
}
__global__ void k_alive_flags(int n, const int *type, int *flags) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= n) flags[i] = (i < n && type[i] > 0) ? 1 : 0;
}
template <class T>
__global__ void k_compact(int n, const int *flags, const int *scanout, const T *src, T *dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && flags[i]) dst[scanout[i]] = src[i];
}

__global__ void k_pack_move(int nscan, int axis, double sft, const int *flags, const int *scanout, const double *sx, const double *sy, const double *sz,
                            const double *vx, const double *vy, const double *vz, int *type, const long long *gid, const double *q,
                            const double *qsfp, const double *qsfv, double *buf, int W) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= nscan || !flags[n]) return;
  double a = sx[n], b = sy[n], c = sz[n];
  if (axis == 0) a += sft; else if (axis == 1) b += sft; else c += sft;
  double *o = buf + static_cast<size_t>(W) * scanout[n];
  o[0] = a; o[1] = b; o[2] = c; o[3] = vx[n]; o[4] = vy[n]; o[5] = vz[n];
  o[6] = static_cast<double>(type[n]); o[7] = static_cast<double>(gid[n]); o[8] = q[n]; o[9] = qsfp[n]; o[10] = qsfv[n];
  type[n] = -1;   // comm.F90:440
}
__global__ void k_unpack_move(int cnt, int base, const double *buf, double *sx, double *sy, double *sz, double *vx, double *vy, double *vz,
                              int *type, long long *gid, double *q, double *qsfp, double *qsfv, int W) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= cnt) return;
  const double *o = buf + static_cast<size_t>(W) * k;
  const int m = base + k;
  sx[m] = o[0]; sy[m] = o[1]; sz[m] = o[2]; vx[m] = o[3]; vy[m] = o[4]; vz[m] = o[5];
  type[m] = static_cast<int>(llrint(o[6])); gid[m] = llrint(o[7]); q[m] = o[8]; qsfp[m] = o[9]; qsfv[m] = o[10];
}

void Engine::migrate() {
  if (!multi() && stage_pairs) { migrate_fused(); return; }     // single rank: four kernels, no host wait
  const BoxDev B = boxdev(box);
  k_to_normalised<<<nblk(N, 256), 256, 0, stream>>>(B, 0, N, pos[0], pos[1], pos[2], spos[0], spos[1], spos[2]);
  int cp[7];
  cp[0] = N;
  int moved = 0;
  if (multi() && stage_pairs) {
    // the two stages of an axis are independent here too (an atom cannot leave through both faces of one axis): selections, size
    // messages and payloads of both together, as in the ghost build
    const int W = ff.pqeq ? 14 : 11;             // + shell displacement (comm.F90:153,165-167)
    for (int d0 = 1; d0 <= 5; d0 += 2) {
      const int d1 = d0 + 1, nscan = cp[cptridx_[d0]], axis = (d0 - 1) / 2;
      k_slab_flags<<<nblk(nscan + 1, 256), 256, 0, stream>>>(nscan, d0, box.lbox[axis], 0.0, spos[axis], type, 1, flags);
      k_slab_flags<<<nblk(nscan + 1, 256), 256, 0, stream>>>(nscan, d1, box.lbox[axis], 0.0, spos[axis], type, 1, flags2);
      size_t tb = cubtmp_bytes;
      RX_HIP(hipcub::DeviceScan::ExclusiveSum(cubtmp, tb, flags, scanout, nscan + 1, stream));
      tb = cubtmp_bytes;
      RX_HIP(hipcub::DeviceScan::ExclusiveSum(cubtmp, tb, flags2, scanout2, nscan + 1, stream));
      int t0 = 0, t1 = 0;
      RX_HIP(hipMemcpyAsync(h_cnt + 0, scanout + nscan, sizeof(int), hipMemcpyDeviceToHost, stream));
      RX_HIP(hipMemcpyAsync(h_cnt + 1, scanout2 + nscan, sizeof(int), hipMemcpyDeviceToHost, stream));
      sync_stream();
      t0 = h_cnt[0]; t1 = h_cnt[1];   // counts arrive in pinned host memory
      ensure_xbuf(migrate_xbuf_doubles(static_cast<size_t>(std::max(t0 + t1, 1)) * W + 4096));
      double *b1 = xbuf_send + static_cast<size_t>(W) * t0;
      if (t0 > 0) {
        if (ff.pqeq) k_pack_extra3<<<nblk(nscan, 256), 256, 0, stream>>>(nscan, flags, scanout, W, 11, shl[0], shl[1], shl[2], xbuf_send);
        k_pack_move<<<nblk(nscan, 256), 256, 0, stream>>>(nscan, axis, -box.lbox[axis], flags, scanout, spos[0], spos[1], spos[2], vel[0], vel[1], vel[2], type, gid, q, qsfp, qsfv, xbuf_send, W);
      }
      if (t1 > 0) {
        if (ff.pqeq) k_pack_extra3<<<nblk(nscan, 256), 256, 0, stream>>>(nscan, flags2, scanout2, W, 11, shl[0], shl[1], shl[2], b1);
        k_pack_move<<<nblk(nscan, 256), 256, 0, stream>>>(nscan, axis, box.lbox[axis], flags2, scanout2, spos[0], spos[1], spos[2], vel[0], vel[1], vel[2], type, gid, q, qsfp, qsfv, b1, W);
      }
      long long r0 = 0, r1 = 0;
      exchange_pair_sized(d0, static_cast<long long>(W) * t0, static_cast<long long>(W) * t1, r0, r1);
      const int c0 = static_cast<int>(r0 / W), c1 = static_cast<int>(r1 / W);
      if (static_cast<long long>(cp[d0 - 1]) + c0 + c1 > NB) throw EngineError(RXMD_E_NBUFFER, "over capacity in append_atoms (MODE_MOVE)");
      if (c0 + c1 > 0) {
        k_unpack_move<<<nblk(c0 + c1, 256), 256, 0, stream>>>(c0 + c1, cp[d0 - 1], xbuf_recv, spos[0], spos[1], spos[2], vel[0], vel[1], vel[2], type, gid, q, qsfp, qsfv, W);
        if (ff.pqeq) k_unpack_extra3<<<nblk(c0 + c1, 256), 256, 0, stream>>>(c0 + c1, cp[d0 - 1], W, 11, xbuf_recv, shl[0], shl[1], shl[2]);
      }
      cp[d0] = cp[d0 - 1] + c0; cp[d1] = cp[d0] + c1;
      moved += t0 + t1 + c0 + c1;
    }
  } else
  for (int d = 1; d <= 6; ++d) {
    const int nscan = cp[cptridx_[d]], axis = (d - 1) / 2;
    const double sft = (d & 1) ? -box.lbox[axis] : box.lbox[axis];
    k_slab_flags<<<nblk(nscan + 1, 256), 256, 0, stream>>>(nscan, d, box.lbox[axis], 0.0, spos[axis], type, 1, flags);
    size_t tb = cubtmp_bytes;
    RX_HIP(hipcub::DeviceScan::ExclusiveSum(cubtmp, tb, flags, scanout, nscan + 1, stream));
    int total = 0;
    RX_HIP(hipMemcpyAsync(h_cnt + 0, scanout + nscan, sizeof(int), hipMemcpyDeviceToHost, stream));
    sync_stream();
    total = h_cnt[0];   // counts arrive in pinned host memory
    if (multi()) {
      const int W = ff.pqeq ? 14 : 11;             // + shell displacement (comm.F90:153,165-167)
      ensure_xbuf(migrate_xbuf_doubles(static_cast<size_t>(std::max(total, 1)) * W + 4096));
      if (total > 0) {
        if (ff.pqeq) k_pack_extra3<<<nblk(nscan, 256), 256, 0, stream>>>(nscan, flags, scanout, W, 11, shl[0], shl[1], shl[2], xbuf_send);
        k_pack_move<<<nblk(nscan, 256), 256, 0, stream>>>(nscan, axis, sft, flags, scanout, spos[0], spos[1], spos[2], vel[0], vel[1], vel[2], type, gid, q, qsfp, qsfv, xbuf_send, W);
      }
      const long long nr = exchange_stage(d, false, static_cast<long long>(W) * total);
      const int cnt = static_cast<int>(nr / W);
      if (static_cast<long long>(cp[d - 1]) + cnt > NB) throw EngineError(RXMD_E_NBUFFER, "over capacity in append_atoms (MODE_MOVE)");
      if (cnt > 0) {
        k_unpack_move<<<nblk(cnt, 256), 256, 0, stream>>>(cnt, cp[d - 1], xbuf_recv, spos[0], spos[1], spos[2], vel[0], vel[1], vel[2], type, gid, q, qsfp, qsfv, W);
        if (ff.pqeq) k_unpack_extra3<<<nblk(cnt, 256), 256, 0, stream>>>(cnt, cp[d - 1], W, 11, xbuf_recv, shl[0], shl[1], shl[2]);
      }
      cp[d] = cp[d - 1] + cnt;
      moved += total + cnt;
      continue;
    }
    if (static_cast<long long>(cp[d - 1]) + total > NB) throw EngineError(RXMD_E_NBUFFER, "over capacity in append_atoms (MODE_MOVE)");
    if (total > 0 && ff.pqeq)
      for (int a = 0; a < 3; ++a) k_move_append_extra<<<nblk(nscan, 256), 256, 0, stream>>>(nscan, cp[d - 1], flags, scanout, shl[a]);
    if (total > 0)
      k_move_append<<<nblk(nscan, 256), 256, 0, stream>>>(nscan, cp[d - 1], axis, sft, flags, scanout, spos[0], spos[1], spos[2], vel[0], vel[1], vel[2], type, gid, q, qsfp, qsfv);
    cp[d] = cp[d - 1] + total;
    moved += total;
  }
  int newN = N;
  if (moved > 0) {
    const int n = cp[6];
    k_alive_flags<<<nblk(n + 1, 256), 256, 0, stream>>>(n, type, flags);
    size_t tb = cubtmp_bytes;
    RX_HIP(hipcub::DeviceScan::ExclusiveSum(cubtmp, tb, flags, scanout, n + 1, stream));
    RX_HIP(hipMemcpyAsync(h_cnt + 0, scanout + n, sizeof(int), hipMemcpyDeviceToHost, stream));
    sync_stream();
    newN = h_cnt[0];   // counts arrive in pinned host memory
    if (newN > rows10) throw EngineError(RXMD_E_NBUFFER, "resident count grew beyond the 10 A list capacity");
    // scratch: reuse force + bonded scratch arrays as compaction targets (they are recomputed every step)
    double *tmpd[9] = {frc[0], frc[1], frc[2], cds, cd, cc_, deltap, delta, nlp};
    double *srcd[9] = {spos[0], spos[1], spos[2], vel[0], vel[1], vel[2], q, qsfp, qsfv};
    for (int a = 0; a < 9; ++a) {
      k_compact<double><<<nblk(n, 256), 256, 0, stream>>>(n, flags, scanout, srcd[a], tmpd[a]);
      RX_HIP(hipMemcpyAsync(srcd[a], tmpd[a], sizeof(double) * newN, hipMemcpyDeviceToDevice, stream));
    }
    if (ff.pqeq) {
      double *ts[3] = {A0, A1, A2};
      for (int a = 0; a < 3; ++a) {
        k_compact<double><<<nblk(n, 256), 256, 0, stream>>>(n, flags, scanout, shl[a], ts[a]);
        RX_HIP(hipMemcpyAsync(shl[a], ts[a], sizeof(double) * newN, hipMemcpyDeviceToDevice, stream));
      }
    }
    k_compact<long long><<<nblk(n, 256), 256, 0, stream>>>(n, flags, scanout, gid, reinterpret_cast<long long *>(dDlp));
    RX_HIP(hipMemcpyAsync(gid, dDlp, sizeof(long long) * newN, hipMemcpyDeviceToDevice, stream));
    k_compact<int><<<nblk(n, 256), 256, 0, stream>>>(n, flags, scanout, type, perm_in);
    RX_HIP(hipMemcpyAsync(type, perm_in, sizeof(int) * newN, hipMemcpyDeviceToDevice, stream));
  }
  N = newN; G = N;
  k_to_real<<<nblk(N, 256), 256, 0, stream>>>(B, 0, N, spos[0], spos[1], spos[2], pos[0], pos[1], pos[2]);
  lists_valid = false; ghosts_valid = false;
  st.natoms = N;
}

// ---------------------------------------------------------------------------------------------
// cell binning over residents+ghosts: stable radix sort by cell id, z fastest (LINKEDLIST, main.F90:277-318)
__global__ void k_cell_ids(int G, Grid g, const double *sx, const double *sy, const double *sz, int *cellid, int *idx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= G) return;
  int cx = static_cast<int>(floor((sx[i] - g.org[0]) * g.inv[0]));
  int cy = static_cast<int>(floor((sy[i] - g.org[1]) * g.inv[1]));
  cx = min(max(cx, 0), g.n[0] - 1); cy = min(max(cy, 0), g.n[1] - 1);
  // z: the slice index (fz slices per cell), one monotone expression of sz -- the sweeps rely on bin(s1) <= bin(s2) for s1 <= s2
  int czf = static_cast<int>(floor((sz[i] - g.org[2]) * (g.inv[2] * g.fz)));
  czf = min(max(czf, 0), g.nzf - 1);
  cellid[i] = (cx * g.n[1] + cy) * g.nzf + czf;
  idx[i] = i;
}
// cellstart[b] = first sorted position whose slice id is >= b (lower bound; one thread per slice, so empty stretches of a sparse box
// cost nothing serial); cellstart[nfine] = G
__global__ void k_cell_starts(int G, int nfine, const int *__restrict__ cid_sorted, int *__restrict__ cellstart) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b > nfine) return;
  int lo = 0, hi = G;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (cid_sorted[mid] < b) lo = mid + 1; else hi = mid; }
  cellstart[b] = lo;
}
// w component of the packed copy: low 32 bits atom index, bits 32.. type (the list sweeps read both from it; k_sorted_charge later puts the charge there)
__global__ void k_sorted_pos(int G, int N, const int *perm, const int *groot, const double *x, const double *y, const double *z, const int *type, double4 *out, unsigned char *st, int *rootperm, int *invpos) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= G) return;
  const int i = perm[k];
  const double xi = x[i], yi = y[i], zi = z[i];
  const int ti = type[i];
  out[k] = make_double4(xi, yi, zi, __longlong_as_double((static_cast<long long>(ti) << 32) | static_cast<unsigned int>(i)));
  st[k] = static_cast<unsigned char>(ti);
  rootperm[k] = (i < N) ? i : groot[i];
  invpos[i] = k;
}
// xs[k] = v[owner of the atom at cell-sorted position k]: the ghost refresh (MODE_QCOPY1/2, comm.F90:187-212) and the
// spatially sorted gather copy of the vector in one pass
__global__ void k_sorted_vec(int G, const int *__restrict__ rootperm, const double2 *__restrict__ v, double2 *__restrict__ xs) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < G) xs[k] = v[rootperm[k]];
}

void Engine::bin_cells() {
  // residents need fresh normalised coordinates when ghost_build did not just compute them
  k_cell_ids<<<nblk(G, 256), 256, 0, stream>>>(G, grid, spos[0], spos[1], spos[2], cellid, perm_in);
  size_t tb = cubtmp_bytes;
  int bits = 1;
  while ((1LL << bits) < grid.nfine + 1 && bits < 31) ++bits;
  RX_HIP(hipcub::DeviceRadixSort::SortPairs(cubtmp, tb, cellid, cellid_sorted, perm_in, perm, G, 0, bits, stream));
  k_cell_starts<<<nblk(grid.nfine + 1, 256), 256, 0, stream>>>(G, grid.nfine, cellid_sorted, cellstart);
  k_sorted_pos<<<nblk(G, 256), 256, 0, stream>>>(G, N, perm, groot, pos[0], pos[1], pos[2], type, sorted_xyzi, sorted_type, rootperm, invpos);
  if (ff.pqeq) pqeq_sorted_shells();
}

void Engine::sorted_copy(const double2 *v) {
  if (multi()) {   // ghost slots first (six-stage exchange), then the plain permuted copy
    halo_staged(reinterpret_cast<double *>(const_cast<double2 *>(v)), 2);
    k_sorted_vec<<<nblk(G, 256), 256, 0, stream>>>(G, perm, v, xs);
    return;
  }
  k_sorted_vec<<<nblk(G, 256), 256, 0, stream>>>(G, rootperm, v, xs);
}

bool Engine::poison_on() const { return g_poison; }
// RXMD_POISON_ALLOC: what a step rebuilds from scratch holds the pattern again before the rebuild
void Engine::poison_step_scratch() {
  if (!g_poison) return;
  auto fill = [&](void *p, size_t off_bytes, size_t bytes) { if (p && bytes) RX_HIP(hipMemsetAsync(static_cast<char *>(p) + off_bytes, 0xFF, bytes, stream)); };
  const size_t nb = NB, ng = nb - N, ns = nb * 32, nl = static_cast<size_t>(rows10) * S10;
  for (int a = 0; a < 3; ++a) { fill(pos[a], sizeof(double) * N, sizeof(double) * ng); fill(spos[a], sizeof(double) * N, sizeof(double) * ng); fill(frc[a], sizeof(double) * N, sizeof(double) * ng); }
  fill(q, sizeof(double) * N, sizeof(double) * ng); fill(type, sizeof(int) * N, sizeof(int) * ng); fill(gid, sizeof(long long) * N, sizeof(long long) * ng);
  fill(gsrc, 0, sizeof(int) * nb); fill(groot, 0, sizeof(int) * nb); fill(rootperm, 0, sizeof(int) * nb); fill(invpos, 0, sizeof(int) * nb); fill(xs, 0, sizeof(double2) * nb);
  fill(cellid, 0, sizeof(int) * nb); fill(cellid_sorted, 0, sizeof(int) * nb); fill(perm, 0, sizeof(int) * nb); fill(perm_in, 0, sizeof(int) * nb);
  fill(cellstart, 0, sizeof(int) * (static_cast<size_t>(grid.nfine) + 2)); fill(sorted_xyzi, 0, sizeof(double4) * nb); fill(sorted_type, 0, nb);
  if (ff.pqeq) { fill(sorted_shl, 0, sizeof(double4) * nb); fill(hsc, 0, sizeof(double) * nl); fill(pqrow, 0, sizeof(double4) * rows10); for (int a = 0; a < 3; ++a) fill(shl[a], sizeof(double) * N, sizeof(double) * ng); }
  fill(nbr_sm, 0, sizeof(int) * ns); fill(nbrcnt, 0, sizeof(int) * (nb + 1)); fill(boff, 0, sizeof(int) * (nb + 2));
  fill(nbr, 0, sizeof(int) * bcap); fill(brev, 0, sizeof(int) * bcap); fill(bown, 0, sizeof(int) * bcap); fill(btype, 0, bcap);
  for (double *t : {bo0, bo1, bo2, bo3, dln2, dln3, dBOp, A0, A1, A2, A3, cf1, cf2, cf3, cdn, fnx, fny, fnz, etor, econ, epen, ecoa, bt1, bt2, bt3}) fill(t, 0, sizeof(double) * bcap);
  fill(ecoef, 0, sizeof(double) * 6 * nb); fill(ehb_don, 0, sizeof(int2) * ehb_don_cap);
  for (double *t : {deltap, delta, nlp, dDlp, deltalp, cds, cd, cc_}) fill(t, 0, sizeof(double) * nb);
  fill(nb10, 0, sizeof(int) * nl); fill(hess, 0, sizeof(double) * nl); fill(sl10, 0, sizeof(unsigned short) * nl); fill(n10, 0, sizeof(int) * rows10);
  fill(rows_int, 0, sizeof(int) * rows10); fill(rows_bnd, 0, sizeof(int) * rows10);
  { const size_t ngr = win_groups_bound(rows10) + 1;
    fill(rows_sorted, 0, sizeof(int) * ngr * WIN_ROWS); fill(rowcols, 0, sizeof(int) * ngr * WIN_ROWS * 64); fill(grp_base, 0, sizeof(int) * ngr * 32); fill(win_flag, 0, sizeof(int) * (ngr + 1)); fill(win_k, 0, sizeof(int) * ngr * WIN_MAXUNITS); fill(win_cnt, 0, sizeof(int) * ngr);
    fill(win_gint, 0, sizeof(int) * ngr); fill(win_gbnd, 0, sizeof(int) * ngr); }
  fill(sall, 0, sizeof(double2) * rows10); fill(sgh, 0, sizeof(double2) * rows10); fill(wall, 0, sizeof(double2) * rows10); fill(wgh, 0, sizeof(double2) * rows10);
  fill(partials, 0, sizeof(double) * (partials_cap + 1024));
  fill(flags, 0, sizeof(int) * (nb + 1)); fill(scanout, 0, sizeof(int) * (nb + 1)); fill(flags2, 0, sizeof(int) * (nb + 1)); fill(scanout2, 0, sizeof(int) * (nb + 1));
  for (double2 *t : {qst, hst, gst, hst2}) fill(t, sizeof(double2) * N, sizeof(double2) * ng);
  if (xbuf_owned) { fill(xbuf_send, 0, sizeof(double) * xbuf_doubles); fill(xbuf_recv, 0, sizeof(double) * xbuf_doubles); }
}

void Engine::build_ghosts_and_lists(bool qeq_prepass) {
  if (!atoms_set) throw EngineError(RXMD_E_STATE, "atoms were never set");
  const KtPair t_lists = outer_begin(&st.ms_lists);   // (an event pair read at a later host wait: no wait of its own)
  poison_step_scratch();
  { const bool kt = kt_begin(&st.ms_ghost_build); ghost_build(); kt_end(kt); }
  bin_cells();
  build_prologue(3);
  { const bool kt = kt_begin(&st.ms_k_blist); build_bonded_list(); kt_end(kt); }
  sums_from_list = qeq_prepass;
  if (qeq_prepass) qeq_start_vectors();             // the sweep below also forms H.(qs,qt) of the CG start vector
  build_list10();
  // The host wait of the build.  First of all the bond tables: a build with more bonds than they hold left them partially packed (k_bond_csr skips
  // the atoms beyond the capacity) -- they are grown and packed again from the intact staging lines BEFORE anything else can throw, so that no
  // error path leaves undersized tables and stale counts behind.
  if (h_pub && !multi()) {                           // (one rank: the words through pinned memory, no copy + stream synchronisation; h_pub exists once the fused ghost build has run)
    const unsigned seq = ++pub_seq;
    k_publish_words<<<1, 64, 0, stream>>>(16, d_err, h_pub, seq);
    pinned_wait(16, seq, "list build");
    for (int k = 0; k < 16; ++k) h_err[k] = static_cast<int>(h_pub[k] & 0xffffffffull);
  } else
    fetch_device_error();
  if (static_cast<size_t>(h_err[7]) > bcap) {
    free_bond_tables();
    alloc_bond_tables(static_cast<size_t>(h_err[7]) + static_cast<size_t>(h_err[7]) / 4 + 4096);
    build_bonded_list(true);
    fetch_device_error();          // the re-pack searched the mirror slots of the atoms the first pass had skipped: its DERR_NBRINDX must be seen THIS step, under this label
  }
  try {
    check_device_error("list build", false);             // (the words are here already)
  } catch (const EngineError &er) {
    // The row stride of the 10 A list is sized from the MEAN density; a dense region inside a sparse box (a nanoparticle in
    // vacuum) can need more.  The reference stops (fixed MAXNEIGHBS10 = 1500, qeq.F90:248-252); here the list grows once to
    // what the sweep reported -- unless the caller fixed the stride (cfg.maxneighbs10), which keeps the reference's trap.
    if (er.code != RXMD_E_MAXNEIGHBS10 || cfg.maxneighbs10 > 0) throw;
    const int need = h_err[1];
    S10 = (static_cast<int>(need * 1.1) + 64 + 63) / 64 * 64;
    const size_t n = static_cast<size_t>(rows10) * S10;
    dfree(nb10); dfree(hess); dfree(sl10); dmalloc(nb10, n); dmalloc(hess, n); dmalloc(sl10, n);
    if (ff.pqeq) { dfree(hsc); dmalloc(hsc, n); }
    st.n10_stride = S10;
    list10_retry = true;
    try { build_list10(); } catch (...) { list10_retry = false; throw; }
    list10_retry = false;
    check_device_error("list build");
  }
  nbonds = h_err[7]; nbonds_res = h_err[9];
  win_groups = h_err[8];                               // groups of this build (build_windows; the sweep ran over the host-side bound)
  max_row10 = h_err[3]; min_row10 = std::min(h_err[4], h_err[3]);   // longest / shortest 10 A row of this build (k_list10)
  win_maxunits = h_err[5]; win_valid = win_groups > 0 && h_err[6] == 0 && win_maxunits > 0 && (!multi() || (win_nbnd >= 0 && win_nbnd <= win_groups)) && !opt.spmv_no_win;   // window form of the matrix (build_windows)
  collect_timers();
  outer_end(t_lists);
  lists_valid = true;
}

}  // namespace rxmd
