// engine.h -- device-resident state of one rank (= one MI355X) and the kernel-group entry points.
//
// Data layout in HBM (all FP64 unless noted; N = residents, G = residents + ghosts, NB = capacity):
//   per-atom SoA, index = the reference's LOCAL atom index (residents in rxff.bin / arrival order,
//   then ghosts in exchange-stage order, reference src/comm.F90:414-446,494-518).  The force
//   semantics depend on this order, so atoms are never physically re-sorted;
//   spatial sorting exists only as an index permutation for the list builds.
//   bonded tables are COMPACT (CSR): bond o = boff[i] + s is slot s of atom i, its mirror image in the partner's list is brev[o]
//   (round 4; until then slot-major [slot * NB + atom]: 30-slot strides for 5.3 bonds per atom, half-empty cache lines in every kernel)
//   the 10 A list is row-major ELL [row * S10 + k]    (coalesced for wave-per-row kernels),
//   S10 a multiple of 64 so that every row starts on a 512-byte boundary.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/rxmd_hip.h"
#include "ffparams.h"
#include "options.h"

namespace rxmd {

#define RX_HIP(call)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (call);                                                                       \
    if (e_ != hipSuccess) throw EngineError(RXMD_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)

struct EngineError {
  int code;
  std::string msg;
  EngineError(int c, std::string m) : code(c), msg(std::move(m)) {}
};

// ---- flattened force field in device memory -------------------------------------------------
struct DevAtomP { double Val, Valboc, mass, Vale, nlpopt, plp2, povun2, povun5, pval3, pval5, Valangle, Valval, chi, eta; };
struct DevBondP {
  double Desig, Depi, Depipi, pbe1, pbe2, povun1, ovc, v13cor;
  double pbo2, pbo4, pbo6, pboc3, pboc4, pboc5;
  double cBOp1, cBOp3, cBOp5, pbo2h, pbo4h, pbo6h, sw0, sw1, sw2, rc2;
};
struct DevAngleP { double theta00, pval1, pval2, pcoa1, pval7, ppen1, pval4; };
struct DevTorsP { double V1, V2, V3, ptor1, pcot1; };
struct DevHbP { double r0hb, phb1, phb2, phb3; };
// one r^2-table node, 64 bytes = one cache-line half: value and (next - this) for the four tabulated functions,
// so that an interpolation touches ONE node:  f(r2) = v + t * d   (reference lerp, pot.F90:729-743)
struct DevNBTab { double Evdw, dEvdw_, CEvdw, dCEvdw_, Eclmb, dEclmb_, CEclmb, dCEclmb_; };

struct DevFF {
  int nso, n1, nboty, ntoty, nvaty;
  const DevAtomP *atom; const DevBondP *bond; const DevAngleP *angle; const DevTorsP *tors; const DevHbP *hb;
  const int *inxn2, *inxn3, *inxn3hb, *inxn4;
  const unsigned *tor_bits;     // inxn4 != 0 as a bit table of 4096 bits (valid when n1 <= 8): k_e4b keeps it in LDS
  const DevNBTab *tabNB;   // [inxn * (NTABLE+2) + i]
  const double *tabQEq;    // [inxn * (NTABLE+2) + i]
  const double2 *tabQEq2;  // the same as pairs (T[i], T[i+1])
  double UDR, UDRi, rctap2, rctap_pad, cutoff_vpar30, vpar1, vpar2;   // rctap_pad: taper cutoff + the sweep padding (lists.hip)
  double plp1, povun3, povun4, povun6, povun7, povun8;
  double pval6, pval8, pval9, pval10, ppen2, ppen3, ppen4, pcoa2, pcoa3, pcoa4, ptor2, ptor3, ptor4, pcot2;
  // PQEq (pqeq != 0): per-type core charge Z and shell spring K, pair rows, and the three screened-Coulomb tables
  // (core-core, shell-core, shell-shell) as nodes (E, E_next - E, F, F_next - F), F = (1/r) dE/dr   (module.F90:537-611)
  int pqeq, npq1;
  const double *Zpq, *Kspq;
  const int *inxnpq;            // [a * npq1 + b]
  const double4 *tabPcc, *tabPsc, *tabPss;   // [row * (NTABLE+2) + i]
};

struct Box {
  double H[3][3], Hi[3][3];  // GetBoxParams / matinv, reference src/init.F90:610-633, main.F90:557-579
  double lat[6];
  double lbox[3];   // normalised local box edge = 1/vprocs          (init.F90:659)
  double obox[3];   // normalised origin of this rank                (init.F90:665)
  double volume;
};

struct Grid {       // the engine's own cell grid over [-shell, L+shell) in normalised local coords
  int n[3];         // cells per dimension
  double org[3];    // normalised origin (negative)
  double inv[3];    // cells per unit of normalised coordinate
  int ncell;
  // every cell is cut into fz slices along z and the atoms are sorted by (x, y, z-slice): a column (x, y) of the grid is one contiguous,
  // z-ordered run of the sorted arrays, so a sweep takes from each column only the slices within reach of ITS atom (lists.hip)
  int fz, nzf;      // slices per cell; n[2] * fz
  int nfine;        // n[0] * n[1] * nzf = length of cellstart - 1
  double wid[3];    // perpendicular real width of one unit of normalised coordinate (orthorhombic: the lattice constant)
  double cw[3];     // 1 / inv: cell edge in normalised units
  double iwz;       // 1 / wid[2]
  int ortho;        // 1: the three directions are orthogonal (distance^2 = sum of the three gaps^2), 0: only max(gap) is a bound
  int probe;        // timing experiments only (RXMD_LIST_PROBE): 1 = the 10 A sweep stops after its per-row set-up, 2 = it skips the emission
};

// The reference's own cell meshes (non-orthogonal boxes only).  Its linked-list cells are laid out in units of the lattice VECTORS and
// its stencils assume orthogonal axes (NEIGHBORLIST: +-1 cell, main.F90:349-351; the non-bonded mesh: cells whose offset passes an
// orthogonal distance test, init.F90:549-592), and its QEq ghost shell is rctap divided by the lattice constants (qeq.F90:32).  In a
// skewed cell these select FEWER pairs than the cutoffs do; the lists reproduce the selection so that the matrix and the forces are
// the reference's.  With 90-degree angles every pair inside a cutoff passes all three and nothing is evaluated.
struct RefMesh {
  double Hi[9], obox[3];   // xu2xs: the reference bins ALL atoms, ghosts included, by Hi.r - obox (LINKEDLIST, main.F90:298)
  double lc[3];            // bonded cell, normalised (lcsize, init.F90:661)
  double nbl[3];           // non-bonded cell, normalised (nblcsize, init.F90:605)
  double nblr[3];          // the same in length units of its lattice vector (init.F90:545)
  double qlo[3], qhi[3];   // QEq ghost shell in normalised local coordinates: (-QCopyDr, lbox + QCopyDr]  (qeq.F90:32,69; comm.F90:551-576)
};

constexpr int WAVE = 64;

// XCD-aware workgroup order: the dispatcher deals consecutive workgroup ids round-robin to the 8 XCDs (each with its own L2),
// so without a remap every L2 sees rows from all over the box and re-fetches the whole gather vector.  With it the
// workgroups that share an XCD own one contiguous eighth of the rows = one compact region of space (bijective for any grid).
__device__ inline int xcd_swizzle(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// Sum of a double over the 64 lanes of a wavefront, the total returned to every lane.  HIP's __shfl_xor compiles to ds_bpermute (two per
// double and step: 12 dependent LDS-crossbar round trips per sum); this form stays in the vector ALU: four steps inside each row of 16
// lanes with DPP lane permutations (quad swaps, half-row mirror, row mirror), two row broadcasts (lane 15 -> next row, lane 31 -> upper
// half, written only where the row mask says), the total read from lane 63 into scalar registers.  Fixed summation order.
template <int CTRL, int ROW_MASK>
__device__ inline double dpp_move(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  // The four full-row permutes write every lane: v_mov_dpp with NO defined `old` value (mov_dpp, not update_dpp(0, ...)) -- a defined one costs
  // a v_mov per DPP move to initialise the destination (48 of the ~300 vector instructions of a matrix-pass row).  The two masked row
  // broadcasts leave rows unwritten (rows 0 / 2, rows 0 / 1): there `old` is a defined 0, so the lanes they skip add 0.0 and every lane of the
  // wavefront holds a well-defined partial sum whatever the compiler does with the move (4 v_mov more per sum).
  if (ROW_MASK == 0xf) return __hiloint2double(__builtin_amdgcn_mov_dpp(hi, CTRL, ROW_MASK, 0xf, false), __builtin_amdgcn_mov_dpp(lo, CTRL, ROW_MASK, 0xf, false));
  return __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false), __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false));
}
__device__ inline double wave_sum64(double v) {
  v += dpp_move<0xb1, 0xf>(v);     // quad_perm:[1,0,3,2]
  v += dpp_move<0x4e, 0xf>(v);     // quad_perm:[2,3,0,1]
  v += dpp_move<0x141, 0xf>(v);    // row_half_mirror
  v += dpp_move<0x140, 0xf>(v);    // row_mirror: every lane of a row now holds the row's sum
  v += dpp_move<0x142, 0xa>(v);    // row_bcast:15 into rows 1 and 3
  v += dpp_move<0x143, 0xc>(v);    // row_bcast:31 into rows 2 and 3: lane 63 holds the total
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

#ifndef WIN_ROWS_DEF
#define WIN_ROWS_DEF 16
#endif
constexpr int WIN_ROWS = WIN_ROWS_DEF;   // rows of a window group = wavefronts of a workgroup of the window pass (measured: 8 -> see NOTES.md 3)
constexpr int WIN_UNIT = 8;         // cell-sorted positions per window unit (8 x 16 bytes = one 128-byte line of the sorted vector)
constexpr int WIN_MAXUNITS = 448;   // units a group's descriptor holds: 3,584 slots = 56 KB of LDS (two workgroups per CU)
constexpr int WIN_BMW = 2048;       // 64-bit words of the coverage map the build kernel keeps in LDS: a group's positions may span 2048 x 64 x 8 = 1 M

struct ScaleArgs;
struct Engine {
  const Options opt = Options::from_env();   // the environment switches of this engine (options.def), read once at create
  rxmd_config cfg{};
  std::string ffield_path, pqeq_path, err;
  ForceField ff;
  Box box{};
  int vID[3] = {0, 0, 0}, target_node[7] = {0}, nprocs = 1;
  double dt = 0, Lex_w2 = 0;
  std::vector<double> dthm, hmas;
  rxmd_comm_ops comm{};
  bool has_comm = false, tables_ready = false, atoms_set = false, lists_valid = false, ghosts_valid = false;

  // capacities
  int NB = 0, MAXNB = 30, S10 = 0, rows10 = 0;
  int max_row10 = 0, min_row10 = 0;   // longest / shortest row of the current 10 A list
  int num_cu = 256;               // compute units of the device: grid of the persistent kernels (one workgroup per CU)
  int N = 0, G = 0, copyptr[7] = {0};
  int cc[3] = {1, 1, 1};          // reference bonded cell counts, only to derive the ghost shell (init.F90:656)
  double shell[3] = {0, 0, 0};    // FORCE ghost shell in normalised units: NMINCELL*lcsize (pot.F90:28)
  Grid grid{};
  RefMesh rmesh{};

  // ---- device memory ----
  DevFF dff{};
  void *ffblob = nullptr, *pqblob = nullptr;
  double *pos[3] = {}, *vel[3] = {}, *frc[3] = {}, *spos[3] = {};  // real pos, v, f ; normalised-local pos (ghost build)
  double *q = nullptr, *qsfp = nullptr, *qsfv = nullptr;
  // PQEq: shell displacement of every atom (reference spos, module.F90:286; real units, travels with the atom), its cell-sorted
  // copy, the shell-core matrix values of the 10 A list and per-row constants (fpqeq, sum H Z, sum Hsc Z, shell-shell energy)
  double *shl[3] = {}; double4 *sorted_shl = nullptr; double *hsc = nullptr; double4 *pqrow = nullptr;
  int *type = nullptr; long long *gid = nullptr;
  double2 *qst = nullptr, *hst = nullptr, *gst = nullptr;  // (qs,qt) (hs,ht) (gs,gt) interleaved
  double2 *hst2 = nullptr;     // second (hs,ht) buffer: the fused direction kernel reads the old and writes the new one (qeq.hip)
  unsigned *tickets = nullptr; // arrival counters of the in-kernel final reductions (qeq.hip, block_finish)
  double2 *sall = nullptr, *sgh = nullptr, *wall = nullptr, *wgh = nullptr;  // qeq_mode 1: row sums H.(qs,qt), H.(hs,ht): all columns / ghost columns
  int *gsrc = nullptr, *groot = nullptr;                    // ghost -> source index on sender ; -> resident root (self exchange)
  int *rootperm = nullptr;                                  // cell-sorted position -> resident that owns the value (ghosts resolved)
  int *invpos = nullptr;                                    // atom (resident or ghost) -> its cell-sorted position: the direction kernel of the CG scatters the new vector straight into xs
  double2 *xs = nullptr;                                    // cell-sorted gather copy of a QEq vector pair (residents+ghosts)
  int *sendidx = nullptr; int sendoff[8] = {0};             // per-stage send index lists (local indices), concatenated
  // cell binning
  int *cellid = nullptr, *cellid_sorted = nullptr, *perm = nullptr, *perm_in = nullptr, *cellstart = nullptr;
  double4 *sorted_xyzi = nullptr;   // cell-sorted (x,y,z,index-as-bits) copy of real positions
  unsigned char *sorted_type = nullptr;   // cell-sorted atom types (the window form of ENbond stages them next to the positions)
  bool list_selfcheck = false;      // this list build: some box edge is shorter than two cut-offs, an atom can meet its own image
  void *cubtmp = nullptr; size_t cubtmp_bytes = 0;
  int *flags = nullptr, *scanout = nullptr;
  // bonded tables, compact: boff[i] .. boff[i + 1] are the bonds of atom i (residents and ghosts) in list order; per bond its partner (nbr), its owner
  // (bown) and its mirror image, the same bond in the partner's list (brev = the reference's nbrindx, main.F90:383-399, as a direct index)
  int *nbr = nullptr, *nbrcnt = nullptr, *boff = nullptr, *brev = nullptr, *bown = nullptr;
  unsigned char *btype = nullptr; // per bond: type of the partner atom
  int *nbr_sm = nullptr;          // atom-major staging of the list sweep [atom * 32 + slot] (a thread appends without knowing the totals; one 128-byte line per atom)
  size_t bcap = 0; int nbonds = 0; // capacity of the per-bond arrays (grown on demand) / bonds of the current build
  void alloc_bond_tables(size_t cap); void free_bond_tables();
  double *bo0 = nullptr, *bo1 = nullptr, *bo2 = nullptr, *bo3 = nullptr, *dln2 = nullptr, *dln3 = nullptr, *dBOp = nullptr;
  double *A0 = nullptr, *A1 = nullptr, *A2 = nullptr, *A3 = nullptr;
  double *cf1 = nullptr, *cf2 = nullptr, *cf3 = nullptr, *cdn = nullptr, *fnx = nullptr, *fny = nullptr, *fnz = nullptr;
  double *etor = nullptr, *econ = nullptr, *epen = nullptr, *ecoa = nullptr;   // per-bond exponentials shared by many angles/torsions
  double *bt1 = nullptr, *bt2 = nullptr, *bt3 = nullptr;                        // per-bond scratch: terms a lane-per-bond kernel leaves for the per-atom sum behind it
  double *ecoef = nullptr;                                                      // 6 per-atom coefficients of Elnpr (bonded.hip)
  // one-visit torsions (bonded.hip, k_e4b<*, true>): the k-l side of a torsion -- ForceB coefficient of bond k-l and the force on l, summed over i -- per
  // (bond (k, l1), slot of j in the list of k), TW columns per bond, with one flag byte per entry (all zero between FORCE calls)
  double4 *e4b_t = nullptr; unsigned char *e4b_flag = nullptr; size_t e4b_cap = 0; bool e4b_dirty = false;
  void alloc_e4b_delivery(size_t entries);
  int nbonds_res = 0;                                                           // bonds of the residents = boff[N]: the first nbonds_res entries of the tables
  int2 *ehb_don = nullptr; size_t ehb_don_cap = 0; int *ehb_cnt = nullptr; unsigned ehb_donor_types = 0u; int ehb_blocks_per_cu = 0; // hydrogen bonds (bonded.hip): donor list (atom, mask of its hydrogen slots), its length, types X with a row (X, H = 2, any)
  double *deltap = nullptr, *delta = nullptr, *nlp = nullptr, *dDlp = nullptr, *deltalp = nullptr;
  double *cds = nullptr, *cd = nullptr, *cc_ = nullptr;
  // 10 A list
  int *nb10 = nullptr, *n10 = nullptr; double *hess = nullptr;
  size_t partials_cap = 0;
  // reductions
  double *partials = nullptr;  // [nblocks_red * 16]
  double *scal = nullptr;      // device scalars (CG state)
  double *h_scal = nullptr;    // pinned host mirror
  int *d_err = nullptr, *h_err = nullptr, *h_cnt = nullptr;   // h_err: pinned, 4 ints of the device error word + 4 (h_cnt) for the counts the host waits for
  double *xbuf_send = nullptr, *xbuf_recv = nullptr; size_t xbuf_doubles = 0; bool xbuf_owned = false;   // staged-exchange message buffers
  double pe[14] = {0}, astr[6] = {0};

  hipStream_t stream = nullptr;
  // second stream for the halo exchanges that overlap with compute (multi-rank): pack / RCCL send-recv / unpack run here while
  // the main stream works on what does not need the ghosts yet; events order the two (engine.hip: on_comm_stream)
  hipStream_t comm_stream = nullptr; hipEvent_t ev_main = nullptr, ev_comm = nullptr, ev_est = nullptr, ev_spec[2] = {nullptr, nullptr}, ev_upd[2] = {nullptr, nullptr}, ev_pass[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // ev_est: Est of a CG iteration has reached the host
  template <class F> void on_comm_stream(F &&body) {          // body runs with `stream` == comm_stream, after everything queued on the main stream so far
    RX_HIP(hipEventRecord(ev_main, stream));
    RX_HIP(hipStreamWaitEvent(comm_stream, ev_main, 0));
    std::swap(stream, comm_stream);
    in_comm_region = true;
    try { body(); } catch (...) { in_comm_region = false; std::swap(stream, comm_stream); throw; }
    in_comm_region = false;
    std::swap(stream, comm_stream);
    RX_HIP(hipEventRecord(ev_comm, comm_stream));
  }
  bool in_comm_region = false;
  // third stream: the part of FORCE that needs no charges (bond orders, every bonded term, the assembly of the bonded forces) runs next to the
  // ghost-charge halo and the nonbonded sweep (assemble.hip: bonded_chain_begin).  RXMD_NO_BOND_OVERLAP=1: one stream, as until round 5.
  hipStream_t bond_stream = nullptr; hipEvent_t ev_fork = nullptr, ev_bond = nullptr;
  double *fsort[3] = {nullptr, nullptr, nullptr};           // hydrogen-bond acceptor forces by CELL-SORTED position (neighbouring candidates = neighbouring addresses: the atomics coalesce); added to frc behind the sweep
  double *fnb[3] = {nullptr, nullptr, nullptr};             // ENbond's force on the residents while the bonded chain owns frc; added behind the join
  bool bond_overlap() const { return bond_stream != nullptr && !ff.pqeq; }
  void bonded_chain_begin();
  // main stream waits for what on_comm_stream queued; the wait itself is timed: that is the part of the exchange the compute did not hide
  void join_comm_stream() { const bool kt = kt_begin(&st.ms_halo_exposed, nullptr, nullptr, 2); RX_HIP(hipStreamWaitEvent(stream, ev_comm, 0)); kt_end(kt); }
  hipEvent_t ev[8] = {};
  std::vector<double> last_atype, last_pos[3];    // what the array-shaped entry points uploaded last (capi.hip)
  std::vector<double> lex_p, lex_v; bool lex_pending = false;   // qsfp/qsfv handed over by rxmd_hip_put_lex for the next array-shaped QEq/PQEq
  rxmd_stats st{};
  int nstep_qeq = 0; double last_est = 0;
  std::vector<double> est_trace;   // Est of the start vector and of every CG iteration of the last QEq call (host side, a few hundred doubles)
  double atype_resid = 0.0;        // geninit packs atype = type + gid*1e-13 + 1e-14 (geninit.F90:459), ReadXYZ without it (fileio.F90:421): what came in goes out
  long long step_count = 0;
  unsigned long long velocity_draws = 0;    // INITVELOCITY calls so far: the draw index of the counter-based generator (assemble.hip)

  // ---- host API ----
  explicit Engine(const rxmd_config &c);
  ~Engine();
  void kinetic_and_charge(double &ke, double &qsum);   // sum hmas v^2 and sum q over the residents (PRINTE, main.F90:225-230); assemble.hip
  void set_atoms_rxff(int natoms, const double *rec10);
  // the same from the reference's own arrays (atype packed as type + gid*1e-13, REAL coordinates by component): no host-side record,
  // the packed type is split on the device.  Only once the engine is sized (after a first set_atoms_rxff); velocities are zeroed.
  void set_atoms_arrays(int natoms, const double *atype, const double *x, const double *y, const double *z, const double *q, const double *lexp, const double *lexv);
  int get_atoms_rxff(double *rec10, int capacity);
  bool poison_on() const;
  void poison_step_scratch();   // RXMD_POISON_ALLOC=1 (engine.hip): the per-step scratch holds 0xFF bytes again before every rebuild
  void build_ghosts_and_lists(bool qeq_prepass = false);   // COPYATOMS(MODE_COPY) + LINKEDLIST + NEIGHBORLIST + 10 A list/hessian, once per step
  void qeq_start_vectors();        // qs, qt, hs, ht of qeq.F90:36-63 and their cell-sorted copy (before the list sweep that uses them)
  int *rows_int = nullptr, *rows_bnd = nullptr; int n_bnd = 0; bool rows_split_pending = false;   // interior / boundary rows (multi-rank)
  bool rows_split_pending_invalid() const { return n_bnd < 0 || n_bnd > N; }
  bool sums_from_list = false;     // the list sweep left H.(qs,qt) of the CG start vector in sall / sgh
  void qeq();
  void force(bool defer_host_read = false);   // defer: the energies stay in the pinned buffer until finish_force() (step(): no host wait between FORCE and the next step)
  void finish_force();
  bool force_pending = false;
  // run-ahead CG loop (qeq.hip): the scalars of an iteration reach the host as a snapshot the update kernel's tail writes straight into pinned host memory
  // (h_scal + 64 + 64 * parity, sequence number in word 63), no copy, no event: the host polls the sequence number
  unsigned long long snap_seq = 0; double snap_expect[2] = {0.0, 0.0};
  void wait_snapshot(int parity, double seq);
  // the matrix pass is timed on a SAMPLE of its launches there (opt.pass_timing_every): sum and count of the timed ones; rxmd_stats.ms_qeq_spmv = average x all launches
  unsigned long long pass_counter = 0; bool pass_timed_k[2] = {false, false}; double pass_timed_ms = 0.0; long long pass_timed_n = 0;
  void step(int nsteps);
  void migrate();                  // COPYATOMS(MODE_MOVE)
  void thermostat(int mdmode, double treq_K, double vsfact, double gke);   // velocity scaling of the MD loop head (assemble.hip)
  int minimise(double ftol, int max_loops, double *pe_final, long long *evaluations);   // mdmode 10: the reference's conjugate-gradient minimiser (minimise.hip)

  // pieces (each in its own .hip)
  void setup_after_atoms(const std::vector<long long> &natoms_per_type_global);
  void upload_ff();
  void alloc_device();
  void free_device();
  void ghost_build();
  void ghost_build_fused(); void migrate_fused(); void ensure_seg_buffers(int nblocks);   // single rank: the six-stage self exchange as 26 image segments (engine.hip)
  // counts the host waits for, handed over through pinned host memory: word = sequence number << 32 | value; the host polls until every word carries the
  // sequence number of its request (rccl_comm.hip: pinned_wait) -- no copy, no stream synchronisation, and the kernels queued behind the producer keep running
  unsigned long long *h_pub = nullptr; unsigned pub_seq = 0u;
  void pinned_wait(int nwords, unsigned seq, const char *what);
  int *seg_cnt = nullptr, *seg_tot = nullptr, *h_seg = nullptr; unsigned char *seg_code_ = nullptr; int seg_blocks_cap = 0;
  void bin_cells();
  void build_bonded_list(bool pack_only = false);
  void build_list10();
  void build_prologue(int what);   // lists.hip: the words a list build starts from, one launch (1: bonded list, 2: 10 A list and windows)
  bool list10_retry = false;
  // Window form of the 10 A matrix (lists.hip, qeq.hip k_spmv_win): the residents in cell-sorted order in groups of WIN_ROWS rows; per group the
  // set of cell-sorted positions its rows couple to, in units of WIN_UNIT consecutive positions (win_k: first position of each unit, ascending;
  // win_cnt: units); per list entry a 16-bit slot in that window (sl10, bit 15 = ghost column).  The matrix pass stages the window's vector
  // entries in LDS with coalesced loads and reads them from there instead of gathering 16 bytes per entry.
  void build_windows();
  void tune_window_placement();   // qeq.hip: a few placements of the pass's streams in physical memory, the fastest kept (once per engine)
  bool place_tuned = false;
  int *rows_sorted = nullptr, *win_k = nullptr, *win_cnt = nullptr;
  int *win_flag = nullptr;                       // per group: 1 when one of its rows has a ghost partner (the sweep sets it; multi-rank: the boundary groups)
  int *rowcols = nullptr, *grp_base = nullptr;   // per row: first position and length of its 25 candidate runs (64 ints) ; per group: slot base of its 25 stencil columns (32 ints)
  int *win_gint = nullptr, *win_gbnd = nullptr; int win_nbnd = 0;     // multi-rank: groups without / with a row that has a ghost partner
  unsigned short *sl10 = nullptr;
  // run-ahead CG loop in ROW ORDER (qeq.hip): the CG vectors of the residents indexed by their place in rows_sorted (group * WIN_ROWS + wavefront of the window pass)
  double2 *r_qst = nullptr, *r_hst = nullptr, *r_hst2 = nullptr, *r_gst = nullptr, *r_sall = nullptr, *r_sgh = nullptr, *r_wall = nullptr, *r_wgh = nullptr;
  bool rows_live = false;       // the last QEq call ran its loop in row order: r_type, r_n10, r_hst, r_gst hold that call's rows (the placement search times the pass in the form that runs)
  int *r_type = nullptr, *r_n10 = nullptr, *r_xpos = nullptr, *rpos = nullptr, *g_rrow = nullptr;   // per row: type (0 = not a row), row length, cell-sorted position; per atom: its row; per ghost: the row of its owner
  int win_groups = 0, win_maxunits = 0;
  // upper bound of the window groups of n rows: a group holds WIN_ROWS rows of ONE cell column (x, y) of the grid, every column may end in a short group
  size_t win_groups_bound(long long n) const { return static_cast<size_t>(n) / WIN_ROWS + static_cast<size_t>(grid.n[0]) * grid.n[1] + 1; }
  double qeq_iters_smooth = -1.0;               // running mean of the CG iterations per QEq call (the exit test of qeq.F90:114-115 lets single calls stop after two or three)
  bool win_valid = false, win_used = false;    // win_used: the last matrix pass was a window pass
  void halo_refresh(double2 *v2, double *v1);       // QCOPY1/QCOPY2: ghosts <- owners (self exchange, resolved roots)
  void halo_staged(double *v, int ncomp);           // the same through the six-stage exchange (multi-rank)
  long long exchange_stage(int d, bool reverse, long long nsend, long long known_nrecv = -1);  // one send_recv of comm.F90:291-364; returns #doubles received
  void exchange_pair(int d0, bool reverse, long long n0, long long r0, long long n1, long long r1);  // stages d0 and d0+1 (one axis) in one round, counts known
  void exchange_pair_sized(int d0, long long n0, long long n1, long long &r0, long long &r1);       // the same when the receive counts are not known yet (ghost build)
  void rccl_exchange_pair_sized(int to0, int from0, long long n0, long long &r0, int to1, int from1, long long n1, long long &r1);
  int *flags2 = nullptr, *scanout2 = nullptr;       // second stage of an axis pair (ghost build)
  void rccl_exchange_pair(int to0, int from0, long long n0, long long r0, int to1, int from1, long long n1, long long r1);
  bool stage_pairs = true;                          // RXMD_NO_STAGE_PAIRS=1: one round per stage as the reference does (six per halo)
  // Direct vector halo (RXMD_HALO_DIRECT=1): every ghost value comes straight from the rank that OWNS the atom, all peers in one grouped
  // exchange, instead of the reference's x -> y -> z forwarding (comm.F90:68-86: three dependent rounds per halo).  The ghost build carries
  // (owner rank, owner's local index) along with every atom; after it each rank asks its ghosts' owners for their index lists once.
  long long *gowner = nullptr;                      // per atom: owner's local index * 1024 + owner rank
  bool halo_direct = false, dh_ready = false;
  int *dh_ghost = nullptr, *dh_serve = nullptr;     // ghost indices grouped by owner rank ; my residents grouped by the rank that asked for them
  int *dh_keys = nullptr, *dh_keys2 = nullptr, *dh_vals = nullptr, *dh_off = nullptr; int dh_serve_cap = 0;
  std::vector<long long> dh_need_off, dh_serve_off; // prefix offsets per rank (atoms)
  void direct_halo_setup();
  void halo_direct_exchange(double *v, int ncomp);
  void exchange_many(const std::vector<long long> &soff, const std::vector<long long> &roff, int ncomp);   // segment p of xbuf_send -> rank p, segment p of xbuf_recv <- rank p
  void rccl_exchange_many(const std::vector<long long> &soff, const std::vector<long long> &roff, int ncomp);
  void ensure_xbuf(size_t doubles);
  size_t migrate_xbuf_doubles(size_t from_send_count) const;
  void grow_xbuf_keep_send(size_t need, size_t keep);
  bool multi() const { return nprocs > 1 || force_staged; }
  // native RCCL transport (rccl_comm.hip); force_staged / force_remote (env RXMD_FORCE_STAGED / RXMD_FORCE_REMOTE) push a
  // single rank through the staged exchange and through RCCL self send/recv: how the multi-GPU code path runs on ONE GPU in the tests
  void *nccl = nullptr; double *cnt_dev = nullptr, *cnt_host = nullptr; bool force_staged = false, force_remote = false;
  void rccl_init(const unsigned char id128[128], int rank, int world);
  void rccl_destroy();
  long long rccl_exchange(int to, int from, long long nsend, long long known_nrecv);
  void rccl_allreduce_dev(double *dev, int n);
  // every host wait of the engine.  With a RCCL communicator attached a wait can depend on a peer that died or went another way (a
  // rank-local overflow, a mismatch of the exchange pattern): the wait is then a bounded poll -- RXMD_COMM_TIMEOUT_S seconds, default
  // 300 -- that aborts the communicator and throws RXMD_E_COMM instead of hanging the job.
  void sync_stream();
  void sync_event(hipEvent_t e);
  double comm_timeout_s = 300.0;
  bool spin_wait = true;
  void allreduce_scal4(int n = 4);                 // MPI_ALLREDUCE of scal[S_RAW0..n-1] (qeq.hip)
  void allreduce_host(double *buf, int n);         // the same for a host vector (setup paths)
  void ghost_build_staged();
  void migrate_staged();
  void sorted_copy(const double2 *v);               // QCOPY1/QCOPY2 fused with the cell-sorted gather copy -> xs
  void fold_ghost_forces();                         // CPBK
  void bond_orders();
  void bonded_energies();
  void charge_halo();
  void nonbonded(bool to_fnb = false);
  void pqeq_sorted_shells();      // ghost shells <- owners, cell-sorted copy (MODE_COPY payload of spos, comm.F90:129-131)
  void pqeq_update_shells();      // update_shell_positions, pqeq.F90:184-259
  void nonbonded_pqeq();          // ENbond_PQEq, pot.F90:784-923
  void efield_force();            // EEfield, module.F90:359-383
  void remove_momentum();         // LinearMomentum, main.F90:766-797
  void type_sums_device();        // per-type count / KE / momentum / mass of the residents, all ranks, left in tsum (assemble.hip)
  double *tsum = nullptr; struct ScaleArgs *sargs = nullptr;   // 6 x 16 per-type sums ; the factors of a velocity-scaling action (device memory)
  void assemble_forces();
  void accumulate_stress(bool kinetic);   // astr(1:6) on the device (scal[48..53])
  void check_device_error(const char *where, bool fetch = true);
  void fetch_device_error();      // the error word and the counts that ride with it -> h_err (one host wait), nothing thrown
  double reduce_partials(int ncomp, int nblocks, double *out);  // host-side helper
  // Event pairs around single launches / exchanges, from a small pool; a finished pair is read back the next time the host has synchronised
  // with a stream anyway (collect_timers: only pairs whose end event has completed -- an exchange on the second stream may still be running),
  // so timing costs no host wait.  A pair adds its milliseconds to up to two rxmd_stats fields.
  struct KtPair { hipEvent_t a = nullptr, b = nullptr; double *dst = nullptr, *dst2 = nullptr; long long *cnt = nullptr; double scale = 1.0; };
  std::vector<KtPair> kt_free, kt_pending;
  int kt_open = -1;                                // (one nesting level is enough: begin ... end on the stream current at the time)
  KtPair kt_cur{};
  // site >= 0: a timer that fires in EVERY CG iteration (the all-reduces, the vector halo and the join behind it on several ranks): an event pair between two
  // dependent operations costs the chain ~7 us per event (profiles/r06_ab_pass_events.txt), four pairs per iteration were ~50 us of every iteration of the
  // multi-rank loop.  Such a site is timed on every n-th call only (opt.pass_timing_every) and its milliseconds count n-fold; its calls are all counted.
  unsigned long long kt_site_calls[4] = {0, 0, 0, 0};
  int kt_every = 1, kt_phase = 0;                  // Engine::step: section / kernel timers on every kt_every-th step of a long call, milliseconds counted kt_every-fold
  bool kt_begin(double *dst, double *dst2 = nullptr, long long *cnt = nullptr, int site = -1) {
    double scale = 1.0;
    if (site < 0 && kt_every > 1) {
      if (cnt) { *cnt += 1; cnt = nullptr; }
      if (kt_phase % kt_every != 0) return false;
      scale = static_cast<double>(kt_every);
    }
    if (site >= 0) {
      if (cnt) { *cnt += 1; cnt = nullptr; }
      const unsigned long long every = static_cast<unsigned long long>(opt.pass_timing_every > 1 ? opt.pass_timing_every : 1);
      if ((kt_site_calls[site & 3]++ % every) != 0) return false;
      scale = static_cast<double>(every);
    }
    if (kt_free.empty() || kt_cur.a) { if (kt_free.empty()) ++st.timer_pairs_dropped; return false; }
    kt_cur = kt_free.back(); kt_free.pop_back();
    kt_cur.dst = dst; kt_cur.dst2 = dst2; kt_cur.cnt = cnt; kt_cur.scale = scale;
    hipEventRecord(kt_cur.a, stream);
    return true;
  }
  void kt_end(bool open) {
    if (!open) return;
    hipEventRecord(kt_cur.b, stream);
    kt_pending.push_back(kt_cur);
    kt_cur = KtPair{};
  }
  // the same for a section that CONTAINS begin ... end pairs (a whole QEq call, a whole FORCE): its own pair from the pool, handed back explicitly
  KtPair outer_begin(double *dst) {
    KtPair p{};
    if (kt_free.empty()) { ++st.timer_pairs_dropped; return p; }
    p = kt_free.back(); kt_free.pop_back();
    p.dst = dst; p.dst2 = nullptr; p.cnt = nullptr; p.scale = 1.0;          // (the sections -- a whole QEq call, FORCE, the list build -- are timed on every step: the iteration count differs from step to step)
    hipEventRecord(p.a, stream);
    return p;
  }
  void outer_end(KtPair p) { if (!p.a) return; hipEventRecord(p.b, stream); kt_pending.push_back(p); }
  void collect_timers();          // after a stream synchronisation
  void tic(int k) { hipEventRecord(ev[k], stream); }
  double toc(int k0, int k1) { hipEventRecord(ev[k1], stream); hipEventSynchronize(ev[k1]); float ms = 0; hipEventElapsedTime(&ms, ev[k0], ev[k1]); return ms; }
};

// 10 A list entry: bits 0-25 cell-sorted position of the partner, 26-29 its atom type, 30 ghost, 31 periodic self image
constexpr int NB10_IDX_BITS = 26;
constexpr unsigned NB10_IDX_MASK = (1u << NB10_IDX_BITS) - 1u;
constexpr unsigned NB10_GHOST = 1u << 30, NB10_SELF = 1u << 31;

// n10[row] = entries of the row; bit 30: the row has a ghost partner (a boundary row of the domain).  The matrix pass needs the sums over
// ghost columns only there (74 % of the rows of a 979,776-atom domain have none) and reads the flag with the length it needs anyway.
constexpr int N10_GHOST_ROW = 1 << 30, N10_COUNT = N10_GHOST_ROW - 1;

// device error codes written by kernels into Engine::d_err
enum { DERR_NONE = 0, DERR_MAXNB = 1, DERR_MAXN10 = 2, DERR_GRID = 3, DERR_NBRINDX = 4, DERR_TYPE = 5 };

}  // namespace rxmd
