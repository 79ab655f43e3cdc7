/* rxmd_hip.h -- C ABI of the MI355X-native ReaxFF force + charge-equilibration engine.
 *
 * This is the drop-in boundary for the per-step hot path of USCCACS/RXMD.  The reference has
 * no FFI layer: the path sits behind three external Fortran subroutines with implicit
 * interfaces plus module-global state:
 *
 *     subroutine QEq  (atype, pos, q)       reference src/qeq.F90:2
 *     subroutine FORCE(atype, pos, f, q)    reference src/pot.F90:2
 *     subroutine COPYATOMS(imode,dr,...)    reference src/comm.F90:2   (MODE_MOVE from main.F90:75)
 *
 * and the velocity-Verlet loop body around them (reference src/main.F90:64-98).  Every entry
 * point below names the reference interface it replaces.  All functions return 0 on success or
 * a negative RXMD_E_* code (the reference prints and calls MPI_FINALIZE/stop instead,
 * main.F90:402-407, qeq.F90:248-252, comm.F90:467-472); rxmd_hip_last_error() gives the text.
 *
 * Plain C types only (pointers and sizes); one handle per GPU / rank; the caller owns host
 * arrays, the library owns device state.  Not re-entrant per handle (as the reference).
 * The Fortran-side binding a maintainer would add is shown in INTEGRATION.md and shipped as
 * bindings/rxmd_hip_mod.F90 (iso_c_binding).
 */
#ifndef RXMD_HIP_H
#define RXMD_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct rxmd_hip_engine *rxmd_handle;

#define RXMD_OK 0
#define RXMD_E_ARG (-1)         /* bad argument / missing file */
#define RXMD_E_FFIELD (-2)      /* ffield could not be parsed */
#define RXMD_E_NBUFFER (-3)     /* residents+ghosts exceed capacity      (comm.F90:467-472) */
#define RXMD_E_MAXNEIGHBS (-4)  /* bonded neighbour overflow             (main.F90:402-407) */
#define RXMD_E_MAXNEIGHBS10 (-5)/* 10 A neighbour overflow               (qeq.F90:248-252)  */
#define RXMD_E_HIP (-6)         /* HIP runtime error / no device */
#define RXMD_E_STATE (-7)       /* call order (e.g. force before atoms were set) */
#define RXMD_E_COMM (-8)        /* multi-rank exchange failed */
#define RXMD_E_NAN (-9)         /* non-finite energy (e.g. collinear torsion, pot.F90:1391-1394) */

/* Run parameters: the union of what GETPARAMS/INITSYSTEM/get_rxmd_parms hand to the path through
 * module globals (reference src/cmdline.F90:255-297, src/init.F90:28-106, src/module.F90:80-84). */
typedef struct rxmd_config {
  const char *ffield_path; /* --ffield ; parsed as src/param.F90:2-375 */
  double lattice[6];       /* lata,latb,latc,lalpha,lbeta,lgamma of the WHOLE box (rxff.bin header, fileio.F90:493-494) */
  int vprocs[3];           /* rxmd.in `processors`; 1 1 1 for a single GPU */
  int myid;                /* rank in the vprocs grid, x fastest (init.F90:75-77) */
  int isQEq;               /* 0 off, 1 CG, 2 extended Lagrangian (qeq.F90:36-63) */
  int NMAXQEq;             /* max CG iterations */
  double QEq_tol;          /* energy criterion (qeq.F90:114-115) */
  int qstep;               /* QEq every qstep MD steps (main.F90:77) */
  double dt_fs;            /* time step in fs (rxmd.in `time`) */
  double Lex_fqs, Lex_k;   /* rxmd.in `exL` (module.F90:164) */
  int nbuffer;             /* capacity residents+ghosts (NBUFFER); 0 = size from the box */
  int maxneighbs;          /* MAXNEIGHBS, 0 = 30 */
  int maxneighbs10;        /* MAXNEIGHBS10 (row stride of the 10 A list), 0 = size from density */
  int device;              /* HIP device ordinal */
  int qeq_mode;            /* 0 = reference algebra (two matrix passes per CG iteration, qeq.F90:96-166)
                              1 = one pass per iteration (gradient by recurrence); same fixed point */
  int lg;                  /* --lg (cmdline.F90:148-151): the ffield is in the low-gradient format (five-line atom blocks, C_lg column;
                              param.F90:83-86,107-109,197-200) and the vdW table carries the LG + core terms (init.F90:496-514) */
  const char *pqeq_path;   /* --pqeq / rxmd.in PQEqParm (cmdline.F90:112-128,291-293): NULL = plain QEq.  Switches the charge solver to
                              PQEq (pqeq.F90), the nonbonded term to ENbond_PQEq (pot.F90:784-923) and the taper cutoff to 12.5 A */
  int efield_dir;          /* rxmd.in `efield <dir> <strength>` / --efield (cmdline.F90:131-137,286-289): 0 = off, 1..3 = x,y,z; PQEq only */
  int reserved1;
  double efield_strength;  /* [V/A]: force -(q_i + Z_i) E Eev_kcal on every core (module.F90:359-383), -Z_i E Eev_kcal on every shell (pqeq.F90:205),
                              centre-of-mass momentum removed every step (main.F90:70-71).  NOTE the reference's EEfield addresses
                              its force array with the wrong leading extent, so for dir > 1 it pushes the x component of ghost slots;
                              the engine applies the field to atom i along dir, which equals the reference for dir = 1 */
} rxmd_config;

void rxmd_hip_default_config(rxmd_config *cfg);
int rxmd_hip_create(const rxmd_config *cfg, rxmd_handle *out);
int rxmd_hip_destroy(rxmd_handle h);
const char *rxmd_hip_last_error(rxmd_handle h);

/* ---- state in / out ------------------------------------------------------------------------- */
/* ReadBIN (src/fileio.F90:444-555): natoms records of 10 doubles exactly as DAT/rxff.bin holds them
 * for this rank: normalised LOCAL pos[3], v[3], q, atype(=type+gid*1e-13), qsfp, qsfv. */
int rxmd_hip_set_atoms_rxff(rxmd_handle h, int natoms, const double *rec10);
/* WriteBIN (src/fileio.F90:558-653): the same record layout back; returns natoms (may have changed
 * by migration).  Pass NULL to query the count. */
int rxmd_hip_get_atoms_rxff(rxmd_handle h, double *rec10, int capacity);
/* Per-atom arrays of the residents in the engine's current local order (the reference's local
 * index order, which the force semantics depend on -- SURVEY 0.9): any pointer may be NULL.
 * pos/v/f are [natoms][3] row-major REAL coordinates; gid = l2g(atype) (main.F90:582-593). */
int rxmd_hip_get_atoms(rxmd_handle h, int capacity, long long *gid, int *type, double *pos, double *v, double *f, double *q);
int rxmd_hip_set_charges(rxmd_handle h, int natoms, const double *q);
int rxmd_hip_set_velocities(rxmd_handle h, int natoms, const double *v);
/* PQEq shell displacements spos(natoms,3) of the residents, real units (module.F90:286; written by WriteXYZ, fileio.F90:332-333).
 * get returns natoms; both fail with RXMD_E_STATE when the engine was created without pqeq_path */
int rxmd_hip_get_shells(rxmd_handle h, double *spos3, int capacity);
/* Bond lists of the residents after the last rxmd_hip_force: what WriteBND reads from nbrlist / BO(0,:,:) (fileio.F90:27-148).
 * count[i] = nbrlist(i,0); partner_gid[i*maxnb + s] = l2g(atype(nbrlist(i,s+1))); bo[i*maxnb + s] = BO(0,i,s+1); maxnb >= the
 * engine's MAXNEIGHBS (30 unless rxmd_config.maxneighbs says otherwise).  Returns natoms. */
int rxmd_hip_get_bonds(rxmd_handle h, int capacity, int maxnb, int *count, long long *partner_gid, double *bo);
int rxmd_hip_set_shells(rxmd_handle h, int natoms, const double *spos3);

/* ---- the hot path, device resident ---------------------------------------------------------- */
/* QEq(atype,pos,q), src/qeq.F90:2-178.  iters = nstep_qeq, est = last GEst1. */
int rxmd_hip_qeq(rxmd_handle h, int *iters, double *est);
/* FORCE(atype,pos,f,q), src/pot.F90:2-90.  pe[0:13] as module.F90:143-146 (pe[0]=sum). */
int rxmd_hip_force(rxmd_handle h, double pe[14]);
/* nsteps passes of the MD loop body, src/main.F90:64-98 (vkick, Lex charges, drift, COPYATOMS(MOVE),
 * QEq every qstep, FORCE, vkick); mdmode 1 (NVE).  Call rxmd_hip_qeq + rxmd_hip_force once before
 * the first step, as main.F90:27-32 does. */
int rxmd_hip_step(rxmd_handle h, int nsteps);
/* mdmode 10: the reference's geometry minimiser (ConjugateGradient, src/cg.F90:26-98 -- Polak-Ribiere directions, bracketing by the
 * Armijo rule, golden-section line search, every trial point a COPYATOMS(MODE_MOVE) + QEq + FORCE) on the device-resident state:
 * positions, search direction and gradients never leave HBM.  ftol = rxmd.in `CG_tol` (energy criterion per atom), max_loops =
 * CG_MaxMinLoop (500).  Returns the number of CG loops (>= 0) or an error; *pe = final potential energy, *evaluations = QEq+FORCE
 * pairs spent.  Velocities are zero afterwards, as the reference leaves them (cg.F90:39). */
int rxmd_hip_minimise(rxmd_handle h, double ftol, int max_loops, double *pe, long long *evaluations);
/* nstep_qeq of the last QEq call (printed in the MDstep line, src/main.F90:261); negative = error */
int rxmd_hip_last_qeq_iters(rxmd_handle h);
/* The velocity scaling the reference's MD loop applies at its head when mod(nstep,sstep)==0 (src/main.F90:45-61), on the
 * device: mdmode 0 and 6 (INITVELOCITY, src/init.F90:292-360: fresh unit-variance Gaussian velocities for every atom -- a counter-based
 * generator keyed by the GLOBAL atom id, so the draw does not depend on the decomposition; RXMD_SEED in the environment changes the
 * stream -- centre-of-mass velocity removed through the all-reduce, scaled to kinetic energy 1.5 treq per atom), 4 (v *= vsfact), 5 (rescale to treq_K; gke_per_atom = kinetic energy per atom of the last PRINTE, <= 0: the
 * current one), 7 (per element, ScaleTemperature :722-763), 8 (only beyond 5 %, AdjustTemperature :684-719); 7 and 8 remove the
 * centre-of-mass momentum afterwards (LinearMomentum :766-797).  The caller keeps the sstep cadence:
 *   for (n = 0; n < nsteps; n += sstep) { rxmd_hip_thermostat(h, mdmode, treq, vsfact, -1); rxmd_hip_step(h, sstep); } */
int rxmd_hip_thermostat(rxmd_handle h, int mdmode, double treq_K, double vsfact, double gke_per_atom);
/* PRINTE reductions (src/main.F90:210-274) for this rank: ke = sum hmas*v^2, qsum, pe[14], and the stress accumulators
 * astr[6] = (xx,yy,zz,yz,zx,xy): virial sum over residents+ghosts of pos*f before the fold (pot.F90:65-72) plus m*v*v of every
 * step (main.F90:86-94), raw sums since the previous read -- passing astr != NULL resets them, as PRINTE does (main.F90:270).
 * Pressure as printed: sum(astr[0..2])/3 / MDBOX * 6.94728103 / pstep  [GPa] (main.F90:233,252). */
int rxmd_hip_get_energy(rxmd_handle h, double *ke, double *qsum, double pe[14], double astr[6]);

/* ---- the same path behind the reference's own argument shapes ------------------------------- */
/* Host arrays in the reference layout: atype(NBUFFER) packed type+gid*1e-13, pos/f(NBUFFER,3)
 * column-major REAL coordinates, residents 1..natoms.  These upload, run the device path and
 * download; they are what a Fortran `QEq`/`FORCE` shim calls (bindings/rxmd_hip_mod.F90). */
int rxmd_hip_QEq(rxmd_handle h, int nbuffer, int natoms, const double *atype, const double *pos, double *q);
int rxmd_hip_FORCE(rxmd_handle h, int nbuffer, int natoms, const double *atype, const double *pos, double *f, const double *q, double pe[14]);
/* the same for `subroutine PQEq(atype,pos,q)` (src/pqeq.F90:2) and FORCE with isPQEq: the shell displacements spos(NBUFFER,3)
 * of module atoms (module.F90:286) go in, PQEq returns them moved (update_shell_positions, pqeq.F90:184-259) */
int rxmd_hip_PQEq(rxmd_handle h, int nbuffer, int natoms, const double *atype, const double *pos, double *q, double *spos);
int rxmd_hip_FORCE_pqeq(rxmd_handle h, int nbuffer, int natoms, const double *atype, const double *pos, double *f, const double *q, const double *spos, double pe[14]);
/* the fictitious charges of the extended-Lagrangian method, module atoms' qsfp/qsfv (module.F90:289, integrated by the driver,
 * main.F90:67-68,98; read AND written by QEq/PQEq, qeq.F90:41-42,51-52): put_lex hands the host's arrays to the NEXT
 * rxmd_hip_QEq / rxmd_hip_PQEq call, get_lex returns what that call left (isQEq = 1: qsfp = q, qsfv = 0).  With isQEq = 2 the
 * array-shaped QEq/PQEq refuse to run (RXMD_E_ARG) unless put_lex came first: the charges would start from qsfp = 0. */
int rxmd_hip_put_lex(rxmd_handle h, int natoms, const double *qsfp, const double *qsfv);
int rxmd_hip_get_lex(rxmd_handle h, int natoms, double *qsfp, double *qsfv);

/* ---- introspection for tests, roofline accounting and the timers table (main.F90:135-182) ---- */
typedef struct rxmd_stats {
  int natoms, nghost_qeq, nghost_force; /* copyptr(6)-NATOMS after the QEq / FORCE ghost builds */
  long long nnz10;                      /* entries of the 10 A list (sum nbplist(0,:)) */
  long long nbonds;                     /* sum nbrlist(:,0) over residents+ghosts */
  int max_n10, max_nb;                  /* maxas(:,3), maxas(:,2) */
  int qeq_iters_last; long long qeq_iters_total; long long qeq_calls;
  double ms_qeq, ms_qeq_list, ms_qeq_spmv, ms_force, ms_lists, ms_bo, ms_nonbond, ms_bonded, ms_step_total;
  long long spmv_launches;              /* number of matrix passes timed in ms_qeq_spmv */
  int n10_stride, nbuffer, cells10[3], cells3[3];
  int n_boundary_rows;                  /* rows of the 10 A matrix with a ghost partner (vprocs > 1: the launch that waits for the vector halo) */
  int spmv_noop_launches;               /* run-ahead CG loop: matrix-pass launches that returned at once (the iteration queued ahead of the exit decision; one per QEq
                                         * call that ends by the exit test).  NOT in spmv_launches / ms_qeq_spmv; a kernel trace counts them as k_spmv calls */
  int reserved[6];
  /* the exchanges of a vprocs > 1 run (comm.F90:2-100), event-timed on the stream they run on, summed since rxmd_hip_reset_timers:
   * ghost build (MODE_COPY incl. its size messages), migration (MODE_MOVE), vector / charge halos (MODE_QCOPY1/2: pack, send-recv,
   * unpack), the part of them the main stream had to WAIT for (exposed: from the join with the halo stream to the halo's end; the
   * whole halo when overlap is off), scalar all-reduces (qeq.F90:107,129,144,357), reverse force fold (MODE_CPBK) */
  double ms_ghost_build, ms_migrate, ms_halo, ms_halo_exposed, ms_allreduce, ms_fold;
  long long halo_calls, allreduce_calls;
  /* single kernels, HIP events on the engine's stream around each launch, summed like the timers above (bench.py: roofline.kernels):
   * the 10 A sweep, ENbond (PQEq: ENbond_PQEq), E3b, E4b, Ehb, BOPRIM + BOFULL, the assembly gathers of ForceBondedTerms */
  double ms_k_list10, ms_k_nonbond, ms_k_e3b, ms_k_e4b, ms_k_ehb, ms_k_bondorder, ms_k_assemble;
  /* window form of the 10 A matrix (groups of 16 cell-sorted rows whose partners the matrix pass holds in LDS): its build per list build,
   * the number of groups, the largest window in units of 8 cell-sorted positions, 1 when the matrix pass uses it (0: the row pass) */
  double ms_k_winbuild;
  int win_groups, win_max_units, win_in_use, reserved2;
  /* the window pass timed on the first placement of its streams in memory and on the one that was kept (0: no search ran; see
   * Engine::tune_window_placement, RXMD_PLACE_TRIES) */
  double place_ms_first, place_ms_kept;
  /* (round 5, append-only) the placement search itself: placements timed including the first, its wall time, the bytes of device memory it
   * held at its peak beyond the engine's own streams (0 draws: it did not run -- RXMD_PLACE_TRIES=1, a small system, a nearly full device,
   * or the first placement was already within 3 % of the best this process has seen for the same matrix shape) */
  double place_total_ms, place_bytes_held;
  int place_draws;
  /* the instance of the matrix pass the engine dispatched last: k_spmv_win<MODE, STORE, PQ, spmv_nstep, spmv_var> (qeq.hip); 0, 0 = the row pass k_spmv */
  int spmv_nstep, spmv_var, reserved3;
  double ms_k_blist;                    /* the bonded list: sweep + prefix sum + packing into the compact tables (k_bonded_list, k_bond_csr) */
  /* (append-only) the charge-free part of FORCE (bond orders, bonded terms, assembly: ms_bo, ms_bonded, ms_k_e3b ... are its stream times) runs on a
   * stream of its own next to the QEq iterations and ENbond: what of it the main stream had to WAIT for in front of the stress sums, and 1 when it ran so */
  double ms_bond_exposed;
  int bond_overlap;
  long long spmv_launches_timed;        /* (round 6) matrix passes that carried the HIP event pair (RXMD_PASS_TIMING_EVERY, default every 8th in the run-ahead CG loop, every launch elsewhere);
                                         * ms_qeq_spmv = their average duration x spmv_launches */
  int timer_pairs_dropped;              /* event pairs the ms_* timers could not get since rxmd_hip_reset_timers (the pool of 64 ran dry: long rxmd_hip_step calls without a list build or
                                         * a CG loop to collect them): when > 0 the ms_* breakdown under-counts */
} rxmd_stats;
int rxmd_hip_get_stats(rxmd_handle h, rxmd_stats *out);
int rxmd_hip_reset_timers(rxmd_handle h);
/* switch rxmd_config.qeq_mode (0|1) between calls; both modes share every buffer */
int rxmd_hip_set_qeq_mode(rxmd_handle h, int mode);

/* host-side derived tables for unit tests (CUTOFFLENGTH/POTENTIALTABLE, src/init.F90:363-522):
 * which: 0 Evdw 1 dEvdw 2 Eclmb 3 dEclmb 4 Eclmb_QEq  -> out[nboty][5000]; returns nboty */
int rxmd_hip_get_table(rxmd_handle h, int which, double *out, int capacity);
int rxmd_hip_get_cutoffs(rxmd_handle h, double *rc, int capacity, double *maxrc);
/* debugging taps on device state after rxmd_hip_force (residents+ghosts, engine order):
 * what: 0 delta  1 deltap  2 bonded neighbour count  3 real pos (x3)  4 gid  5 type  6 n10 count (residents)
 *       7 hessian row sums (residents)  8 cdbnd gathered per atom  9 charges incl. ghosts
 *       10 ccbnd as ForceBondedTerms consumes it (pot.F90:129-135)  11 window form of the 10 A matrix: entries per resident whose 16-bit slot
 *       leads back to the entry's own position and ghost flag (== what 6 returns; -1 without windows)  100 read-bandwidth probe
 *       102 stripped-down forms of the row pass, 104 the real window pass / row pass back to back: {ms, ms} (experiments, after rxmd_hip_qeq) */
int rxmd_hip_debug_get(rxmd_handle h, int what, double *out, int capacity);

/* ---- multi-rank surface (COPYATOMS, src/comm.F90) ------------------------------------------- */
/* The engine performs the reference's six-direction staged exchange.  With vprocs = 1 1 1 every
 * partner is the rank itself and the exchange is a device-side copy.  For vprocs > 1 the host
 * supplies the transport: `exchange` must deliver `nsend` doubles from device pointer `send` to
 * rank `to` and receive up to `cap` doubles from rank `from` into device pointer `recv`, return the
 * number received; `allreduce_sum` sums n host doubles in place over all ranks.  (RCCL via
 * torch.distributed in bench.py; MPI_Send/Recv + MPI_Allreduce in a Fortran driver.) */
typedef struct rxmd_comm_ops {
  void *ctx;
  long long (*exchange)(void *ctx, int to, const double *send, long long nsend, int from, double *recv, long long cap);
  int (*allreduce_sum)(void *ctx, double *buf, int n);
  /* optional (may be NULL): the same as `exchange` when the receiver already knows it will get exactly `nrecv` doubles
   * (vector halos and force returns re-use the index lists of the ghost build), so no size message is needed.  `send` and
   * `recv` may point INSIDE the exchange buffers: the two stages of an axis are packed back to back and handed over as two
   * consecutive calls (RXMD_NO_STAGE_PAIRS=1 restores one message per buffer) */
  long long (*exchange_known)(void *ctx, int to, const double *send, long long nsend, int from, double *recv, long long nrecv);
} rxmd_comm_ops;
int rxmd_hip_set_comm(rxmd_handle h, const rxmd_comm_ops *ops);
/* Native transport: RCCL send/recv + all-reduce on the engine's own stream (rccl_comm.hip), one communicator per engine.
 * Rank 0 obtains a 128-byte id with rxmd_hip_rccl_unique_id, the host distributes it (MPI_Bcast, torch.distributed, a file)
 * and every rank calls rxmd_hip_comm_init_rccl(h, id, myid, nprocs); a collective call.  Takes precedence over rxmd_comm_ops. */
int rxmd_hip_rccl_unique_id(unsigned char out128[128]);
int rxmd_hip_comm_init_rccl(rxmd_handle h, const unsigned char id128[128], int rank, int world);
/* What a host transport without GPU-aware messaging needs (the MPI callbacks of bindings/rxmd_hip_mod.F90 stage every message
 * through host memory): synchronous copies between the device pointers handed to `exchange` and host buffers, and the number
 * of HIP devices this process sees (a Fortran driver picks `device = myid mod count`). */
int rxmd_hip_copy_to_host(const double *dev, double *host, long long ndoubles);
int rxmd_hip_copy_to_device(double *dev, const double *host, long long ndoubles);
int rxmd_hip_device_count(void);
/* Optional: let the host own the two message buffers (device memory, `ndoubles` each) that `exchange` is called
 * with -- e.g. torch tensors, so that torch.distributed can send them without wrapping foreign pointers. */
int rxmd_hip_set_exchange_buffers(rxmd_handle h, double *send, double *recv, long long ndoubles);
/* Transport self-test with HOST buffers (no GPU needed): every rank sends a pattern round the ring of `nprocs` ranks in
 * both directions with different lengths and all-reduces a vector; returns 0 if the callbacks honour the contract. */
int rxmd_host_comm_selftest(const rxmd_comm_ops *ops, int myid, int nprocs);

/* ---- host front-end helpers (no GPU needed) ------------------------------------------------- */
/* geninit (reference init/geninit.F90:399-575): replicate a fractional-coordinate unit cell mc times,
 * split into vprocs domains, and return this rank's rxff.bin records.  `elem` holds natoms0
 * 2-character element names ("C\0","H\0"...) 4 bytes apart.  Returns the number of atoms of rank
 * `myid` (call with rec10 == NULL to size); lattice_out gets the super-cell lattice. */
long long rxmd_host_geninit(const char *ffield_path, int natoms0, const char *elem4, const double *frac,
                            const double lattice[6], const int mc[3], const int vprocs[3], int myid,
                            double *rec10, long long capacity, double lattice_out[6]);
/* rxff.bin header + this rank's records (ReadBIN); returns natoms of `myid` or <0 */
long long rxmd_host_read_rxff(const char *path, int myid, double lattice_out[6], int vprocs_out[3], double *rec10, long long capacity);
/* ffield parser + derived tables on the host (GETPARAMS src/param.F90:2-375, CUTOFFLENGTH/POTENTIALTABLE
 * src/init.F90:363-522) without touching a GPU.  natoms_per_type[1..nso] selects the types present (index 0 unused).
 * which: 0 Evdw 1 dEvdw 2 Eclmb 3 dEclmb 4 Eclmb_QEq -> out[nboty][5000]; 5 -> out = rc[nboty] then maxrc;
 * 6 -> out = {nso,nboty,nvaty,ntoty,nhbty, chi[1..nso], eta[1..nso], mass[1..nso]}.  Returns nboty or <0. */
int rxmd_host_ffield_table(const char *ffield_path, const long long *natoms_per_type, int which, double *out, long long capacity);
/* geninit's `-lg` / rxmd's `--lg` for the host helpers above (geninit, ffield_table): nonzero = they read ffields in the
 * low-gradient format (init/geninit.F90:233,347).  Process-wide like the reference's module variable; returns the previous value.
 * The engine itself takes the switch per handle (rxmd_config.lg). */
int rxmd_host_ffield_lg(int on);
/* The environment switches of the library (rxmd_amd/csrc/options.def: one table, read once per engine at rxmd_hip_create) as the markdown rows
 * README.md shows: "| `ENV` | default | meaning |" per line, experiments-only switches marked (exp).  Returns the length of the text; copies at
 * most capacity - 1 characters and a terminating 0 into buf (buf may be NULL to size). */
int rxmd_host_describe_options(char *buf, int capacity);
/* library build info: returns 1 if the HIP code object for gfx950 is linked in */
int rxmd_hip_has_device_code(void);

#ifdef __cplusplus
}
#endif
#endif
