"""Host-side mirror of the reference's operator surface for the hot path.

The reference exposes the path as Fortran subroutines `QEq(atype,pos,q)` (src/qeq.F90:2),
`FORCE(atype,pos,f,q)` (src/pot.F90:2) and the MD loop body (src/main.F90:64-98) working on
module-global arrays.  `RxmdEngine` keeps those names and argument meanings on top of the C ABI
(include/rxmd_hip.h); device state lives in the library, numpy arrays are views for the caller.
Errors that make the reference print + MPI_FINALIZE + stop raise RxmdError with the same text.
"""
import ctypes as C
import numpy as np

from . import _lib
from ._lib import RxmdConfig, RxmdStats

PE_NAMES = ["Esystem", "Ebond", "Elp", "Eover", "Eunder", "Eval", "Epen", "Ecoa", "Etors", "Econj", "Ehbond", "Evdwaals", "Ecoulomb", "Echarge"]


class RxmdError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("rxmd_hip error %d: %s" % (code, msg))
        self.code = code


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class RxmdEngine:
    """One engine = one rank of the vprocs grid = one MI355X."""

    def __init__(self, ffield, lattice, vprocs=(1, 1, 1), myid=0, isQEq=1, NMAXQEq=500, QEq_tol=1e-7, qstep=1, dt_fs=0.25,
                 nbuffer=0, maxneighbs=0, maxneighbs10=0, device=0, qeq_mode=0, Lex_fqs=1.0, Lex_k=2.0, pqeq=None, efield=None, lg=False):
        self.L = _lib.load()
        cfg = RxmdConfig()
        self.L.rxmd_hip_default_config(C.byref(cfg))
        self._ff = str(ffield).encode()
        cfg.ffield_path = self._ff
        for i in range(6):
            cfg.lattice[i] = float(lattice[i])
        for i in range(3):
            cfg.vprocs[i] = int(vprocs[i])
        cfg.myid = myid; cfg.isQEq = isQEq; cfg.NMAXQEq = NMAXQEq; cfg.QEq_tol = QEq_tol; cfg.qstep = qstep; cfg.dt_fs = dt_fs
        cfg.nbuffer = nbuffer; cfg.maxneighbs = maxneighbs; cfg.maxneighbs10 = maxneighbs10; cfg.device = device; cfg.qeq_mode = qeq_mode
        cfg.Lex_fqs = Lex_fqs; cfg.Lex_k = Lex_k
        self._pq = str(pqeq).encode() if pqeq else None       # --pqeq <file> (cmdline.F90:112-128): PQEq instead of QEq
        cfg.pqeq_path = self._pq
        cfg.lg = 1 if lg else 0                                # --lg: LG ffield format + low-gradient dispersion tables
        if efield:                                             # (dir 1..3, strength V/A): rxmd.in `efield`
            cfg.efield_dir = int(efield[0]); cfg.efield_strength = float(efield[1])
        self.cfg = cfg
        h = C.c_void_p()
        rc = self.L.rxmd_hip_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise RxmdError(rc, (self.L.rxmd_hip_last_error(None) or b"").decode())
        self.h = h
        self.lattice = [float(x) for x in lattice]

    def close(self):
        if getattr(self, "h", None):
            self.L.rxmd_hip_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc < 0:
            raise RxmdError(rc, self.L.rxmd_hip_last_error(self.h).decode())
        return rc

    # ---- state ----
    def set_atoms_rxff(self, rec10):
        rec10 = np.ascontiguousarray(rec10, np.float64).reshape(-1, 10)
        self._chk(self.L.rxmd_hip_set_atoms_rxff(self.h, len(rec10), _ptr(rec10)))

    def get_atoms_rxff(self):
        n = self._chk(self.L.rxmd_hip_get_atoms_rxff(self.h, None, 0))
        out = np.zeros((n, 10))
        self._chk(self.L.rxmd_hip_get_atoms_rxff(self.h, _ptr(out), n))
        return out

    @property
    def natoms(self):
        return self._chk(self.L.rxmd_hip_get_atoms_rxff(self.h, None, 0))

    def atoms(self):
        n = self.natoms
        gid = np.zeros(n, np.int64); typ = np.zeros(n, np.int32)
        pos = np.zeros((n, 3)); v = np.zeros((n, 3)); f = np.zeros((n, 3)); q = np.zeros(n)
        self._chk(self.L.rxmd_hip_get_atoms(self.h, n, _ptr(gid), _ptr(typ), _ptr(pos), _ptr(v), _ptr(f), _ptr(q)))
        return dict(gid=gid, type=typ, pos=pos, v=v, f=f, q=q)

    def set_charges(self, q):
        q = np.ascontiguousarray(q, np.float64); self._chk(self.L.rxmd_hip_set_charges(self.h, len(q), _ptr(q)))

    def set_velocities(self, v):
        v = np.ascontiguousarray(v, np.float64).reshape(-1, 3); self._chk(self.L.rxmd_hip_set_velocities(self.h, len(v), _ptr(v)))

    def thermostat(self, mdmode, treq=300.0, vsfact=1.0, gke=-1.0):
        """velocity scaling of the MD loop head, main.F90:45-61 (mdmode 4, 5, 7, 8); call every sstep steps"""
        self._chk(self.L.rxmd_hip_thermostat(self.h, int(mdmode), float(treq), float(vsfact), float(gke)))

    def write_xyz(self, path, natoms_total=None):
        """trajectory frame in the reference's WriteXYZ layout (fileio.F90:241-355); with several ranks every rank writes its part
        (natoms_total = global count on rank 0, -1 on the others) and the parts are concatenated in rank order"""
        from . import system
        a = self.atoms()
        txt = system.format_xyz(self.lattice, system.ffield_type_names(self._ff.decode()), a["gid"], a["type"], a["pos"], a["q"],
                                self.shells() if self._pq else None, natoms_total)
        with open(path, "w") as f:
            f.write(txt)

    def bonds(self):
        """bond lists of the residents after the last FORCE (nbrlist / BO(0,:,:), what WriteBND reads): (count[n], partner_gid[n,maxnb], bo[n,maxnb])"""
        n = self.natoms; mx = 32
        cnt = np.zeros(n, np.int32); pg = np.zeros((n, mx), np.int64); bo = np.zeros((n, mx))
        self._chk(self.L.rxmd_hip_get_bonds(self.h, n, mx, _ptr(cnt), _ptr(pg), _ptr(bo)))
        return cnt, pg, bo

    def write_bnd(self, path):
        """bond file of the current frame in the reference's WriteBND layout (fileio.F90:27-148); with several ranks the parts are concatenated in rank order"""
        from . import system
        a = self.atoms(); cnt, pg, bo = self.bonds()
        with open(path, "w") as f:
            f.write(system.format_bnd(a["gid"], a["type"], a["pos"], cnt, pg, bo))

    def write_pdb(self, path, astr=None):
        """pdb frame in the reference's WritePDB layout (fileio.F90:151-238); astr: the stress accumulators whose mean diagonal goes to the last
        column (default: the engine's own since the last energy read)"""
        from . import system
        a = self.atoms()
        if astr is None:
            astr = self.energy()["astr"]
        with open(path, "w") as f:
            f.write(system.format_pdb(system.ffield_type_names(self._ff.decode()), a["gid"], a["type"], a["pos"], a["q"], astr))

    def write_rxff(self, path, current_step=0):
        """checkpoint in the reference's WriteBIN layout (fileio.F90:558-653), single rank"""
        from . import system
        system.write_rxff(path, self.lattice, (1, 1, 1), [self.get_atoms_rxff()], current_step)

    # ---- native multi-GPU transport (include/rxmd_hip.h: rxmd_hip_comm_init_rccl) ----
    def rccl_unique_id(self):
        buf = C.create_string_buffer(128); self._chk(self.L.rxmd_hip_rccl_unique_id(buf)); return buf.raw

    def init_rccl(self, unique_id, rank, world):
        """collective: every rank of the vprocs grid calls it with the id rank 0 generated"""
        self._chk(self.L.rxmd_hip_comm_init_rccl(self.h, C.c_char_p(bytes(unique_id)), int(rank), int(world)))

    def shells(self):
        """PQEq shell displacements spos(natoms,3)"""
        n = self.natoms
        out = np.zeros((n, 3)); rc = self.L.rxmd_hip_get_shells(self.h, _ptr(out), n)
        if rc < 0:
            self._chk(rc)
        return out[:rc]

    def set_shells(self, s):
        s = np.ascontiguousarray(s, np.float64).reshape(-1, 3); self._chk(self.L.rxmd_hip_set_shells(self.h, len(s), _ptr(s)))

    # ---- the hot path (device resident) ----
    def QEq(self):
        """QEq(atype,pos,q), reference src/qeq.F90:2 -> (nstep_qeq, Est)"""
        it = C.c_int(0); est = C.c_double(0.0)
        self._chk(self.L.rxmd_hip_qeq(self.h, C.byref(it), C.byref(est)))
        return it.value, est.value

    def FORCE(self):
        """FORCE(atype,pos,f,q), reference src/pot.F90:2 -> PE(0:13)"""
        pe = np.zeros(14)
        self._chk(self.L.rxmd_hip_force(self.h, _ptr(pe)))
        return pe

    def step(self, nsteps=1):
        """nsteps passes of the MD loop body, reference src/main.F90:64-98"""
        self._chk(self.L.rxmd_hip_step(self.h, int(nsteps)))

    def minimise(self, ftol=1e-4, max_loops=500):
        """mdmode 10, the reference's ConjugateGradient (src/cg.F90:26-98) on the device-resident state -> (CG loops, final PE, QEq+FORCE evaluations)"""
        pe = C.c_double(0); ev = C.c_longlong(0)
        loops = self._chk(self.L.rxmd_hip_minimise(self.h, float(ftol), int(max_loops), C.byref(pe), C.byref(ev)))
        return loops, pe.value, ev.value

    def energy(self):
        ke = C.c_double(0); qs = C.c_double(0); pe = np.zeros(14); astr = np.zeros(6)
        self._chk(self.L.rxmd_hip_get_energy(self.h, C.byref(ke), C.byref(qs), _ptr(pe), _ptr(astr)))
        return dict(KE=ke.value, qsum=qs.value, PE=pe, astr=astr)

    # ---- the reference's own array shapes ----
    def QEq_arrays(self, atype, pos, q, natoms):
        """atype(NBUFFER), pos(NBUFFER,3) column-major (Fortran order), q(NBUFFER) updated in place"""
        nbuf = len(atype)
        posf = np.asfortranarray(pos, np.float64); atype = np.ascontiguousarray(atype, np.float64)
        assert q.flags["C_CONTIGUOUS"] and q.dtype == np.float64
        self._chk(self.L.rxmd_hip_QEq(self.h, nbuf, natoms, _ptr(atype), posf.ctypes.data_as(C.c_void_p), _ptr(q)))
        return q

    def FORCE_arrays(self, atype, pos, q, natoms):
        nbuf = len(atype)
        posf = np.asfortranarray(pos, np.float64); atype = np.ascontiguousarray(atype, np.float64); q = np.ascontiguousarray(q, np.float64)
        f = np.zeros((nbuf, 3), order="F"); pe = np.zeros(14)
        self._chk(self.L.rxmd_hip_FORCE(self.h, nbuf, natoms, _ptr(atype), posf.ctypes.data_as(C.c_void_p), f.ctypes.data_as(C.c_void_p), _ptr(q), _ptr(pe)))
        return f, pe

    # ---- introspection ----
    def stats(self):
        s = RxmdStats(); self._chk(self.L.rxmd_hip_get_stats(self.h, C.byref(s))); return s.asdict()

    def reset_timers(self):
        self._chk(self.L.rxmd_hip_reset_timers(self.h))

    def set_qeq_mode(self, mode):
        self._chk(self.L.rxmd_hip_set_qeq_mode(self.h, int(mode)))

    def table(self, which):
        rc = C.c_double(0)
        nboty = self._chk(self.L.rxmd_hip_get_cutoffs(self.h, None, 0, C.byref(rc)))
        out = np.zeros((nboty, 5000))
        self._chk(self.L.rxmd_hip_get_table(self.h, which, _ptr(out), out.size))
        return out

    def cutoffs(self):
        mx = C.c_double(0)
        n = self._chk(self.L.rxmd_hip_get_cutoffs(self.h, None, 0, C.byref(mx)))
        rc = np.zeros(n)
        self._chk(self.L.rxmd_hip_get_cutoffs(self.h, _ptr(rc), n, C.byref(mx)))
        return rc, mx.value

    def debug(self, what, width=1, cap=None):
        cap = cap or (self.stats()["nbuffer"] * width)
        out = np.zeros(cap)
        n = self._chk(self.L.rxmd_hip_debug_get(self.h, what, _ptr(out), cap))
        out = out[:n * width]
        return out.reshape(n, width) if width > 1 else out
