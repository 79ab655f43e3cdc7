"""ctypes driver for oracle/liboracle.so (TEST INFRASTRUCTURE -- the product never imports this).

Also holds a numpy restatement of the reference's pre-processor `geninit`
(reference init/geninit.F90:399-527: read fractional xyz, replicate in ix,iy,iz,atom order,
shift by the minimum, wrap, +1e-9, split into vprocs domains) so that tests can build the same
rxff.bin content the reference starts from.
"""
import ctypes as C
import os, subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(HERE, "golden")
INP = os.path.join(GOLD, "inputs")
_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(ROOT, "oracle", "liboracle.so")
        src = os.path.join(ROOT, "oracle", "rxmd_oracle.c")
        if (not os.path.exists(so)) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"])
        L = C.CDLL(so)
        L.rxo_create.restype = C.c_void_p
        L.rxo_create.argtypes = [C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int]
        L.rxo_error.restype = C.c_char_p
        L.rxo_kinetic.restype = C.c_double
        for f in ("rxo_set_atoms", "rxo_init", "rxo_qeq", "rxo_force", "rxo_step", "rxo_natoms", "rxo_nghost_total", "rxo_qeq_iters",
                  "rxo_ntrace", "rxo_get", "rxo_get_bonds"):
            getattr(L, f).restype = C.c_int
        _lib = L
    return _lib


def ffield_names(path, lg=False):
    """atom names in ffield order (geninit.F90 getAtomNames; -lg: five lines per atom block, geninit.F90:233)"""
    lines = open(path).read().split("\n")
    npar = int(lines[1].split()[0])
    nso = int(lines[2 + npar].split()[0])
    base = 2 + npar + 4
    return [lines[base + (5 if lg else 4) * i][1:3].strip() for i in range(nso)]


def read_xyz(path_or_text):
    txt = open(path_or_text).read() if os.path.exists(str(path_or_text)) else str(path_or_text)
    lines = txt.split("\n")
    n = int(lines[0].split()[0])
    lat = [float(x) for x in lines[1].split()[:6]]
    names, fr = [], []
    for l in lines[2:2 + n]:
        t = l.split()
        names.append(t[0]); fr.append([float(t[1]), float(t[2]), float(t[3])])
    return names, np.array(fr), lat


def geninit(names, frac, lattice, ffnames, mc=(1, 1, 1), vprocs=(1, 1, 1)):
    """returns (lattice_super, [per-rank dict(rnorm, type, gid)]) exactly as rxff.bin would hold them"""
    n0 = len(names)
    itype0 = np.array([ffnames.index(s) + 1 for s in names], np.int32)
    mc = np.array(mc)
    ix, iy, iz, ia = np.meshgrid(np.arange(mc[0]), np.arange(mc[1]), np.arange(mc[2]), np.arange(n0), indexing="ij")
    cell = np.stack([ix.ravel(), iy.ravel(), iz.ravel()], 1).astype(np.float64)
    a = ia.ravel()
    pos = (frac[a] + cell) / mc.astype(np.float64)
    typ = itype0[a]
    gid = np.arange(1, len(a) + 1, dtype=np.int64)
    pos = pos - pos.min(axis=0)
    pos = np.fmod(pos, 1.0) + 1e-9
    vp = np.array(vprocs)
    dom = (pos * vp).astype(np.int64)
    sid = dom[:, 0] + dom[:, 1] * vp[0] + dom[:, 2] * vp[0] * vp[1]
    lbox = 1.0 / vp
    ranks = []
    for p in range(int(vp.prod())):
        sel = np.nonzero(sid == p)[0]
        obox = lbox * np.array([p % vp[0], (p // vp[0]) % vp[1], p // (vp[0] * vp[1])])
        ranks.append(dict(rnorm=pos[sel] - obox, type=typ[sel].copy(), gid=gid[sel].copy()))
    lat = [lattice[0] * mc[0], lattice[1] * mc[1], lattice[2] * mc[2]] + list(lattice[3:6])
    return lat, ranks


class Oracle:
    def __init__(self, ffield, lattice, ranks, vprocs=(1, 1, 1), isQEq=1, NMAXQEq=500, QEq_tol=1e-7, dt_fs=0.25,
                 nbuffer=None, maxn10=1500, q0=None, v0=None, pqeq=None, lg=False):
        L = lib()
        self.L = L
        nmax = max(len(r["type"]) for r in ranks)
        if nbuffer is None:
            nbuffer = max(30000, 8 * nmax)
        lat = (C.c_double * 6)(*lattice)
        vp = (C.c_int * 3)(*vprocs)
        L.rxo_global_lg(1 if lg else 0)               # --lg changes the ffield format: known before the file is read
        self.w = L.rxo_create(ffield.encode(), lat, vp, isQEq, NMAXQEq, QEq_tol, dt_fs, nbuffer, maxn10)
        L.rxo_global_lg(0)
        assert self.w, "rxo_create failed"
        self.w = C.c_void_p(self.w)
        if pqeq:                                      # --pqeq <file>: before the tables are built (init.F90:28-43)
            L.rxo_enable_pqeq.restype = C.c_int; L.rxo_enable_pqeq.argtypes = [C.c_void_p, C.c_char_p]
            self._chk(L.rxo_enable_pqeq(self.w, pqeq.encode()))
        self.nranks = len(ranks)
        for p, r in enumerate(ranks):
            n = len(r["type"])
            rn = np.ascontiguousarray(r["rnorm"], np.float64)
            ty = np.ascontiguousarray(r["type"], np.int32)
            gd = np.ascontiguousarray(r["gid"], np.int64)
            q = None if q0 is None else np.ascontiguousarray(q0[p], np.float64)
            v = None if v0 is None else np.ascontiguousarray(v0[p], np.float64)
            rc = L.rxo_set_atoms(self.w, p, n, rn.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p) if v is not None else None,
                                 q.ctypes.data_as(C.c_void_p) if q is not None else None, ty.ctypes.data_as(C.c_void_p), gd.ctypes.data_as(C.c_void_p))
            assert rc == 0
        assert L.rxo_init(self.w) == 0

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError("oracle: " + self.L.rxo_error(self.w).decode())

    def qeq(self):
        self._chk(self.L.rxo_qeq(self.w)); return self.L.rxo_qeq_iters(self.w)

    def force(self):
        self._chk(self.L.rxo_force(self.w))

    def step(self, n=1):
        self._chk(self.L.rxo_step(self.w, n))

    def get(self, what, rank=0, width=1):
        n = self.L.rxo_nghost_total(self.w, rank) if what >= 100 else self.L.rxo_natoms(self.w, rank)
        n = max(n, self.L.rxo_natoms(self.w, rank))
        out = np.zeros(n * width)
        m = self.L.rxo_get(self.w, rank, what, out.ctypes.data_as(C.c_void_p))
        assert m >= 0
        out = out[:m * width]
        return out.reshape(m, width) if width > 1 else out

    def pos(self, rank=0): return self.get(0, rank, 3)
    def vel(self, rank=0): return self.get(1, rank, 3)
    def forces(self, rank=0): return self.get(2, rank, 3)
    def charges(self, rank=0): return self.get(3, rank)
    def spos(self, rank=0): return self.get(8, rank, 3)
    def set_efield(self, direction, strength):
        self.L.rxo_set_efield.argtypes = [C.c_void_p, C.c_int, C.c_double]; self.L.rxo_set_efield(self.w, int(direction), float(strength))

    def set_pqeq_clean(self, flag=1):
        self.L.rxo_set_pqeq_clean.argtypes = [C.c_void_p, C.c_int]; self.L.rxo_set_pqeq_clean(self.w, int(flag))

    def pqeq_stale(self):
        self.L.rxo_pqeq_stale.restype = C.c_longlong; self.L.rxo_pqeq_stale.argtypes = [C.c_void_p]
        return int(self.L.rxo_pqeq_stale(self.w))
    def types(self, rank=0): return self.get(4, rank).astype(np.int32)
    def gids(self, rank=0): return self.get(5, rank).astype(np.int64)

    def energy(self):
        pe = np.zeros(14); self.L.rxo_get_energy(self.w, pe.ctypes.data_as(C.c_void_p)); return pe

    def kinetic(self): return self.L.rxo_kinetic(self.w)

    def thermostat(self, mdmode, treq=300.0, vsfact=1.0, gke=0.0):
        """the velocity scaling the MD loop head applies when mod(nstep,sstep)==0 (main.F90:45-61); gke = KE per atom of the last PRINTE"""
        self.L.rxo_thermostat.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double]
        assert self.L.rxo_thermostat(self.w, int(mdmode), float(treq), float(vsfact), float(gke)) == 0

    def astr(self, reset=True):
        """stress accumulators astr(1:6) (pot.F90:65-72 virial + main.F90:86-94 kinetic), reset like PRINTE"""
        a = np.zeros(6); self.L.rxo_get_astr.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        self.L.rxo_get_astr(self.w, a.ctypes.data_as(C.c_void_p), int(reset)); return a

    def mdbox(self):
        self.L.rxo_mdbox.restype = C.c_double; self.L.rxo_mdbox.argtypes = [C.c_void_p]; return self.L.rxo_mdbox(self.w)

    def trace(self):
        n = self.L.rxo_ntrace(self.w); t = np.zeros((n, 3)); self.L.rxo_get_trace(self.w, t.ctypes.data_as(C.c_void_p)); return t

    def info(self):
        o = np.zeros(32); self.L.rxo_get_info(self.w, o.ctypes.data_as(C.c_void_p)); return o

    def table(self, which):
        nboty = int(self.info()[14]); t = np.zeros((nboty, 5000)); self.L.rxo_get_table(self.w, which, t.ctypes.data_as(C.c_void_p)); return t

    def set_charges(self, q, rank=0):
        q = np.ascontiguousarray(q, np.float64); self.L.rxo_set_charges(self.w, rank, q.ctypes.data_as(C.c_void_p))

    def set_qstep(self, qstep):
        self.L.rxo_set_qstep.argtypes = [C.c_void_p, C.c_int]; self.L.rxo_set_qstep(self.w, int(qstep))

    def set_qeq(self, isQEq, nmax, tol): self.L.rxo_set_qeq(self.w, isQEq, nmax, C.c_double(tol))

    def bonds(self, rank=0):
        G = self.L.rxo_nghost_total(self.w, rank)
        nbr = np.zeros((G, 30), np.int32); bo = np.zeros((G, 30))
        self.L.rxo_get_bonds(self.w, rank, nbr.ctypes.data_as(C.c_void_p), bo.ctypes.data_as(C.c_void_p))
        return nbr, bo


def parse_rxff(buf):
    """rxff.bin bytes (WriteBIN, src/fileio.F90:558-653) -> (lattice, vprocs, step, [records (n,10) per rank])"""
    b = bytes(buf)
    npr = int(np.frombuffer(b[:4], np.int32)[0])
    head = np.frombuffer(b[:4 * (4 + npr + 1)], np.int32)
    vp = tuple(int(x) for x in head[1:4]); nat = [int(x) for x in head[4:4 + npr]]; step = int(head[4 + npr])
    o = 4 * (4 + npr + 1)
    lat = [float(x) for x in np.frombuffer(b[o:o + 48], np.float64)]; o += 48
    recs = []
    for n in nat:
        recs.append(np.frombuffer(b[o:o + 80 * n], np.float64).reshape(n, 10).copy()); o += 80 * n
    return lat, vp, step, recs


def oracle_from_rxff(ffield, buf, **kw):
    """an Oracle started from a restart file exactly as ReadBIN would (src/fileio.F90:444-555)"""
    lat, vp, step, recs = parse_rxff(buf)
    ranks, q0, v0 = [], [], []
    for r in recs:
        ty = np.rint(r[:, 7]).astype(np.int32)
        gid = np.rint((r[:, 7] - ty) * 1e13).astype(np.int64)
        ranks.append(dict(rnorm=r[:, 0:3].copy(), type=ty, gid=gid)); q0.append(r[:, 6].copy()); v0.append(r[:, 3:6].copy())
    o = Oracle(ffield, lat, ranks, vprocs=vp, q0=q0, v0=v0, **kw)
    o.L.rxo_set_lex.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    for p, r in enumerate(recs):
        a = np.ascontiguousarray(r[:, 8]); b = np.ascontiguousarray(r[:, 9])
        o.L.rxo_set_lex(o.w, p, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p))
    return o, recs, lat


def make_system(case):
    """(ffield_path, lattice, ranks) for the named golden case family"""
    if case.startswith("example"):                 # the reference's examples/1-reaxff and 3-reaxpq+: polyethylene, geninit -mc 2 3 5
        ff = os.path.join(INP, "example1", "ffield_pe")
        names, frac, lat = read_xyz(os.path.join(INP, "example1", "pe_cell.xyz"))
        return ff, names, frac, lat
    conf = {"fes": ("conf/fes.xyz", "conf/ffield_fes"), "mos2": ("conf/mos2_ortho.xyz", "conf/ffield_mos2"), "sic512": ("conf/sic.xyz", "ffield_sicnp"),
            "aloslab": ("conf/aloslab.xyz", "conf/ffield_aloslab")}            # more of the reference's conf/ systems
    if case.startswith("mos2_tri"):                # the reference's conf/init.mos2 as shipped: hexagonal cell, gamma = 120 degrees
        names, frac, lat = read_xyz(os.path.join(INP, "conf/mos2_tri.xyz"))
        return os.path.join(INP, "conf/ffield_mos2"), names, frac, lat
    for k, (x, f) in conf.items():
        if case.startswith(k):
            names, frac, lat = read_xyz(os.path.join(INP, x))
            return os.path.join(INP, f), names, frac, lat
    if case.startswith("pbt"):                     # the fractional input the golden was generated from travels inside the fixture
        g = np.load(os.path.join(GOLD, "pbt2272_md2.npz"))
        names, frac, lat = read_xyz(str(g["input_xyz"]))
        return os.path.join(INP, "ffield_rdx"), names, frac, lat
    if case.startswith("rdx168_lg"):
        ff = os.path.join(INP, "ffield_rdx_lg")
        names, frac, lat = read_xyz(os.path.join(INP, "rdx_lg.xyz"))
    elif case.startswith("rdx"):
        ff = os.path.join(INP, "ffield_rdx")
        names, frac, lat = read_xyz(os.path.join(INP, "rdx.xyz"))
    elif case.startswith("ice"):
        ff = os.path.join(INP, "ffield_water")
        g = np.load(os.path.join(GOLD, "ice644_tight.npz"))
        names, frac, lat = read_xyz(str(g["input_xyz"]))
    elif case.startswith("sicfrag"):               # the isolated SiC + O2 cluster in a 40 A box (make_golden.sic_fragment_fractional); input travels in the fixture
        ff = os.path.join(INP, "ffield_sicnp")
        g = np.load(os.path.join(GOLD, "sicfrag26_pqeq_efieldx_md8.npz"))
        names, frac, lat = read_xyz(str(g["input_xyz"]))
    elif case.startswith("sicnp"):
        ff = os.path.join(INP, "ffield_sicnp")
        names, frac, lat = read_xyz(os.path.join(INP, "sicnp.xyz"))
    else:
        raise KeyError(case)
    return ff, names, frac, lat


PQEQ_SICNP = os.path.join(INP, "pqeq_sicnp.in")
PQEQ_EXAMPLE3 = os.path.join(INP, "example3", "pqeq1_example3.par")


# The reference's own published known answer: README.md:157, step-0 line of the 168-atom RDX sample run (per-atom energies as PRINTE
# prints them: GPE es13.5, the groups es11.3).  (value, half a unit of the last printed digit)
README_KNOWN_ANSWER = {
    "GPE": (-9.82464E+01, 0.5e-4), "Ebond": (-1.369E+02, 0.5e-1), "lp+ov+un": (1.287E+00, 0.5e-3), "val+pen+coa": (-1.362E+00, 0.5e-3),
    "tors+conj": (5.208E-01, 0.5e-4), "Ehb": (-1.398E-03, 0.5e-6), "vdW+Clmb+chg": (3.821E+01, 0.5e-2),
}


def check_readme_known_answer(pe, natoms=168):
    """pe = PE(0:13) of the whole system after the pre-loop QEq + FORCE (default rxmd.in: QEq tol 1e-7, q0 = 0)"""
    got = {"GPE": pe[0], "Ebond": pe[1], "lp+ov+un": pe[2:5].sum(), "val+pen+coa": pe[5:8].sum(), "tors+conj": pe[8:10].sum(),
           "Ehb": pe[10], "vdW+Clmb+chg": pe[11:14].sum()}
    for k, (ref, half) in README_KNOWN_ANSWER.items():
        assert abs(got[k] / natoms - ref) <= half * 1.02, (k, got[k] / natoms, ref)
