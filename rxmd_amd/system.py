"""Input-side host helpers that keep the reference's file surface: fractional xyz unit cells,
`geninit` replication (init/geninit.F90), DAT/rxff.bin (src/fileio.F90:444-653), rxmd.in
(src/cmdline.F90:255-297).  The replication and the binary reader run in the C++ front-end."""
import ctypes as C
import numpy as np

from . import _lib


def read_xyz(path):
    """geninit input: line 1 `natoms "note"`, line 2 `a b c alpha beta gamma`, then `Elem x y z` (fractional)"""
    with open(path) as fh:
        lines = fh.read().split("\n")
    n = int(lines[0].split()[0])
    lattice = [float(x) for x in lines[1].split()[:6]]
    names, frac = [], []
    for l in lines[2:2 + n]:
        t = l.split()
        names.append(t[0]); frac.append([float(t[1]), float(t[2]), float(t[3])])
    return names, np.array(frac), lattice


def geninit(ffield, names, frac, lattice, mc=(1, 1, 1), vprocs=(1, 1, 1), myid=0, lg=False):
    """-> (lattice_super, rec10[natoms_of_rank, 10]) exactly what DAT/rxff.bin would hold for `myid`; lg = geninit's -lg (LG ffield format)"""
    L = _lib.load()
    was = L.rxmd_host_ffield_lg(1 if lg else 0)
    try:
        return _geninit(L, ffield, names, frac, lattice, mc, vprocs, myid)
    finally:
        L.rxmd_host_ffield_lg(was)


def _geninit(L, ffield, names, frac, lattice, mc, vprocs, myid):
    n0 = len(names)
    elem = bytearray(4 * n0)
    for i, s in enumerate(names):
        b = s.encode()[:2]
        elem[4 * i:4 * i + len(b)] = b
    elem = bytes(elem)
    frac = np.ascontiguousarray(frac, np.float64)
    lat = np.array(lattice, np.float64); mcv = np.array(mc, np.int32); vp = np.array(vprocs, np.int32)
    lat_out = np.zeros(6)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    n = L.rxmd_host_geninit(str(ffield).encode(), n0, elem, p(frac), p(lat), p(mcv), p(vp), myid, None, 0, p(lat_out))
    if n < 0:
        raise RuntimeError("rxmd_host_geninit failed: %d" % n)
    rec = np.zeros((n, 10))
    n2 = L.rxmd_host_geninit(str(ffield).encode(), n0, elem, p(frac), p(lat), p(mcv), p(vp), myid, p(rec), n, p(lat_out))
    assert n2 == n
    return list(lat_out), rec


def read_rxff(path, myid=0):
    L = _lib.load()
    lat = np.zeros(6); vp = np.zeros(3, np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    n = L.rxmd_host_read_rxff(str(path).encode(), myid, p(lat), p(vp), None, 0)
    if n < 0:
        raise RuntimeError("cannot read %s" % path)
    rec = np.zeros((n, 10))
    L.rxmd_host_read_rxff(str(path).encode(), myid, p(lat), p(vp), p(rec), n)
    return list(lat), [int(x) for x in vp], rec


def write_rxff(path, lattice, vprocs, recs, current_step=0):
    """WriteBIN layout (src/fileio.F90:558-653): int32 nprocs,vprocs[3],natoms[nprocs],step; f64 lattice[6]; records"""
    with open(path, "wb") as fh:
        npr = int(np.prod(vprocs))
        np.array([npr] + list(vprocs) + [len(r) for r in recs] + [current_step], np.int32).tofile(fh)
        np.array(lattice, np.float64).tofile(fh)
        for r in recs:
            np.ascontiguousarray(r, np.float64).tofile(fh)


def format_xyz(lattice, type_names, gid, types, pos, q, shells=None, natoms_total=None):
    """One trajectory frame exactly as the reference's WriteXYZ lays it out (src/fileio.F90:241-355): `i9` atom count, the
    lattice as `3f12.5,3f8.3`, then per atom `a3` name, position, charge, `i9` global id -- `3f12.5` + `f8.3`, or with PQEq
    `3es20.12` + `es20.12` and the shell displacement `3es20.12` -- so util/ and analysis tools of rxmd read it unchanged.
    type_names[t] is the ffield element name of type t (1-based).  Returns the text; ranks concatenate in rank order."""
    n = len(gid) if natoms_total is None else natoms_total
    out = []
    if natoms_total is None or natoms_total >= 0:
        out.append("%9d\n" % n)
        out.append("%12.5f%12.5f%12.5f%8.3f%8.3f%8.3f\n" % tuple(lattice[:6]))
    for i in range(len(gid)):
        nm = " " + type_names[int(types[i])][:2].ljust(2)            # a3 of a character(2) name: right-justified
        if shells is None:
            out.append("%s%12.5f%12.5f%12.5f%8.3f%9d\n" % (nm, pos[i][0], pos[i][1], pos[i][2], q[i], gid[i]))
        else:
            out.append("%s%20.12E%20.12E%20.12E%20.12E%9d%20.12E%20.12E%20.12E\n" % (nm, pos[i][0], pos[i][1], pos[i][2], q[i], gid[i],
                                                                                   shells[i][0], shells[i][1], shells[i][2]))
    return "".join(out)


BND_CUTOFF = 0.3      # bonds below this bond order are not written (fileio.F90:47,113)
USTRS = 6.94728103     # [GPa], module.F90:200


def _fortran_f(width, dec, v):
    """Fortran fw.d: right-justified, all asterisks when the number does not fit"""
    t = "%*.*f" % (width, dec, v)
    if len(t) > width and t.startswith("0."):       # gfortran and flang drop the optional leading zero before they give up
        t = t[1:]
    elif len(t) > width and t.startswith("-0."):
        t = "-" + t[2:]
    return t if len(t) <= width else "*" * width


def format_bnd(gid, types, pos, count, partner_gid, bo):
    """The bond file of one frame as the reference's WriteBND writes it (src/fileio.F90:27-148), the input of its util/ tools: per
    resident `i12.12` global id, `3f12.3` position, `2i3` type and number of bonds listed, then `i12.12` partner id and `f6.3` bond
    order for every bond with BO >= 0.3, the line left-adjusted.  count / partner_gid / bo as rxmd_hip_get_bonds returns them: the
    partners of a line come in the engine's list order, a permutation of the order the reference's linked-list cells produce."""
    out = []
    for i in range(len(gid)):
        sel = [s for s in range(int(count[i])) if not bo[i][s] < BND_CUTOFF]
        line = "%012d %12.3f%12.3f%12.3f %3d%3d" % (gid[i], pos[i][0], pos[i][1], pos[i][2], types[i], len(sel))
        line += "".join(" %012d%s" % (partner_gid[i][s], _fortran_f(6, 3, bo[i][s])) for s in sel)
        out.append(line.strip() + "\n")
    return "".join(out)


def format_pdb(type_names, gid, types, pos, q, astr=(0.0, 0.0, 0.0)):
    """One frame as the reference's WritePDB writes it (src/fileio.F90:151-238): `A6,I5,1x,A2,i12,4x,3f8.3,f6.2,f6.2` -- 'ATOM', a zero,
    the element, the global id, the position, the CHARGE in the temperature-factor column (`tt = q(i)`, :213) and
    sum(astr(1:3))/3*USTRS of the run's stress accumulators in the last column (the same number on every line; asterisks when it
    does not fit f6.2, as the reference prints); 67 characters per line with the new-line."""
    ss = (astr[0] + astr[1] + astr[2]) / 3.0 * USTRS
    out = []
    for i in range(len(gid)):
        nm = type_names[int(types[i])][:2].ljust(2)
        body = "ATOM  %5d %s%12d    %s%s%s%s%s" % (0, nm, gid[i], _fortran_f(8, 3, pos[i][0]), _fortran_f(8, 3, pos[i][1]), _fortran_f(8, 3, pos[i][2]),
                                                  _fortran_f(6, 2, q[i]), _fortran_f(6, 2, ss))
        out.append(body.ljust(66)[:66] + "\n")
    return "".join(out)


def ffield_type_names(ffield, lg=False):
    """element names of the ffield atom types, 1-based list (index 0 unused) -- the reference's atmname (param.F90:103)"""
    L = _lib.load()
    out = np.zeros(4096)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    was = L.rxmd_host_ffield_lg(1 if lg else 0)
    rc = L.rxmd_host_ffield_table(str(ffield).encode(), None, 6, p(out), len(out))
    L.rxmd_host_ffield_lg(was)
    if rc < 0:
        raise RuntimeError("cannot parse %s" % ffield)
    nso = int(out[0])
    names = [""]
    lines = open(ffield).read().split("\n")
    npar = int(lines[1].split()[0])
    base = 2 + npar
    nso2 = int(lines[base].split()[0])
    assert nso2 == nso
    for t in range(nso):
        names.append(lines[base + 4 + (5 if lg else 4) * t].split()[0])
    return names


def parse_rxmd_in(path):
    """keywords of rxmd.in (src/cmdline.F90:255-297); unknown keyword = error, as in the reference"""
    out = {}
    keys = {"mdmode": ["mdmode"], "time": ["dt", "ntime_step"], "temperature": ["treq", "vsfact", "sstep"], "io_step": ["fstep", "pstep"],
            "io_type": ["isBinary", "isBondFile", "isPDB", "isXYZ"], "processors": ["vprocs1", "vprocs2", "vprocs3"],
            "QEq": ["isQEq", "NMAXQEq", "QEq_tol", "qstep"], "exL": ["Lex_fqs", "Lex_k"], "CG_tol": ["ftol"], "efield": ["eFieldDir", "eFieldStrength"],
            "PQEqParm": ["PQEqParmPath"]}
    for line in open(path):
        t = line.strip()
        if not t or t.startswith("#"):
            continue
        tok = t.split()
        if tok[0] not in keys:
            raise ValueError("ERROR: %s is not found" % tok[0])
        for name, val in zip(keys[tok[0]], tok[1:]):
            v = val.replace("d", "e").replace("D", "e")
            if v in (".true.", ".false."):
                out[name] = v == ".true."
            else:
                try:
                    out[name] = int(v)
                except ValueError:
                    try:
                        out[name] = float(v)
                    except ValueError:
                        out[name] = val
    if "vprocs1" in out:
        out["vprocs"] = (out.pop("vprocs1"), out.pop("vprocs2"), out.pop("vprocs3"))
    return out
