#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz by running the REAL reference.

TEST INFRASTRUCTURE.  Runs oracle/_ref/{geninit,rxmd} (the unmodified USCCACS/RXMD Fortran
sources compiled by oracle/Makefile, `-DNOMPI -DRFDUMP -DQEQDUMP`) on the data files under
tests/golden/inputs/ and converts the reference's own debug dumps into small npz fixtures:

  rfdump0.txt   gid / pos / force / charge, f20.12     (reference src/pot.F90:76-88)
  qeqdump0.txt  every hessian entry                     (reference src/qeq.F90:75-82)
  stdout        per-iteration QEq trace (qeq.F90:111), `MDstep:` energy line (main.F90:261)

Only this container has oracle/_ref (built from /root/reference); the fixtures travel.
Usage: python tests/golden/make_golden.py [case ...]
"""
import os, re, subprocess, sys, tempfile, shutil
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFBIN = os.path.join(ROOT, "oracle", "_ref")
INP = os.path.join(HERE, "inputs")


def perturbed_ice_fractional(path, seed=12345, sigma=0.02):
    """conf/init.water/ice-1h.xyz holds REAL coordinates and exactly collinear O-H...O triples
    (NaN in the reference): write fractional coords with a seeded Gaussian kick."""
    lines = open(os.path.join(INP, "ice-1h_real.xyz")).read().split("\n")
    n = int(lines[0].split()[0])
    lat = [float(x) for x in lines[1].split()[:6]]
    rng = np.random.default_rng(seed)
    out = [lines[0], lines[1]]
    for l in lines[2:2 + n]:
        e, x, y, z = l.split()[:4]
        r = np.array([float(x), float(y), float(z)]) + rng.normal(0.0, sigma, 3)
        fr = r / np.array(lat[:3])
        out.append("%s %.12f %.12f %.12f" % (e, fr[0], fr[1], fr[2]))
    open(path, "w").write("\n".join(out) + "\n")


def real_to_fractional(src, dst):
    """conf/init.a-polys/*.xyz hold REAL coordinates of an orthorhombic box: fractional = r / (a, b, c), 12 decimals"""
    lines = open(src).read().split("\n")
    n = int(lines[0].split()[0]); lat = [float(x) for x in lines[1].split()[:6]]
    out = [lines[0], lines[1]]
    for l in lines[2:2 + n]:
        e, x, y, z = l.split()[:4]
        out.append("%s %.12f %.12f %.12f" % (e, float(x) / lat[0], float(y) / lat[1], float(z) / lat[2]))
    open(dst, "w").write("\n".join(out) + "\n")


def sic_fragment_fractional(path, box=40.0, radius=5.8):
    """An isolated cluster cut out of conf/init.sicnp (SiC nanoparticle in O2): the atoms within `radius` of the midpoint between one
    O atom and its nearest nanoparticle atom, O2 molecules kept whole -- the candidate whose cluster has a diameter below 11.8 A
    (under the 12.5 A PQEq cut-off minus every shell displacement) and the most O atoms: 8 Si + 8 C + 5 O2 = 26 atoms, diameter
    11.1 A.  Centred in a cubic box of 40 A, so that no pair -- core or shell, own image or neighbour image -- lies anywhere near
    the cut-off: the reference's beyond-cut-off table look-ups (module.F90:401, which leave stale values behind) cannot occur,
    and the trajectory of the reference is a clean PQEq trajectory."""
    lines = open(os.path.join(INP, "sicnp.xyz")).read().split("\n")
    n = int(lines[0].split()[0]); lat = [float(x) for x in lines[1].split()[:6]]
    names = np.array([l.split()[0] for l in lines[2:2 + n]])
    pos = np.array([[float(t) for t in l.split()[1:4]] for l in lines[2:2 + n]]) * np.array(lat[:3])
    O = np.where(names == "O")[0]; NPa = np.where(names != "O")[0]
    D = np.linalg.norm(pos[O][:, None] - pos[NPa][None], axis=2)
    best = None
    for oi in np.argsort(D.min(1), kind="stable")[:40]:
        c = 0.5 * (pos[O[oi]] + pos[NPa[np.argmin(D[oi])]])
        keep = set(np.where(np.linalg.norm(pos - c, axis=1) < radius)[0])
        for i in list(keep):
            if names[i] == "O":
                dd = np.linalg.norm(pos[O] - pos[i], axis=1); o2 = np.argsort(dd, kind="stable")
                if dd[o2[1]] < 1.5:
                    keep.add(O[o2[1]])
        idx = np.array(sorted(keep)); P = pos[idx]
        diam = np.linalg.norm(P[:, None] - P[None], axis=2).max()
        nO = int((names[idx] == "O").sum()); nSi = int((names[idx] == "Si").sum())
        if diam < 11.8 and nO >= 4 and nSi >= 8 and (best is None or (nO, len(idx)) > best[0]):
            best = ((nO, len(idx)), idx, diam)
    idx = best[1]; P = pos[idx] - 0.5 * (pos[idx].min(0) + pos[idx].max(0)) + 0.5 * box
    out = ["%d \"SiC fragment + O2 cut from conf/init.sicnp\"" % len(idx), "%.3f %.3f %.3f 90.000 90.000 90.000" % (box, box, box)]
    for e, r in zip(names[idx], P / box):
        out.append("%s %.12f %.12f %.12f" % (e, r[0], r[1], r[2]))
    open(path, "w").write("\n".join(out) + "\n")
    return len(idx), best[2]


CASES = {
    # name: (xyz, ffield, mc, extra rxmd flags, nsteps for the dump run)
    "rdx168_tol7":   ("rdx.xyz", "ffield_rdx", (1, 1, 1), [], 0),
    "rdx168_tight":  ("rdx.xyz", "ffield_rdx", (1, 1, 1), ["--QEq_tol", "1e-12", "--NMAXQEq", "2000"], 0),
    "rdx168_noqeq":  ("rdx.xyz", "ffield_rdx", (1, 1, 1), ["--isQEq", "0"], 0),
    "rdx168_md10":   ("rdx.xyz", "ffield_rdx", (1, 1, 1), [], 10),
    "rdx222_tight":  ("rdx.xyz", "ffield_rdx", (2, 2, 2), ["--QEq_tol", "1e-12", "--NMAXQEq", "2000"], 0),
    # extended-Lagrangian charges (isQEq 2: one CG step per MD step from the fictitious charges, qeq.F90:51-57, main.F90:67-68,98)
    "rdx168_lex_md10": ("rdx.xyz", "ffield_rdx", (1, 1, 1), ["--isQEq", "2"], 10),
    "rdx222_md5":    ("rdx.xyz", "ffield_rdx", (2, 2, 2), [], 5),
    # (round 6) 110 steps at the default tolerance 1e-7: the reference's own iteration count per step above one cell -- the statistic a change of the
    # CG's rounding is judged against (the count of a single step is REAL(4) noise; the mean over 100 steps is not)
    "rdx222_md110":  ("rdx.xyz", "ffield_rdx", (2, 2, 2), [], 110),
    "rdx222_md1000": ("rdx.xyz", "ffield_rdx", (2, 2, 2), [], 1000),     # the same over 1000 steps: 110 calls are too few for the tails (runs of one-iteration exits)
    "ice644_tight":  ("ICE", "ffield_water", (6, 4, 4), ["--QEq_tol", "1e-12", "--NMAXQEq", "2000"], 0),
    # PQEq (pqeq.F90): SiC nanoparticle in O2, conf/init.sicnp, polarizable shells; the --pqeq file is copied next to the run
    "sicnp547_pqeq_tol7":  ("sicnp.xyz", "ffield_sicnp", (1, 1, 1), ["--pqeq", "pqeq.in"], 0),
    "sicnp547_pqeq_tight": ("sicnp.xyz", "ffield_sicnp", (1, 1, 1), ["--pqeq", "pqeq.in", "--QEq_tol", "1e-12", "--NMAXQEq", "2000"], 0),
    "sicnp547_pqeq_md5":   ("sicnp.xyz", "ffield_sicnp", (1, 1, 1), ["--pqeq", "pqeq.in"], 5),
    # external electric field on cores and shells (EEfield module.F90:359-383, pqeq.F90:205, momentum removal main.F90:70-71)
    "sicnp547_pqeq_efieldx_0": ("sicnp.xyz", "ffield_sicnp", (1, 1, 1), ["--pqeq", "pqeq.in", "--QEq_tol", "1e-12", "--NMAXQEq", "2000", "RXMDIN:efield 1 0.05"], 0),
    "sicnp547_pqeq_efieldx_md3": ("sicnp.xyz", "ffield_sicnp", (1, 1, 1), ["--pqeq", "pqeq.in", "--QEq_tol", "1e-12", "--NMAXQEq", "2000", "RXMDIN:efield 1 0.05"], 3),
    "sicnp547_pqeq_efield_md3": ("sicnp.xyz", "ffield_sicnp", (1, 1, 1), ["--pqeq", "pqeq.in", "--QEq_tol", "1e-12", "--NMAXQEq", "2000", "RXMDIN:efield 3 0.05"], 3),
    # PQEq MD on an isolated cluster in a 40 A box (sic_fragment_fractional above): no beyond-cut-off look-up can happen, the reference's
    # own trajectory is the clean one -- 8 steps with the field along x, shells moving
    "sicfrag26_pqeq_efieldx_md8": ("SICFRAG", "ffield_sicnp", (1, 1, 1), ["--pqeq", "pqeq.in", "--QEq_tol", "1e-12", "--NMAXQEq", "2000", "RXMDIN:efield 1 0.05"], 8),
    # low-gradient dispersion correction (--lg: five-line atom blocks and a C_lg column in the ffield, param.F90:83-86,107-109,197-200;
    # table terms init.F90:496-514); conf/init.rdx.lg/{input.xyz,ffield}
    "rdx168_lg_tight": ("rdx_lg.xyz", "ffield_rdx_lg", (1, 1, 1), ["--lg", "--QEq_tol", "1e-12", "--NMAXQEq", "2000"], 0),
    "rdx168_lg_md5":   ("rdx_lg.xyz", "ffield_rdx_lg", (1, 1, 1), ["--lg"], 5),
    # other systems and force fields the reference ships under conf/ (10-, 4-, 7- and 5-type ffields), 3 MD steps at tight tolerance
    "fes576_md3":     ("conf/fes.xyz", "conf/ffield_fes", (2, 2, 2), ["--QEq_tol", "1e-12", "--NMAXQEq", "2000"], 3),
    "mos2_216_md3":   ("conf/mos2_ortho.xyz", "conf/ffield_mos2", (2, 1, 1), ["--QEq_tol", "1e-12", "--NMAXQEq", "2000"], 3),
    # the one non-orthorhombic input the reference ships: conf/init.mos2, 2H-MoS2 in its hexagonal cell (gamma = 120 degrees); 3 x 3 x 2 so that
    # every box edge exceeds twice the 10 A cutoff.  Exercises the full H / HHi transforms (main.F90:596-616, init.F90:636-668) and the reference's
    # cell meshes, which are laid out in lattice-vector units (init.F90:530-605)
    "mos2_tri324_md3": ("conf/mos2_tri.xyz", "conf/ffield_mos2", (3, 3, 2), ["--QEq_tol", "1e-12", "--NMAXQEq", "2000"], 3),
    "sic512_md3":     ("conf/sic.xyz", "ffield_sicnp", (4, 4, 4), ["--QEq_tol", "1e-12", "--NMAXQEq", "2000"], 3),
    "aloslab180_md3": ("conf/aloslab.xyz", "conf/ffield_aloslab", (3, 2, 1), ["--QEq_tol", "1e-12", "--NMAXQEq", "2000"], 3),
    # an amorphous polymer cell of the reference's conf/init.a-polys (poly(butylene terephthalate), 2,272 atoms, real coordinates -> fractional
    # by real_to_fractional above) with the top-level ffield: aromatic rings, esters -- every bonded term on a disordered structure
    "pbt2272_md2":    ("PBT", "ffield_rdx", (1, 1, 1), ["--QEq_tol", "1e-12", "--NMAXQEq", "2000"], 2),
    # charges re-equilibrated every third step only (rxmd.in QEq ... <qstep>, main.F90:77)
    "rdx168_qstep3_md7": ("rdx.xyz", "ffield_rdx", (1, 1, 1), ["RXMDIN:QEq 1 2000 1.d-12 3"], 7),
    # multi-rank (real MPI build oracle/_ref/rxmd_mpi, conda MPICH): name: (..., vprocs)
    "rdx222_v211_tight": ("rdx.xyz", "ffield_rdx", (2, 2, 2), ["--QEq_tol", "1e-12", "--NMAXQEq", "2000"], 0, (2, 1, 1)),
    "rdx222_v222_tight": ("rdx.xyz", "ffield_rdx", (2, 2, 2), ["--QEq_tol", "1e-12", "--NMAXQEq", "2000"], 0, (2, 2, 2)),
    "rdx222_v222_md3":   ("rdx.xyz", "ffield_rdx", (2, 2, 2), ["--QEq_tol", "1e-12", "--NMAXQEq", "2000"], 3, (2, 2, 2)),
    # the reference's examples/2-reaxff-dc: polyethylene, geninit -mc 4 3 5 -v 2 1 1, mpirun -np 2
    "example2_v211_md3": ("example1/pe_cell.xyz", "example1/ffield_pe", (4, 3, 5), ["--QEq_tol", "1e-12", "--NMAXQEq", "2000"], 3, (2, 1, 1)),
    # (round 6) the modes beyond plain isQEq 1 BETWEEN DISTINCT RANKS.  PQEq under MPI: three more doubles per COPY / MOVE record and the
    # shell halo (comm.F90:122,129-131,149-185; pqeq.F90:2-182) -- the SiC nanoparticle replicated so that every rank owns one particle;
    # step 0 at tight tolerance (every shell on its core: no beyond-cut-off look-up, the reference's numbers are clean PQEq numbers)
    "sicnp211_v211_pqeq_tight": ("sicnp.xyz", "ffield_sicnp", (2, 1, 1), ["--pqeq", "pqeq.in", "--QEq_tol", "1e-12", "--NMAXQEq", "2000"], 0, (2, 1, 1)),
    "sicnp222_v222_pqeq_tight": ("sicnp.xyz", "ffield_sicnp", (2, 2, 2), ["--pqeq", "pqeq.in", "--QEq_tol", "1e-12", "--NMAXQEq", "2000"], 0, (2, 2, 2)),
    # ... and 3 MD steps of the reference itself (faithful oracle mode reproduces them; the engine is compared with the clean oracle)
    "sicnp211_v211_pqeq_md3": ("sicnp.xyz", "ffield_sicnp", (2, 1, 1), ["--pqeq", "pqeq.in", "--QEq_tol", "1e-12", "--NMAXQEq", "2000"], 3, (2, 1, 1)),
    # the field along x on two ranks: LinearMomentum's all-reduces every step (main.F90:70-71,766-797)
    "sicnp211_v211_pqeq_efieldx_md3": ("sicnp.xyz", "ffield_sicnp", (2, 1, 1), ["--pqeq", "pqeq.in", "--QEq_tol", "1e-12", "--NMAXQEq", "2000", "RXMDIN:efield 1 0.05"], 3, (2, 1, 1)),
    # extended-Lagrangian charges on two ranks, 10 steps: qsfp / qsfv migrate with their atom (comm.F90:159-163)
    "rdx222_v211_lex_md10": ("rdx.xyz", "ffield_rdx", (2, 2, 2), ["--isQEq", "2"], 10, (2, 1, 1)),
}


def run(cmd, cwd):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    p = subprocess.run(cmd, cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if p.returncode != 0 or "successfully finished" not in p.stdout:
        raise RuntimeError("%s failed:\n%s" % (cmd, p.stdout[-3000:]))
    return p.stdout


def parse_rfdump(path):
    gid, typ, pos, frc, chg = [], [], [], [], []
    for l in open(path):
        t = l.split()
        if t[1] == "pos":
            gid.append(int(t[0])); typ.append(int(t[2])); pos.append([float(x) for x in t[3:6]])
        elif t[1] == "frc":
            frc.append([float(x) for x in t[3:6]])
        elif t[1] == "chg":
            chg.append(float(t[3]))
    return (np.array(gid, np.int64), np.array(typ, np.int32), np.array(pos), np.array(frc), np.array(chg))


def parse_trace(out):
    """lines printed by qeq.F90:111 :  iter  log|gs| log|gt|  Est  Est_old  qsum  (last QEq call)"""
    rows, cur = [], []
    for l in out.split("\n"):
        t = l.split()
        if len(t) == 6 and re.fullmatch(r"\d+", t[0]):
            try:
                v = [float(x) for x in t[1:]]
            except ValueError:
                continue
            if int(t[0]) == 0 and cur:
                rows.append(cur); cur = []
            cur.append([int(t[0])] + v)
    if cur:
        rows.append(cur)
    return rows


def parse_mdstep(out):
    res = []
    for l in out.split("\n"):
        if l.startswith("MDstep:"):
            t = l.split()
            res.append([float(x) for x in t[1:16]])
    return np.array(res)


def make_mpi(name):
    xyz, ff, mc, flags, nsteps, vp = CASES[name]
    extra_in = [f[7:] for f in flags if f.startswith("RXMDIN:")]     # extra rxmd.in lines (see make())
    flags = [f for f in flags if not f.startswith("RXMDIN:")]
    npr = vp[0] * vp[1] * vp[2]
    tmp = tempfile.mkdtemp(prefix="golden_")
    try:
        os.makedirs(os.path.join(tmp, "DAT"))
        shutil.copy(os.path.join(INP, xyz), os.path.join(tmp, "input.xyz"))
        shutil.copy(os.path.join(INP, ff), os.path.join(tmp, "ffield"))
        shutil.copy(os.path.join(INP, "rxmd.in"), os.path.join(tmp, "rxmd.in"))
        shutil.copy(os.path.join(INP, "pqeq_sicnp.in"), os.path.join(tmp, "pqeq.in"))
        if extra_in:
            with open(os.path.join(tmp, "rxmd.in"), "a") as f:
                f.write("\n".join(extra_in) + "\n")
        run([os.path.join(REFBIN, "geninit"), "-i", "input.xyz", "-f", "ffield", "-o", "DAT", "-mc", str(mc[0]), str(mc[1]), str(mc[2]),
             "-v", str(vp[0]), str(vp[1]), str(vp[2])], tmp)
        out = run(["/opt/conda/bin/mpiexec", "-np", str(npr), os.path.join(REFBIN, "rxmd_mpi"), "--ntime_step", str(nsteps), "--pstep", "1",
                   "--fstep", "100000", "--vprocs", str(vp[0]), str(vp[1]), str(vp[2])] + flags, tmp)
        d = dict(mc=np.array(mc), vprocs=np.array(vp), nsteps=np.array(nsteps), flags=np.array(" ".join(flags)))
        for r in range(npr):
            gid, typ, pos, frc, chg = parse_rfdump(os.path.join(tmp, "rfdump%d.txt" % r))
            d["gid_%d" % r] = gid; d["type_%d" % r] = typ; d["pos_%d" % r] = pos; d["force_%d" % r] = frc; d["charge_%d" % r] = chg
        traces = parse_trace(out)
        d["qeq_iters"] = np.array([len(t) - 1 for t in traces])
        d["mdstep"] = parse_mdstep(out)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
        print(name, "ranks", npr, "natoms/rank", [len(d["gid_%d" % r]) for r in range(npr)], "qeq_iters", d["qeq_iters"])
    finally:
        shutil.rmtree(tmp)


def make(name):
    if len(CASES[name]) == 6:
        return make_mpi(name)
    xyz, ff, mc, flags, nsteps = CASES[name]
    extra_in = [f[7:] for f in flags if f.startswith("RXMDIN:")]     # extra rxmd.in lines (the --efield command-line form does not parse under flang)
    flags = [f for f in flags if not f.startswith("RXMDIN:")]
    tmp = tempfile.mkdtemp(prefix="golden_")
    try:
        os.makedirs(os.path.join(tmp, "DAT"))
        if xyz == "ICE":
            perturbed_ice_fractional(os.path.join(tmp, "input.xyz"))
        elif xyz == "SICFRAG":
            print("SiC fragment:", sic_fragment_fractional(os.path.join(tmp, "input.xyz")))
        elif xyz == "PBT":
            real_to_fractional(os.path.join(INP, "conf", "PBT_real.xyz"), os.path.join(tmp, "input.xyz"))
        else:
            shutil.copy(os.path.join(INP, xyz), os.path.join(tmp, "input.xyz"))
        shutil.copy(os.path.join(INP, ff), os.path.join(tmp, "ffield"))
        shutil.copy(os.path.join(INP, "rxmd.in"), os.path.join(tmp, "rxmd.in"))
        shutil.copy(os.path.join(INP, "pqeq_sicnp.in"), os.path.join(tmp, "pqeq.in"))
        if extra_in:
            with open(os.path.join(tmp, "rxmd.in"), "a") as f:
                f.write("\n".join(extra_in) + "\n")
        run([os.path.join(REFBIN, "geninit"), "-i", "input.xyz", "-f", "ffield", "-o", "DAT",
             "-mc", str(mc[0]), str(mc[1]), str(mc[2])] + (["-lg"] if "--lg" in flags else []), tmp)
        rxffbin = open(os.path.join(tmp, "DAT", "rxff.bin"), "rb").read()
        # run A: the dump run
        out = run([os.path.join(REFBIN, "rxmd"), "--ntime_step", str(nsteps), "--pstep", "1",
                   "--fstep", "100000"] + flags, tmp)
        gid, typ, pos, frc, chg = parse_rfdump(os.path.join(tmp, "rfdump0.txt"))
        traces = parse_trace(out)
        md = parse_mdstep(out)
        d = dict(gid=gid, type=typ, pos=pos, force=frc, charge=chg, mc=np.array(mc),
                 nsteps=np.array(nsteps), flags=np.array(" ".join(flags)))
        if traces:
            d["qeq_iters"] = np.array([len(t) - 1 for t in traces])   # last printed row is the exit row
            d["qeq_trace_last"] = np.array(traces[-1])
            d["qeq_trace_first"] = np.array(traces[0])
        # hessian summary of the LAST QEq call (file is rewritten per call)
        qd = os.path.join(tmp, "qeqdump0.txt")
        if os.path.exists(qd) and os.path.getsize(qd) > 0:
            h = np.loadtxt(qd, usecols=(1, 3, 4))
            gi = h[:, 0].astype(np.int64); gj = h[:, 1].astype(np.int64); hv = h[:, 2]
            nn = np.bincount(gi, minlength=gid.max() + 1)
            rs = np.bincount(gi, weights=hv, minlength=gid.max() + 1)
            d["hess_nnz"] = nn[1:]; d["hess_rowsum"] = rs[1:]
            sel = np.linspace(0, len(hv) - 1, 400).astype(int)
            d["hess_sample"] = np.stack([gi[sel], gj[sel], hv[sel]], 1)
        # run B: energies of step 0 .. (printed by PRINTE before each step)
        outB = run([os.path.join(REFBIN, "rxmd"), "--ntime_step", str(max(nsteps, 1)), "--pstep", "1",
                    "--fstep", "100000"] + flags, tmp)
        d["mdstep"] = parse_mdstep(outB)
        if name.startswith("ice") or name.startswith("pbt") or name.startswith("sicfrag"):
            d["input_xyz"] = np.array(open(os.path.join(tmp, "input.xyz")).read())
        if name in ("rdx168_md10", "sicnp547_pqeq_md5", "sicfrag26_pqeq_efieldx_md8"):
            # the reference's own trajectory frame of the last step (OUTPUT -> WriteXYZ, fileio.F90:241-355): output data, kept as text
            # (+ the bond file and the pdb frame of the same step: WriteBND fileio.F90:27-148, WritePDB :151-238)
            outC = run([os.path.join(REFBIN, "rxmd"), "--ntime_step", str(nsteps), "--pstep", "1", "--fstep", str(nsteps), "--isBondFile", "--isPDB"] + flags, tmp)
            d["xyz_last"] = np.array(open(os.path.join(tmp, "DAT", "%09d.xyz" % nsteps)).read())
            for ext in ("bnd", "pdb"):
                fn = os.path.join(tmp, "DAT", "%09d.%s" % (nsteps, ext))
                if os.path.exists(fn):
                    d[ext + "_last"] = np.array(open(fn, errors="replace").read())
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
        print(name, "natoms", len(gid), "qeq_iters", d.get("qeq_iters"), "PE/atom", d["mdstep"][0][2] if len(d["mdstep"]) else None)
    finally:
        shutil.rmtree(tmp)


THERMO = {   # name: rxmd flags of the restart run (7 steps, thermostat action every 3rd step, PRINTE every step so that GKE is current)
    "rdx168_thermo4": ["--mdmode", "4", "--vsfact", "0.9"],
    "rdx168_thermo5": ["--mdmode", "5", "--treq", "300"],
    "rdx168_thermo7": ["--mdmode", "7", "--treq", "300"],
    "rdx168_thermo8": ["--mdmode", "8", "--treq", "300"],
}


def make_thermo():
    """velocity-scaling modes of the MD loop head (main.F90:45-61): 20 NVE steps from rest produce a restart file with
    velocities (the reference's own DAT/rxff.bin), every mode then continues from it for 7 steps with sstep 3."""
    tmp = tempfile.mkdtemp(prefix="golden_")
    try:
        os.makedirs(os.path.join(tmp, "DAT"))
        shutil.copy(os.path.join(INP, "rdx.xyz"), os.path.join(tmp, "input.xyz"))
        shutil.copy(os.path.join(INP, "ffield_rdx"), os.path.join(tmp, "ffield"))
        shutil.copy(os.path.join(INP, "rxmd.in"), os.path.join(tmp, "rxmd.in"))
        run([os.path.join(REFBIN, "geninit"), "-i", "input.xyz", "-f", "ffield", "-o", "DAT", "-mc", "1", "1", "1"], tmp)
        # --isBinary: WriteBIN returns early without it (fileio.F90:585); the real-MPI build on one rank does the file I/O
        mpi = ["/opt/conda/bin/mpiexec", "-np", "1", os.path.join(REFBIN, "rxmd_mpi")]
        tight = ["--QEq_tol", "1e-12", "--NMAXQEq", "2000"]     # converged charges: the trajectory is then reproducible to 1e-9
        run(mpi + ["--ntime_step", "20", "--pstep", "1", "--fstep", "100000", "--isBinary"] + tight, tmp)
        restart = open(os.path.join(tmp, "DAT", "rxff.bin"), "rb").read()
        for name, flags in THERMO.items():
            open(os.path.join(tmp, "DAT", "rxff.bin"), "wb").write(restart)
            run(mpi + ["--ntime_step", "7", "--pstep", "1", "--fstep", "100000", "--sstep", "3", "--isBinary"] + tight + flags, tmp)
            gid, typ, pos, frc, chg = parse_rfdump(os.path.join(tmp, "rfdump0.txt"))
            final = np.frombuffer(open(os.path.join(tmp, "DAT", "rxff.bin"), "rb").read(), np.uint8)
            np.savez_compressed(os.path.join(HERE, name + ".npz"), gid=gid, type=typ, pos=pos, force=frc, charge=chg, flags=np.array(" ".join(flags)),
                                restart_rxff=np.frombuffer(restart, np.uint8), final_rxff=final)
            print(name, "natoms", len(gid))
    finally:
        shutil.rmtree(tmp)


def make_thermo_mpi():
    """(round 6) the same velocity-scaling modes BETWEEN TWO RANKS: RDX 2 x 2 x 2 on vprocs 2 1 1 under real MPI.  The sums of
    ScaleTemperature / AdjustTemperature / LinearMomentum and the kinetic energy of mode 5 are MPI_ALLREDUCEd (main.F90:699,738,783,
    PRINTE :225-244).  20 NVE steps from rest write the restart file (both ranks' records in one rxff.bin), every mode continues
    from it for 7 steps with sstep 3; per-rank dumps of the last step."""
    tmp = tempfile.mkdtemp(prefix="golden_")
    try:
        os.makedirs(os.path.join(tmp, "DAT"))
        shutil.copy(os.path.join(INP, "rdx.xyz"), os.path.join(tmp, "input.xyz"))
        shutil.copy(os.path.join(INP, "ffield_rdx"), os.path.join(tmp, "ffield"))
        shutil.copy(os.path.join(INP, "rxmd.in"), os.path.join(tmp, "rxmd.in"))
        run([os.path.join(REFBIN, "geninit"), "-i", "input.xyz", "-f", "ffield", "-o", "DAT", "-mc", "2", "2", "2", "-v", "2", "1", "1"], tmp)
        mpi = ["/opt/conda/bin/mpiexec", "-np", "2", os.path.join(REFBIN, "rxmd_mpi"), "--vprocs", "2", "1", "1"]
        tight = ["--QEq_tol", "1e-12", "--NMAXQEq", "2000"]
        run(mpi + ["--ntime_step", "20", "--pstep", "1", "--fstep", "100000", "--isBinary"] + tight, tmp)
        restart = open(os.path.join(tmp, "DAT", "rxff.bin"), "rb").read()
        for name, flags in THERMO.items():
            open(os.path.join(tmp, "DAT", "rxff.bin"), "wb").write(restart)
            run(mpi + ["--ntime_step", "7", "--pstep", "1", "--fstep", "100000", "--sstep", "3", "--isBinary"] + tight + flags, tmp)
            d = dict(flags=np.array(" ".join(flags)), restart_rxff=np.frombuffer(restart, np.uint8), vprocs=np.array((2, 1, 1)),
                     final_rxff=np.frombuffer(open(os.path.join(tmp, "DAT", "rxff.bin"), "rb").read(), np.uint8))
            for r in range(2):
                gid, typ, pos, frc, chg = parse_rfdump(os.path.join(tmp, "rfdump%d.txt" % r))
                d["gid_%d" % r] = gid; d["type_%d" % r] = typ; d["pos_%d" % r] = pos; d["force_%d" % r] = frc; d["charge_%d" % r] = chg
            out = name.replace("rdx168_", "rdx222_v211_")
            np.savez_compressed(os.path.join(HERE, out + ".npz"), **d)
            print(out, "natoms/rank", [len(d["gid_%d" % r]) for r in range(2)])
    finally:
        shutil.rmtree(tmp)


def make_minimiser():
    """mdmode 10: the reference's geometry minimiser (src/cg.F90: bracket, golden-section line minimisation, Polak-Ribiere) on
    RDX-168; it calls QEq + FORCE some sixty times and writes DAT/000000000.xyz when the energy has converged."""
    import resource
    tmp = tempfile.mkdtemp(prefix="golden_")
    try:
        os.makedirs(os.path.join(tmp, "DAT"))
        shutil.copy(os.path.join(INP, "rdx.xyz"), os.path.join(tmp, "input.xyz"))
        shutil.copy(os.path.join(INP, "ffield_rdx"), os.path.join(tmp, "ffield"))
        shutil.copy(os.path.join(INP, "rxmd.in"), os.path.join(tmp, "rxmd.in"))
        with open(os.path.join(tmp, "rxmd.in"), "a") as f:
            f.write("CG_tol 1.d-5\n")
        run([os.path.join(REFBIN, "geninit"), "-i", "input.xyz", "-f", "ffield", "-o", "DAT", "-mc", "1", "1", "1"], tmp)
        p = subprocess.run([os.path.join(REFBIN, "rxmd"), "--mdmode", "10", "--QEq_tol", "1e-12", "--NMAXQEq", "2000"], cwd=tmp,
                           env=dict(os.environ, OMP_NUM_THREADS="1"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                           preexec_fn=lambda: resource.setrlimit(resource.RLIMIT_STACK, (resource.RLIM_INFINITY, resource.RLIM_INFINITY)))
        assert "successfully finished structural optimization" in p.stdout, p.stdout[-2000:]
        xyz = open(os.path.join(tmp, "DAT", "000000000.xyz")).read()
        # the line search the reference prints on the way (cg.F90:122-124,262-263): bracket, then per golden-section round ax bx cx dx PEbx PEcx
        bracket = [float(l.split()[-1]) for l in p.stdout.split("\n") if "bracket has been found" in l]
        gs = np.array([[float(x) for x in l.split()[1:7]] for l in p.stdout.split("\n") if l.strip().startswith("ax,bx,cx,dx,PEbx,PEcx:")])
        np.savez_compressed(os.path.join(HERE, "rdx168_minimiser.npz"), xyz=np.array(xyz), stdout_tail=np.array(p.stdout[-1500:]),
                            bracket=np.array(bracket), golden_section=gs)
        print("rdx168_minimiser", len(xyz))
    finally:
        shutil.rmtree(tmp)


def make_examples():
    """The reference's own example workflows as shipped (examples/1-reaxff and examples/3-reaxpq+: polyethylene, geninit -mc 2 3 5,
    their rxmd.in; example 3 adds PQEq (pqeq1.par) and an electric field along x), shortened to 20 steps: MDstep rows and the last frame"""
    for ex, extra in (("example1", []), ("example3", ["pqeq1.par"])):
        tmp = tempfile.mkdtemp(prefix="golden_")
        try:
            os.makedirs(os.path.join(tmp, "DAT"))
            shutil.copy(os.path.join(INP, "example1", "pe_cell.xyz"), os.path.join(tmp, "input.xyz"))
            shutil.copy(os.path.join(INP, "example1", "ffield_pe"), os.path.join(tmp, "ffield"))
            shutil.copy(os.path.join(INP, ex, "rxmd_%s.in" % ex), os.path.join(tmp, "rxmd.in"))
            for f in extra:                                   # the rxmd.in of example 3 names ./pqeq1.par
                shutil.copy(os.path.join(INP, ex, f.replace(".par", "_%s.par" % ex)), os.path.join(tmp, f))
            run([os.path.join(REFBIN, "geninit"), "-i", "input.xyz", "-f", "ffield", "-o", "DAT", "-mc", "2", "3", "5"], tmp)
            out = run([os.path.join(REFBIN, "rxmd"), "--ntime_step", "20", "--pstep", "5", "--fstep", "20"], tmp)
            md = parse_mdstep(out)
            xyz = open(os.path.join(tmp, "DAT", "%09d.xyz" % 20)).read()
            np.savez_compressed(os.path.join(HERE, ex + ".npz"), mdstep=md, xyz_last=np.array(xyz))
            print(ex, "rows", len(md), "PE/atom", md[0][2])
        finally:
            shutil.rmtree(tmp)


if __name__ == "__main__":
    if sys.argv[1:] == ["examples"]:
        make_examples(); sys.exit(0)
    if sys.argv[1:] == ["minimiser"]:
        make_minimiser(); sys.exit(0)
    if sys.argv[1:] == ["thermo"]:
        make_thermo(); sys.exit(0)
    if sys.argv[1:] == ["thermo_mpi"]:
        make_thermo_mpi(); sys.exit(0)
    for n in (sys.argv[1:] or list(CASES)):
        make(n)
