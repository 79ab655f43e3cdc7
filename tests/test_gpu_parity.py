"""Parity of the HIP path (through the C ABI, rxmd_amd.RxmdEngine -> librxmd_hip.so) against the
plain-C oracle on the same inputs, and against the golden vectors of the real reference.

Tolerances (BASELINE.json north_star / SURVEY 8d):
  charges  max_i |q_i - q_ref| / max(|q_ref|, q_rms)            <= 1e-6
  forces   max_i ||f_i - f_ref||_inf / max(||f_ref||_inf, f_rms) <= 1e-6
  energies 14 terms                                              <= 1e-9 relative
QEq parity is gated at tight tolerance (QEq_tol 1e-12): at the benchmark tolerance 1e-7 the reference's
CG exits by chance on REAL(4) step-length noise and is not reproducible to 1e-6 even against itself
under atom re-ordering; that case is reported, with looser documented bounds.
"""
import os
import numpy as np
import pytest

import oracle_api as oa

pytestmark = pytest.mark.gpu

QTOL = 1e-6
FTOL = 1e-6
ETOL = 1e-9


def _engine(case, mc, **kw):
    import rxmd_amd
    from rxmd_amd import system
    ff, names, frac, lat = oa.make_system(case)
    lat3, rec = system.geninit(ff, names, frac, lat, mc=mc, lg=kw.get("lg", False))
    e = rxmd_amd.RxmdEngine(ff, lat3, **kw)
    e.set_atoms_rxff(rec)
    return e


def _oracle(case, mc, **kw):
    ff, names, frac, lat = oa.make_system(case)
    lat2, ranks = oa.geninit(names, frac, lat, oa.ffield_names(ff, lg=kw.get("lg", False)), mc=mc)
    return oa.Oracle(ff, lat2, ranks, **kw)


def q_err(q, qref):
    qrms = np.sqrt((qref ** 2).mean())
    return (np.abs(q - qref) / np.maximum(np.abs(qref), max(qrms, 1e-300))).max()


def f_err(f, fref):
    frms = np.sqrt((fref ** 2).mean())
    return (np.abs(f - fref).max(axis=1) / np.maximum(np.abs(fref).max(axis=1), frms)).max()


def e_err(pe, pref):
    return max(abs(a - b) / abs(b) for a, b in zip(pe, pref) if abs(b) > 1e-6)


def test_library_is_the_hip_build():
    import rxmd_amd
    L = rxmd_amd.load_library()
    assert L.rxmd_hip_has_device_code() == 1
    assert os.path.basename(rxmd_amd.SO_PATH) == "librxmd_hip.so"


@pytest.mark.parametrize("case,mc", [("rdx168", (1, 1, 1)), ("rdx222", (2, 2, 2)), ("ice644", (6, 4, 4))])
def test_tight_tolerance_parity_vs_oracle(case, mc):
    kw = dict(QEq_tol=1e-12, NMAXQEq=2000)
    o = _oracle(case, mc, **kw); o.qeq(); o.force()
    e = _engine(case, mc, **kw)
    it, est = e.QEq()
    pe = e.FORCE()
    a = e.atoms()
    assert (a["gid"] == o.gids()).all()
    assert q_err(a["q"], o.charges()) <= QTOL
    assert f_err(a["f"], o.forces()) <= FTOL
    assert e_err(pe, o.energy()) <= ETOL
    assert abs(est - o.trace()[-1, 0]) <= 1e-9 * abs(est)
    # structure: same ghost set in the same order, same lists
    st = e.stats()
    assert st["natoms"] + st["nghost_force"] == o.L.rxo_nghost_total(o.w, 0)
    assert (e.debug(4).astype(np.int64) == o.get(106).astype(np.int64)).all()
    assert (e.debug(2).astype(int) == o.get(103).astype(int)).all()
    assert (e.debug(6).astype(int) == o.get(104).astype(int)).all()
    assert np.allclose(e.debug(7), o.get(108), rtol=1e-12)
    check_bond_order_taps(e, o)
    e.close()


def check_bond_order_taps(e, o):
    """a9 / a10 / a18 directly: delta' of BOPRIM (bo.F90:28-118), delta and the corrected bond orders of BOFULL (bo.F90:121-298), cdbnd after
    the energy terms and ccbnd as ForceBondedTerms consumes it in local index order (pot.F90:113-144), residents and ghosts"""
    n = len(o.gids())
    cnt, pg, bo = e.bonds()
    onbr, obo = o.bonds()
    gidG = o.get(106).astype(np.int64)
    compare_bond_order_taps(dict(deltap=e.debug(1), delta=e.debug(0), cd=e.debug(8), cc=e.debug(10), cnt=cnt, pg=pg, bo=bo),
                            dict(deltap=o.get(102), delta=o.get(101), cd=o.get(110), cc=o.get(109), cnt=o.get(103)[:n], nbr=onbr[:n], bo=obo[:n], gidG=gidG,
                                 pos=o.get(100, width=3)), n, np.asarray(e.lattice[:3]))


def compare_bond_order_taps(E, O, n, box, rtol=1e-10):
    """E: the engine's taps, O: the oracle's.  Tolerance 1e-10, not 1e-12: the reference sends ALL positions through normalised coordinates and
    back in every COPYATOMS call (comm.F90:222-227,260-264; two calls per CG iteration), which moves them by ~1e-12 A before FORCE sees
    them; the engine keeps the residents where they are.  1e-12 A x d(BO')/dr of a few per A = the 1e-11 measured on delta'."""
    gidG = O["gidG"]
    for k in ("deltap", "delta", "cd"):
        assert np.abs(E[k] - O[k]).max() <= rtol * max(np.abs(O[k]).max(), 1.0), k
    cnt = E["cnt"]
    assert np.array_equal(cnt, O["cnt"].astype(cnt.dtype))
    for i in range(n):
        c = int(cnt[i])
        ge, go = E["pg"][i, :c], gidG[O["nbr"][i, :c] - 1]
        ie, io = np.lexsort((E["bo"][i, :c], ge)), np.lexsort((O["bo"][i, :c], go))      # an atom can be bonded to two images of one partner in a one-cell box
        assert np.array_equal(ge[ie], go[io])
        assert np.abs(E["bo"][i, :c][ie] - O["bo"][i, :c][io]).max(initial=0.0) <= rtol
    # ccbnd.  The reference evaluates a torsion i-j-k-l once, from the centre atom with the smaller gid, and scatters the coefficient of the
    # far bond k-l to WHICHEVER IMAGES of k and l that centre atom sees (pot.F90:1188-1215 -> ForceB); the engine books the far bond from k's
    # own visit of the torsion, i.e. on the image of k that is a centre atom there.  Forces are linear in ccbnd and
    # translation invariant, so after the CPBK fold both give the same forces, but per INDEX the two differ for atoms whose torsions cross
    # the box.  What must agree: ccbnd summed over the images of an atom, and ccbnd per index for atoms deeper inside the box than any
    # torsion reaches (3 bonds < 12 A) -- none in the boxes of 1-2 cut-offs, most of a 36k-atom box.
    G = len(gidG)
    scale = max(np.abs(O["cc"]).max(), 1.0)
    se = np.bincount(gidG, weights=E["cc"][:G]); so = np.bincount(gidG, weights=O["cc"][:G])
    assert np.abs(se - so).max() <= 1e-9 * scale
    depth = np.minimum(O["pos"][:n], box - O["pos"][:n]).min(axis=1)
    deep = depth > 12.0
    if deep.any():
        assert np.abs(E["cc"][:n][deep] - O["cc"][:n][deep]).max() <= 1e-9 * scale
    return int(deep.sum())


@pytest.mark.parametrize("case,npz", [("rdx168", "rdx168_tight"), ("rdx222", "rdx222_tight"), ("ice644", "ice644_tight")])
def test_tight_tolerance_parity_vs_reference_golden(case, npz):
    g = np.load(os.path.join(oa.GOLD, npz + ".npz"))
    e = _engine(case, tuple(int(x) for x in g["mc"]), QEq_tol=1e-12, NMAXQEq=2000)
    e.QEq(); e.FORCE()
    a = e.atoms()
    assert (a["gid"] == g["gid"]).all()
    assert q_err(a["q"], g["charge"]) <= QTOL
    assert f_err(a["f"], g["force"]) <= FTOL
    e.close()


def test_step0_energies_match_the_references_published_sample_output():
    """README.md:157 of the reference: per-atom energies of the 168-atom RDX sample at step 0 (default rxmd.in, QEq tol 1e-7, q0 = 0),
    to the digits it prints -- through the C ABI, both QEq modes"""
    for mode in (0, 1):
        e = _engine("rdx168", (1, 1, 1), qeq_mode=mode)
        e.QEq(); pe = e.FORCE()
        oa.check_readme_known_answer(np.asarray(pe))
        e.close()


def test_forces_only_no_qeq_vs_reference_golden():
    g = np.load(os.path.join(oa.GOLD, "rdx168_noqeq.npz"))
    e = _engine("rdx168", (1, 1, 1), isQEq=0)
    it, _ = e.QEq()
    assert it == 0
    pe = e.FORCE()
    a = e.atoms()
    assert f_err(a["f"], g["force"]) <= 1e-9
    n = len(a["gid"])
    m = g["mdstep"][0]
    assert abs(pe[0] / n - m[2]) <= 1e-5 * abs(m[2])
    assert pe[12] == 0.0 and pe[13] == 0.0
    e.close()


def test_injected_oracle_charges_isolate_force_kernels():
    """benchmark-tolerance protocol, step 2 of SURVEY 8d: oracle charges at QEq_tol 1e-7 injected, forces compared"""
    o = _oracle("rdx222", (2, 2, 2)); o.qeq(); o.force()
    e = _engine("rdx222", (2, 2, 2), isQEq=0)
    e.set_charges(o.charges())
    pe = e.FORCE()
    # not 1e-12: the reference round-trips ALL positions through normalised coordinates in every COPYATOMS call
    # (comm.F90:222-227,260-264; two calls per CG iteration), which moves them by ~1e-12 A over a QEq call; the
    # engine keeps resident positions fixed.  1e-12 A x bond stiffness ~1e3 kcal/mol/A^2 = 1e-9 in the forces.
    assert f_err(e.atoms()["f"], o.forces()) <= 1e-7
    assert e_err(pe, o.energy()) <= ETOL
    e.close()


# the reference's own numbers at QEq_tol 1e-7 on RDX-168: charges move by up to 1.5e-5 when the atoms are merely
# re-ordered, and lie 4.7e-5 from the converged (1e-12) solution
REF_REORDER_SPREAD = 1.5e-5
REF_TRUNCATION_ERR = 4.7e-5


@pytest.mark.parametrize("qeq_mode", [0, 1])
def test_benchmark_tolerance_report(qeq_mode):
    """SURVEY 8d step 2 for BOTH QEq algebras (qeq_mode 1 is what bench.py times): at QEq_tol 1e-7 iteration counts may differ by a
    few (chance exit on REAL(4) step-length noise) and charges by ~1e-5.  Bounds: against the reference's own tol-1e-7 charges no
    more than its truncation error (two runs that each stop within that error of the fixed point), against the converged charges no
    more than the reference's own truncation error plus its re-ordering spread; then forces with the ORACLE's tol-1e-7 charges
    injected (isolates the force kernels from the CG exit noise)."""
    g = np.load(os.path.join(oa.GOLD, "rdx168_tol7.npz"))
    gt = np.load(os.path.join(oa.GOLD, "rdx168_tight.npz"))
    e = _engine("rdx168", (1, 1, 1), qeq_mode=qeq_mode)
    it, est = e.QEq()
    a = e.atoms()
    d7 = np.abs(a["q"] - g["charge"]).max(); dt = np.abs(a["q"] - gt["charge"]).max()
    print("qeq_mode %d at tol 1e-7: %d iterations (reference %d), max|dq| vs reference tol-1e-7 %.2e, vs converged %.2e" % (qeq_mode, it, int(g["qeq_iters"][0]), d7, dt))
    assert abs(it - int(g["qeq_iters"][0])) <= 0.4 * int(g["qeq_iters"][0])     # chance exit on REAL(4) step-length noise, see the module docstring
    assert d7 <= REF_TRUNCATION_ERR
    assert dt <= REF_TRUNCATION_ERR + REF_REORDER_SPREAD
    assert abs(est - g["qeq_trace_last"][-1, 3]) <= 1e-5 * abs(est)
    o = _oracle("rdx168", (1, 1, 1)); o.qeq(); o.force()
    assert np.abs(a["q"] - o.charges()).max() <= REF_TRUNCATION_ERR
    e.set_charges(o.charges())
    pe = e.FORCE()
    assert f_err(e.atoms()["f"], o.forces()) <= 1e-7             # see test_injected_oracle_charges_isolate_force_kernels for the 1e-7
    assert e_err(pe, o.energy()) <= ETOL
    e.close()


def _one_exit_runs(its):
    """lengths of the runs of consecutive QEq calls that left after at most one iteration"""
    out, r = [], 0
    for x in its:
        if x <= 1:
            r += 1
        else:
            if r:
                out.append(r)
            r = 0
    if r:
        out.append(r)
    return out


@pytest.mark.parametrize("qeq_mode,win", [(1, "1"), (0, "1"), (1, "0"), (0, "0")])
def test_iteration_statistics_over_1000_steps(qeq_mode, win, monkeypatch):
    """What a change of the CG's rounding is judged by (round 6).  The iteration count of a SINGLE QEq call at QEq_tol 1e-7 is decided by REAL(4) noise
    of the step length (the reference itself: 35 -> 31..39 under atom re-ordering, and one call in ten leaves after ONE iteration -- the
    relative change of Est over a steepest-descent step from nearly converged charges is below 1e-7 -- sometimes several steps in a row, the charges
    drifting ~4e-5 from the converged solution per such step), so no test can pin it; its STATISTICS over many calls above one cell can be pinned.
    RDX 2 x 2 x 2, 1000 MD steps at the bench settings, both QEq algebras, window pass and row pass, against the reference's own numbers over the same
    1000 steps (tests/golden/rdx222_md1000.npz: `rxmd` itself, iteration counts; rdx222_md1000_dq.npz: the bit-path oracle + the distance of its
    charges from the converged solution of each geometry):
      reference: mean 30.82 iterations per step over steps 11..1000 (one cell: 24; steps 6..25: 36 -- neither is the bar), 9.0 % one-iteration exits;
      oracle:    30.50, 9.7 %, longest run 6, |dq| against converged: median 5.6e-6, 90th percentile 7.9e-5, 99th 4.1e-4, worst 6.7e-4.
    Gates: mean within 10 %; one-iteration exits within a factor 1.6 either way; 90th percentile <= 1.5 x, 99th <= 2 x, worst step <= 5 x the reference's
    (tails of 1000 samples; the four variants measured 0.6-0.9 x, 0.7-1.1 x, 0.8-1.6 x: scripts/gpu_iterstat.py, profiles/r06_iterstat1000.log)."""
    import rxmd_amd
    monkeypatch.setenv("RXMD_SPMV_WIN", win)
    g = np.load(os.path.join(oa.GOLD, "rdx222_md1000.npz")); gd = np.load(os.path.join(oa.GOLD, "rdx222_md1000_dq.npz"))
    ref_its = g["qeq_iters"]; ref_dq = gd["dq_tight"]
    # the oracle reproduces the reference's count call by call for the first 369 calls (and for every shorter golden); there one call takes 38 instead of 36
    # iterations -- a last-bit difference somewhere flipped one REAL(4) rounding -- and from then on the two are different draws of the SAME statistics:
    # mean 30.50 / 30.82 iterations, 9.7 % / 9.0 % one-iteration exits.  Iteration statistics are held against the reference's own run, the distance from
    # the converged charges against the oracle's.
    assert np.array_equal(ref_its[:369], gd["qeq_iters"][:369]) and abs(gd["qeq_iters"][11:].mean() - ref_its[11:].mean()) < 0.02 * ref_its[11:].mean()
    ref_mean = ref_its[11:].mean(); ref_ones = (ref_its[11:] <= 1).mean()
    e = _engine("rdx222", (2, 2, 2), qeq_mode=qeq_mode)
    t = rxmd_amd.RxmdEngine(oa.make_system("rdx222")[0], e.lattice, QEq_tol=1e-12, NMAXQEq=2000, qeq_mode=0)
    it0, _ = e.QEq(); e.FORCE()
    its, dqs = [it0], []
    for s in range(1000):
        e.step(1)
        its.append(e.stats()["qeq_iters_last"])
        rec = e.get_atoms_rxff()
        t.set_atoms_rxff(rec); t.QEq()                                 # the converged charges of this geometry, started from the engine's
        dqs.append(np.abs(t.atoms()["q"] - rec[:, 6]).max())
    its = np.array(its); dqs = np.array(dqs)
    mean = its[11:].mean(); ones = (its[11:] <= 1).mean(); rl = _one_exit_runs(its[11:])
    print("qeq_mode %d win %s: mean iterations/step %.2f (reference %.2f), one-iteration exits %.3f (%.3f), longest run %d (%d); |dq| vs converged p50 %.2e p90 %.2e p99 %.2e max %.2e (reference %.2e %.2e %.2e %.2e)"
          % (qeq_mode, win, mean, ref_mean, ones, ref_ones, max(rl), max(_one_exit_runs(ref_its[11:])), np.percentile(dqs, 50), np.percentile(dqs, 90), np.percentile(dqs, 99), dqs.max(),
             np.percentile(ref_dq, 50), np.percentile(ref_dq, 90), np.percentile(ref_dq, 99), ref_dq.max()))
    assert abs(mean - ref_mean) <= 0.10 * ref_mean, (mean, ref_mean)
    assert ref_ones / 1.6 <= ones <= 1.6 * ref_ones, (ones, ref_ones)
    assert np.percentile(dqs, 90) <= 1.5 * np.percentile(ref_dq, 90)
    assert np.percentile(dqs, 99) <= 2.0 * np.percentile(ref_dq, 99)
    assert dqs.max() <= 5.0 * ref_dq.max()       # (the single worst of 1,000 calls is the noisiest of these numbers -- an early chance exit in the middle of a solve; the hydrogen-bond atomics make the
                                                 #  trajectory, hence the draw, differ from run to run: measured 0.8-1.6 x over eleven runs, the gate leaves room for the tail)
    e.close(); t.close()


def _rec10_from_oracle(o, lat):
    """rxff.bin records of the oracle's present state (orthorhombic box, one rank)"""
    n = len(o.gids())
    rec = np.zeros((n, 10))
    rec[:, 0:3] = o.pos() / np.asarray(lat[:3]); rec[:, 3:6] = o.vel(); rec[:, 6] = o.charges()
    rec[:, 7] = o.types() + o.gids() * 1e-13
    return rec


@pytest.mark.parametrize("qeq_mode", [0, 1])
def test_md_trajectory_at_benchmark_tolerance(qeq_mode):
    """RDX 2x2x2, 10 MD steps at the bench settings (QEq_tol 1e-7, dt 0.25 fs) against the oracle: the trajectory may differ by what
    the CG exit noise of every step injects (charges within the reference's own truncation error each step); with the oracle's final
    state handed to a second engine (positions, its tol-1e-7 charges, no QEq) the force kernels must agree to the parity tolerance."""
    import rxmd_amd
    o = _oracle("rdx222", (2, 2, 2)); o.qeq(); o.force(); o.step(10)
    e = _engine("rdx222", (2, 2, 2), qeq_mode=qeq_mode); e.QEq(); e.FORCE(); e.step(10)
    a = e.atoms()
    assert (a["gid"] == o.gids()).all()
    dq = np.abs(a["q"] - o.charges()).max()
    dx = np.abs(a["pos"] - o.pos()).max()
    print("qeq_mode %d, 10 steps at tol 1e-7: max|dq| %.2e, max|dx| %.2e A, iterations/step %.1f" % (qeq_mode, dq, dx, e.stats()["qeq_iters_total"] / e.stats()["qeq_calls"]))
    assert dq <= REF_TRUNCATION_ERR + REF_REORDER_SPREAD
    assert dx <= 1e-6                                             # 10 steps of 0.25 fs under force differences of ~1e-4 kcal/mol/A
    assert f_err(a["f"], o.forces()) <= 5e-3                      # forces follow the charges: 1.6e-5 e x ~1e2 kcal/mol/A/e on an rms force of 1.2 (measured 1.5e-3 .. 2.2e-3)
    ke, ko = e.energy()["KE"], o.kinetic()
    assert abs(ke - ko) <= 1e-4 * abs(ko)                         # the crystal starts at rest: KE is the small quantity the force noise moves (measured 2.4e-5)
    e.close()
    ff, names, frac, lat = oa.make_system("rdx222")
    lat2 = [lat[0] * 2, lat[1] * 2, lat[2] * 2] + list(lat[3:6])
    e2 = rxmd_amd.RxmdEngine(ff, lat2, isQEq=0)
    e2.set_atoms_rxff(_rec10_from_oracle(o, lat2))
    pe = e2.FORCE()
    assert (e2.atoms()["gid"] == o.gids()).all()
    assert f_err(e2.atoms()["f"], o.forces()) <= 1e-7
    assert e_err(pe, o.energy()) <= ETOL
    e2.close()


def test_md_trajectory_tight():
    kw = dict(QEq_tol=1e-12, NMAXQEq=2000)
    o = _oracle("rdx168", (1, 1, 1), **kw); o.qeq(); o.force(); o.step(5)
    e = _engine("rdx168", (1, 1, 1), **kw); e.QEq(); e.FORCE(); e.step(5)
    a = e.atoms()
    assert (a["gid"] == o.gids()).all()                         # same local order after migration
    assert np.abs(a["pos"] - o.pos()).max() <= 1e-9
    assert np.abs(a["v"] - o.vel()).max() <= 1e-9
    assert q_err(a["q"], o.charges()) <= QTOL
    assert f_err(a["f"], o.forces()) <= FTOL
    en = e.energy()
    assert abs(en["KE"] - o.kinetic()) <= 1e-7 * abs(o.kinetic())
    e.close()


def test_migration_across_periodic_boundary_keeps_reference_order():
    """give atoms a velocity that carries some of them through the box faces; order after MODE_MOVE must match"""
    kw = dict(QEq_tol=1e-12, NMAXQEq=2000)
    rng = np.random.default_rng(7)
    v = rng.normal(0, 0.08, (168, 3))
    ff, names, frac, lat = oa.make_system("rdx168")
    lat2, ranks = oa.geninit(names, frac, lat, oa.ffield_names(ff))
    o = oa.Oracle(ff, lat2, ranks, v0=[v], **kw); o.qeq(); o.force(); o.step(6)
    e = _engine("rdx168", (1, 1, 1), **kw); e.set_velocities(v); e.QEq(); e.FORCE(); e.step(6)
    a = e.atoms()
    og = o.gids()
    assert not (og == np.arange(1, 169)).all(), "test needs at least one migration"
    assert (a["gid"] == og).all()
    assert np.abs(a["pos"] - o.pos()).max() <= 1e-8
    assert f_err(a["f"], o.forces()) <= 1e-5          # hot, far-from-equilibrium trajectory: looser
    e.close()


@pytest.mark.parametrize("case,mc", [("rdx168", (1, 1, 1)), ("rdx168", (5, 5, 5)), ("ice644", (6, 4, 4)), ("sicnp", (1, 1, 1))])
def test_fused_self_exchange_gives_the_staged_order_bit_for_bit(case, mc, monkeypatch):
    """Single rank, round 5: the ghost build and the migration as 26 image segments (three / four kernels, one host wait; engine.hip) against the
    six-stage flag -> scan -> append form they replace (RXMD_NO_STAGE_PAIRS=1 keeps it reachable), which is the reference's own procedure
    (comm.F90:55-100,238-257).  Hot atoms so that some cross the faces in every step; compared after 4 steps: the WHOLE local arrays, residents and
    ghosts in local order -- gid, type exactly; positions, charges, velocities, forces bit for bit on ice, to 1e-11 where hydrogen-bond atomics run.  13 A boxes (every atom has up to 26 images, both faces of an axis at once) and a 65 A box (interior atoms)."""
    kw = dict(QEq_tol=1e-12, NMAXQEq=2000)
    if case == "sicnp":
        kw["pqeq"] = oa.PQEQ_SICNP
    res = []
    for staged in (False, True):
        if staged:
            monkeypatch.setenv("RXMD_NO_STAGE_PAIRS", "1")
        else:
            monkeypatch.delenv("RXMD_NO_STAGE_PAIRS", raising=False)
        e = _engine(case, mc, **kw)
        n = e.stats()["natoms"]
        rng = np.random.default_rng(11)
        e.set_velocities(rng.normal(0, 0.08, (n, 3)))
        e.QEq(); e.FORCE(); e.step(4)
        a = e.atoms()
        res.append(dict(gid_all=e.debug(4).copy(), type_all=e.debug(5).copy(), pos_all=e.debug(3, width=3).copy(), q_all=e.debug(9).copy(),
                        gid=a["gid"].copy(), pos=a["pos"].copy(), q=a["q"].copy(), f=a["f"].copy(), v=a["v"].copy(), st=e.stats()))
        e.close()
    f, s = res
    assert not (f["gid"] == np.arange(1, len(f["gid"]) + 1)).all(), "test needs at least one migration"
    assert f["st"]["nghost_force"] == s["st"]["nghost_force"] > 0
    for k in ("gid_all", "type_all", "gid"):                          # the ORDER: residents after migration, every ghost of every stage
        assert np.array_equal(f[k], s[k]), k
    # Values: bit for bit where no hydrogen-bond acceptor atomics run (ice: Ehb == 0, pot.F90:595); with them (RDX, SiC + O2) the forces of
    # two RUNS of either path differ in the last bits (test_forces_and_charges_are_bitwise_reproducible_run_to_run), and so does what is integrated from them
    for k in ("pos_all", "q_all", "pos", "q", "v", "f"):
        if case == "ice644":
            assert np.array_equal(f[k], s[k]), k
        else:
            assert np.abs(f[k] - s[k]).max() <= 1e-11 * max(np.abs(s[k]).max(), 1.0), k


def test_reference_shaped_entry_points():
    """rxmd_hip_QEq / rxmd_hip_FORCE take atype(NBUFFER), pos(NBUFFER,3) column-major like the Fortran subroutines"""
    kw = dict(QEq_tol=1e-12, NMAXQEq=2000)
    o = _oracle("rdx168", (1, 1, 1), **kw); o.qeq(); o.force()
    import rxmd_amd
    ff, names, frac, lat = oa.make_system("rdx168")
    e = rxmd_amd.RxmdEngine(ff, lat, **kw)
    nbuf = 400
    atype = np.zeros(nbuf); pos = np.zeros((nbuf, 3), order="F"); q = np.zeros(nbuf)
    atype[:168] = o.types() + o.gids() * 1e-13
    pos[:168] = o.pos()
    e.QEq_arrays(atype, pos, q, 168)
    assert q_err(q[:168], o.charges()) <= QTOL
    f, pe = e.FORCE_arrays(atype, pos, q, 168)
    assert f_err(np.ascontiguousarray(f[:168]), o.forces()) <= FTOL
    e.close()


def test_error_codes_mirror_reference_traps():
    import rxmd_amd
    ff, names, frac, lat = oa.make_system("rdx168")
    from rxmd_amd import system
    lat3, rec = system.geninit(ff, names, frac, lat)
    e = rxmd_amd.RxmdEngine(ff, lat3, maxneighbs10=64)           # MAXNEIGHBS10 too small -> qeq.F90:248-252
    e.set_atoms_rxff(rec)
    with pytest.raises(rxmd_amd.RxmdError) as ei:
        e.QEq()
    assert ei.value.code == -5
    e.close()
    e = rxmd_amd.RxmdEngine(ff, lat3, nbuffer=1000)              # NBUFFER too small -> comm.F90:467-472
    e.set_atoms_rxff(rec)
    with pytest.raises(rxmd_amd.RxmdError) as ei:
        e.FORCE()
    assert ei.value.code == -3
    e.close()
    e = rxmd_amd.RxmdEngine(ff, lat3, maxneighbs=4)              # MAXNEIGHBS too small -> main.F90:402-407
    e.set_atoms_rxff(rec)
    with pytest.raises(rxmd_amd.RxmdError) as ei:
        e.FORCE()
    assert ei.value.code == -4
    e.close()


@pytest.mark.parametrize("qeq_mode", [0, 1])
def test_full_size_properties_rdx_1m(qeq_mode):
    """BASELINE configs[1] size (979,776 atoms): size-independent properties instead of an oracle run:
    a replicated crystal must reproduce the unit cell -- per-atom charges/forces equal those of the 168-atom cell
    (same images by periodicity), energies scale with the cell count, net charge and net force vanish."""
    kw = dict(QEq_tol=1e-12, NMAXQEq=2000)
    e1 = _engine("rdx168", (1, 1, 1), **kw); e1.QEq(); pe1 = e1.FORCE(); a1 = e1.atoms(); e1.close()
    mc = (18, 18, 18)
    e = _engine("rdx168", mc, qeq_mode=qeq_mode, **kw)
    it, est = e.QEq(); pe = e.FORCE(); a = e.atoms()
    ncell = mc[0] * mc[1] * mc[2]
    assert len(a["q"]) == 168 * ncell
    q = a["q"].reshape(ncell, 168); f = a["f"].reshape(ncell, 168, 3)
    # two CG runs that stop on a 1e-12 relative energy change agree to ~1e-7 in the charges (the exit iteration differs with
    # the summation order, see the module docstring); the parity tolerance is 1e-6 relative
    assert q_err(q.reshape(-1), np.tile(a1["q"], ncell)) <= 5e-7
    # the index-ordered ccbnd rule makes bonded forces depend on which neighbours are ghosts: compare only cells
    # whose ghost pattern equals the unit cell's (none do exactly), so use invariants instead of per-atom equality
    assert abs(a["q"].sum()) <= 1e-6
    for k in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13):
        assert abs(pe[k] - ncell * pe1[k]) <= 1e-7 * abs(ncell * pe1[k]) + 1e-6
    st = e.stats()
    assert st["max_n10"] == 447 and st["max_nb"] == 12          # RDX crystal statistics
    # the matrix passes were window passes, and the one-time search for a placement of their streams ran and kept the fastest it saw
    assert st["win_in_use"] == 1 and (st["natoms"] + 15) // 16 <= st["win_groups"] <= st["natoms"] // 16 + st["cells10"][0] * st["cells10"][1] + 1
    assert 0.0 < st["place_ms_kept"] <= st["place_ms_first"] < 5.0, (st["place_ms_first"], st["place_ms_kept"])
    e.close()


@pytest.mark.parametrize("case,mc_small,mc_full,natoms", [("ice644", (6, 4, 4), (60, 35, 40), 2016000), ("sicnp", (1, 1, 1), (12, 12, 12), 945216)])
def test_full_size_properties_water_and_sicnp(case, mc_small, mc_full, natoms):
    """BASELINE configs[2] (perturbed ice Ih, 2,016,000 atoms) and configs[4] (SiC nanoparticle + O2 with PQEq, 945,216 atoms) at full
    size, by the same size-independent property as the RDX case: both are replications of a unit cell, so per-atom charges repeat
    those of a small replication of the same cell, every energy term scales with the cell count and the net charge vanishes.
    Water additionally keeps the reference's quirk that no hydrogen bond is ever found (type 2 is O in this ffield)."""
    kw = dict(QEq_tol=1e-12, NMAXQEq=2000)
    if case == "sicnp":
        kw["pqeq"] = oa.PQEQ_SICNP
    e1 = _engine(case, mc_small, **kw); e1.QEq(); pe1 = e1.FORCE(); a1 = e1.atoms(); e1.close()
    e = _engine(case, mc_full, qeq_mode=1, **kw)
    e.QEq(); pe = e.FORCE(); a = e.atoms()
    assert len(a["q"]) == natoms
    n0 = len(a1["q"]) // (mc_small[0] * mc_small[1] * mc_small[2])          # atoms of the unit cell
    ratio = natoms // len(a1["q"])
    # charges: atom t of every unit cell carries the charge of atom t of the small system's first cell (gid order: cell-major)
    o1 = np.argsort(a1["gid"]); o = np.argsort(a["gid"])
    q_cell = a1["q"][o1][:n0]
    assert q_err(a["q"][o], np.tile(q_cell, natoms // n0)) <= 5e-7
    assert abs(a["q"].sum()) <= 1e-6 * np.abs(a["q"]).sum()
    scale = abs(ratio * pe1[0])
    for k in range(1, 14):
        assert abs(pe[k] - ratio * pe1[k]) <= 1e-7 * abs(ratio * pe1[k]) + 1e-9 * scale      # second term: PQEq's PE(12) is a small difference of large sums
    if case == "ice644":
        assert pe[10] == 0.0
    e.close()


@pytest.mark.parametrize("case,mc", [("rdx168", (1, 1, 1)), ("rdx222", (2, 2, 2))])
def test_one_pass_qeq_mode_reaches_the_same_fixed_point(case, mc):
    """qeq_mode=1 (one matrix pass per CG iteration, gradient/Est by recurrence on stored row sums) is an opt-in
    re-association of the reference algebra: at tight tolerance it must land on the reference's charges and forces."""
    kw = dict(QEq_tol=1e-12, NMAXQEq=2000)
    o = _oracle(case, mc, **kw); o.qeq(); o.force()
    e = _engine(case, mc, qeq_mode=1, **kw)
    it, est = e.QEq(); pe = e.FORCE(); a = e.atoms()
    assert q_err(a["q"], o.charges()) <= QTOL
    assert f_err(a["f"], o.forces()) <= FTOL
    assert e_err(pe, o.energy()) <= ETOL
    assert abs(est - o.trace()[-1, 0]) <= 1e-9 * abs(est)
    assert e.stats()["spmv_launches"] == it          # one pass per iteration
    e.close()


@pytest.mark.parametrize("qeq_mode", [0, 1])
def test_matrix_pass_without_the_software_pipeline_is_the_same_operator(qeq_mode, monkeypatch):
    """The QEq matrix pass requests the first batch of a row before the row length is known and every further batch ahead of the
    gathers of the one before (default); RXMD_SPMV_NO_PIPE=1 is the plain load-then-use loop.  Same rows, same summation order per
    lane: the tight-tolerance fixed point and the iteration count must not move."""
    kw = dict(QEq_tol=1e-12, NMAXQEq=2000)
    o = _oracle("rdx222", (2, 2, 2), **kw); o.qeq(); o.force()
    res = []
    monkeypatch.setenv("RXMD_SPMV_WIN", "0")          # the row pass (k_spmv), not the window pass that is the default
    for off in (False, True):
        if off:
            monkeypatch.setenv("RXMD_SPMV_NO_PIPE", "1")
        e = _engine("rdx222", (2, 2, 2), qeq_mode=qeq_mode, **kw)
        it, est = e.QEq(); pe = e.FORCE(); a = e.atoms()
        assert (e.debug(6).astype(int) == o.get(104).astype(int)).all()
        assert np.allclose(e.debug(7), o.get(108), rtol=1e-12)
        assert q_err(a["q"], o.charges()) <= QTOL
        assert f_err(a["f"], o.forces()) <= FTOL
        assert abs(est - o.trace()[-1, 0]) <= 1e-9 * abs(est)
        res.append((it, a["q"].copy()))
        e.close()
    assert res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1])


@pytest.mark.parametrize("case,mc,qeq_mode", [("rdx222", (2, 2, 2), 0), ("rdx222", (2, 2, 2), 1), ("rdx168", (5, 5, 5), 1), ("ice644", (6, 4, 4), 1), ("sicnp", (1, 1, 1), 0), ("sicnp", (1, 1, 1), 1),
                                              ("example1", (2, 3, 5), 1), ("example1", (2, 3, 5), 0)])     # polyethylene: rows LONGER than two batches of the window pass (see below)
def test_window_pass_and_row_pass_are_the_same_operator(case, mc, qeq_mode, monkeypatch):
    """The default QEq matrix pass (k_spmv_win) holds the partners of a group of 16 cell-sorted rows in LDS and reads them through a 16-bit
    slot per entry; RXMD_SPMV_WIN=0 is the wavefront-per-row pass with its 16-byte gather per entry (also the fallback when a window does
    not fit).  Same matrix, same vectors -- only the order in which a row's products are added differs: at the tight tolerance both reach
    the oracle's fixed point, and the stats say which pass ran."""
    kw = dict(QEq_tol=1e-12, NMAXQEq=2000)
    if case == "sicnp":
        kw["pqeq"] = oa.PQEQ_SICNP                  # PQEq: a third stream (shell-core matrix) over the same slots
    o = _oracle(case, mc, **kw)
    if "pqeq" in kw:
        o.set_pqeq_clean(1)
    o.qeq(); o.force()
    res = {}
    for win in ("1", "0"):
        monkeypatch.setenv("RXMD_SPMV_WIN", win)
        e = _engine(case, mc, qeq_mode=qeq_mode, **kw)
        it, est = e.QEq(); pe = e.FORCE(); a = e.atoms(); st = e.stats()
        assert st["win_in_use"] == int(win), st
        if case.startswith("example"):
            # rows of more than 512 entries: the window pass keeps two batches of 256 entries in flight and takes a THIRD trip through its loop here (the
            # path that carried 48 bytes of scratch per lane until round 6; crystalline polyethylene has 0.13 atoms / A^3: ~545 entries at 10 A.  At PQEq's 12.5 A its ~1,050-entry rows need more window slots
            # than a workgroup's LDS holds and the engine falls back to the row pass: test_reference_example3_small_box_pqeq_*)
            assert st["max_n10"] > 512, st["max_n10"]
        assert (st["natoms"] + 15) // 16 <= st["win_groups"] <= st["natoms"] // 16 + st["cells10"][0] * st["cells10"][1] + 1 and 0 < st["win_max_units"] <= 448
        assert q_err(a["q"], o.charges()) <= QTOL
        assert f_err(a["f"], o.forces()) <= FTOL
        assert abs(est - o.trace()[-1, 0]) <= 1e-9 * abs(est)
        res[win] = (a["q"].copy(), a["f"].copy(), pe, e.debug(13, cap=4096).copy(), it)
        e.close()
    # (the exit test of qeq.F90:114-115 stops both a few 1e-9 short of the fixed point, each on its own side of it)
    # (forces: 5e-7 since round 6 -- with the CG vectors of the window pass's loop in row order the two runs of RDX 5 x 5 x 5 leave their loops an iteration apart)
    assert q_err(res["1"][0], res["0"][0]) <= 1e-7 and f_err(res["1"][1], res["0"][1]) <= 5e-7
    assert e_err(res["1"][2], res["0"][2]) <= 1e-8
    # the same operator ITERATION BY ITERATION: Est after every CG iteration (the reference's QEQDUMP trace, qeq.F90:117) of the two passes, up
    # to the first of the two exits -- a pass that applied another matrix would part from the other by far more than the rounding of a row sum
    # long before either stops; where they stop differs only because |Est / Est_prev - 1| < 1e-12 is decided by that rounding
    tw, tr = res["1"][3], res["0"][3]
    assert len(tw) == res["1"][4] + 1 and len(tr) == res["0"][4] + 1
    m = min(len(tw), len(tr))
    # (how close: the step length of every iteration is rounded to REAL(4), qeq.F90:23,133 -- a last-bit difference of a row sum can move that
    #  rounding by one float ulp, 6e-8 of the step, and the two CG paths then run 1e-7 .. 1e-5 apart in Est while both are still far from the
    #  fixed point they share; measured 8.3e-6 of max |Est| on RDX 2 x 2 x 2.  A wrong or missing matrix entry shows at the first iteration.)
    # (polyethylene: nearly uniform charges, the CG is through after 7-8 iterations)
    assert m >= (5 if case.startswith("example") else 10) and np.abs(tw[:m] - tr[:m]).max() <= 5e-5 * np.abs(tr[:m]).max(), (len(tw), len(tr), np.abs(tw[:m] - tr[:m]).max())
    assert np.abs(tw[:3] - tr[:3]).max() <= 1e-10 * np.abs(tr[:3]).max()          # start vector and first two iterations: rounding of the row sums only
    to = o.trace()[:, 0]                             # ... and against the oracle's own trace (sequential row sums, the reference's bit path)
    mo = min(m, len(to))
    d0, d1 = np.abs(tw[:mo] - to[:mo]).max(), np.abs(tw[1:mo] - to[:mo - 1]).max()    # (whichever way the oracle's trace counts the start vector)
    assert min(d0, d1) <= 2e-4 * np.abs(to).max(), (d0, d1)       # (measured 5.5e-5 on RDX 2 x 2 x 2, qeq_mode 0)


@pytest.mark.parametrize("case,mc", [("rdx168", (1, 1, 1)), ("rdx222", (2, 2, 2)), ("rdx168", (6, 6, 6)), ("ice644", (6, 4, 4)), ("sicnp547", (1, 1, 1)), ("pbt2272", (1, 1, 1)), ("mos2_tri324", (3, 3, 2))])
def test_window_slots_lead_back_to_the_list_entries(case, mc):
    """The window form of the 10 A matrix, structurally: every list entry's 16-bit slot, looked up in the window of its row's group (win_k / win_cnt,
    rows_sorted), is the entry's own cell-sorted position with the same ghost flag -- for boxes of one cut-off (every partner a periodic image, groups
    that straddle cell columns), a 36k-atom box, water, a nanoparticle in vacuum, an amorphous polymer and a hexagonal cell."""
    e = _engine(case, mc)
    e.QEq()
    n10 = e.debug(6).astype(int); ok = e.debug(11).astype(int)
    st = e.stats()
    assert st["win_in_use"] == 1 and (st["natoms"] + 15) // 16 <= st["win_groups"] <= st["natoms"] // 16 + st["cells10"][0] * st["cells10"][1] + 1
    assert n10.sum() == st["nnz10"] and (ok == n10).all(), (int((ok != n10).sum()), ok[:8], n10[:8])
    e.close()


@pytest.mark.parametrize("once", ["1", "0"])
@pytest.mark.parametrize("case,mc", [("rdx222", (2, 2, 2)), ("ice644", (6, 4, 4)), ("example1", (2, 3, 5))])
def test_torsion_kernel_instances_give_the_same_forces(case, mc, once, monkeypatch):
    """k_e4b has four instances: eight centre atoms per wavefront with their bond slots laid end to end (default when no bond list of
    the step is longer than 15, as in RDX), four atoms laid end to end (any list; RXMD_E4B_SLOTS=4), four atoms with 16 slots each
    (RXMD_E4B_SLOTS=16, lists <= 15) and two with 32 (any list, the default otherwise; RXMD_E4B_SLOTS=32).  Every (atom, slot) accumulator receives the same additions in the same order in all of them, so the forces must
    be bit-identical; the torsion energies are summed per lane and may differ in the last bits.  That holds for the two-visit form
    (RXMD_E4B_ONCE=0).  In the one-visit form (default) the forces on i and on k of the torsions of a batch meet in the same slot accumulators,
    all i first, and where a batch ends depends on the instance: the same additions in another order, 1e-11 apart at most."""
    kw = dict(QEq_tol=1e-12, NMAXQEq=2000)
    monkeypatch.setenv("RXMD_E4B_ONCE", once)
    o = _oracle(case, mc, **kw); o.qeq(); o.force()
    res = []
    for slots in (None, "4", "16", "32"):
        if slots:
            monkeypatch.setenv("RXMD_E4B_SLOTS", slots)
        e = _engine(case, mc, **kw)
        e.QEq(); pe = e.FORCE(); a = e.atoms()
        assert f_err(a["f"], o.forces()) <= FTOL
        assert e_err(pe, o.energy()) <= ETOL
        res.append((a["f"].copy(), pe.copy()))
        e.close()
    for r in res[1:]:
        # (bit for bit where no hydrogen bond adds its acceptor atomics -- the one sum of FORCE whose order is not fixed: one or two components in thousands
        #  then differ by 1e-16 from engine to engine whatever the instance, scripts/gpu_bits_check.py)
        if once == "0" and res[0][1][10] == 0.0: assert np.array_equal(res[0][0], r[0])
        else: assert np.abs(res[0][0] - r[0]).max() <= (1e-13 if once == "0" else 1e-11)
        assert np.allclose(res[0][1], r[1], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("case,mc,kw", [("rdx168", (1, 1, 1), {}), ("rdx222", (2, 2, 2), {}), ("rdx168", (5, 5, 5), {}), ("ice644", (6, 4, 4), {}), ("example1", (2, 3, 5), {}),
                                        ("sicnp", (1, 1, 1), dict(pqeq=oa.PQEQ_SICNP))])
def test_torsions_visited_once_and_twice_and_the_persistent_grid(case, mc, kw, monkeypatch):
    """Three switches of the torsion kernel: every torsion evaluated once with its k-l side delivered through the table (default) or visited from both
    ends (RXMD_E4B_ONCE=0, rounds 1-5); a persistent grid (default) or one workgroup per group of centre atoms (RXMD_E4B_PERSIST=0); workgroups of four
    wavefronts (default) or of one (RXMD_E4B_WPB=1).  One and two visits book the same terms in other places and orders: forces within 1e-10, stress
    accumulators within 1e-9 relative (the two-visit form corrects the frame of the k-l half, the one-visit form needs no correction), energies
    within 1e-12 relative.  The grid and the workgroup size do not change which additions an accumulator sees: bit-identical forces (where no hydrogen bond adds its atomics)."""
    kw = dict(QEq_tol=1e-12, NMAXQEq=2000, **kw)
    res = {}
    for once, pers, wpb in (("1", "1", "4"), ("0", "1", "4"), ("1", "0", "4"), ("1", "1", "1"), ("0", "0", "1")):
        monkeypatch.setenv("RXMD_E4B_ONCE", once); monkeypatch.setenv("RXMD_E4B_PERSIST", pers); monkeypatch.setenv("RXMD_E4B_WPB", wpb)
        e = _engine(case, mc, **kw)
        e.QEq(); pe = e.FORCE(); a = e.atoms(); en = e.energy()
        res[(once, pers, wpb)] = (a["f"].copy(), pe.copy(), en["astr"].copy())
        e.close()
    f1, pe1, s1 = res[("1", "1", "4")]
    f2, pe2, s2 = res[("0", "1", "4")]
    assert np.abs(f1 - f2).max() <= 1e-10
    assert np.allclose(pe1, pe2, rtol=1e-12, atol=1e-10)
    assert np.abs(s1 - s2).max() <= 1e-9 * max(np.abs(s2).max(), 1.0)
    # (the acceptor forces of the hydrogen bonds are FP64 atomics, the one sum of FORCE whose order is not fixed: bit for bit only without them)
    same = (lambda a, b: np.array_equal(a, b)) if pe1[10] == 0.0 else (lambda a, b: np.abs(a - b).max() <= 1e-11)
    assert same(f1, res[("1", "0", "4")][0]) and same(f1, res[("1", "1", "1")][0])
    assert same(f2, res[("0", "0", "1")][0])


@pytest.mark.parametrize("case,mc,kw", [("rdx168", (1, 1, 1), {}), ("rdx222", (2, 2, 2), {}), ("rdx168", (5, 5, 5), {}), ("ice644", (6, 4, 4), {}), ("example1", (2, 3, 5), {}),
                                        ("sicnp", (1, 1, 1), dict(pqeq=oa.PQEQ_SICNP))])
def test_valence_angles_through_the_queue_and_per_thread(case, mc, kw, monkeypatch):
    """Two forms of E3b (pot.F90:319-557): a thread per centre atom (`k_e3b`, RXMD_E3B_QUEUE=0) and sixteen centre atoms per wavefront with the surviving
    angles compacted into a queue and evaluated 64 at a time (`k_e3q`, default for force fields with at most 7 atom types and 255 angle rows: sixteen / eight /
    four centres per wavefront for bond lists up to 12 / 24 / 30; 2 / 3 = the register budget).  The same terms summed in another order: forces within 1e-10, energies within 1e-12 relative, both against the oracle."""
    kw = dict(QEq_tol=1e-12, NMAXQEq=2000, **kw)
    o = _oracle(case, mc, **kw); o.qeq(); o.force()
    res = {}
    for form in ("0", "2", "3"):
        monkeypatch.setenv("RXMD_E3B_QUEUE", form)
        e = _engine(case, mc, **kw)
        e.QEq(); pe = e.FORCE(); a = e.atoms()
        assert f_err(a["f"], o.forces()) <= FTOL and e_err(pe, o.energy()) <= (1e-8 if "pqeq" in kw else ETOL)    # (PQEq: PE(12) is the small remainder of large sums, see above)
        res[form] = (a["f"].copy(), pe.copy())
        e.close()
    for form in ("2", "3"):
        assert np.abs(res[form][0] - res["0"][0]).max() <= 1e-10
        assert np.allclose(res[form][1], res["0"][1], rtol=1e-12, atol=1e-10)


# ---- PQEq (pqeq.F90 / ENbond_PQEq): SiC nanoparticle in O2, conf/init.sicnp, 547 atoms, polarizable shells ----------------
@pytest.mark.parametrize("qeq_mode", [0, 1])
def test_pqeq_step0_against_the_reference_golden(qeq_mode):
    """state after the pre-loop PQEq + FORCE (main.F90:27-32) against the real reference run with --pqeq at tight tolerance"""
    g = np.load(os.path.join(oa.GOLD, "sicnp547_pqeq_tight.npz"))
    kw = dict(QEq_tol=1e-12, NMAXQEq=2000)
    e = _engine("sicnp", (1, 1, 1), pqeq=oa.PQEQ_SICNP, qeq_mode=qeq_mode, **kw)
    it, est = e.QEq(); pe = e.FORCE(); a = e.atoms()
    o = np.argsort(a["gid"]); go = np.argsort(g["gid"])
    assert (a["gid"][o] == g["gid"][go]).all()
    assert q_err(a["q"][o], g["charge"][go]) <= QTOL
    assert f_err(a["f"][o], g["force"][go]) <= FTOL
    orc = _oracle("sicnp", (1, 1, 1), pqeq=oa.PQEQ_SICNP, **kw); orc.qeq(); orc.force()
    # PE(12) is the small remainder of four core/shell Coulomb sums that cancel (+-1e5 -> 70 kcal/mol): 1e-8 of the result
    # is still 1e-11 of the terms
    assert e_err(pe, orc.energy()) <= 1e-8
    assert abs(est - orc.trace()[-1, 0]) <= 1e-9 * abs(est)
    # the first shell move (Eq. 39, clipped to 1e-3 A) happened at the end of that PQEq call
    s = e.shells()
    assert np.abs(s[o] - orc.spos()[np.argsort(orc.gids())]).max() <= 1e-9
    assert 0 < np.linalg.norm(s, axis=1).max() <= 1e-3 * (1 + 1e-12)
    e.close()


def test_pqeq_md_against_the_clean_oracle():
    """5 MD steps with moving shells.  The reference re-uses the previous pair's table value when a core-shell or shell-shell
    distance falls outside the taper cutoff (module.F90:401: early return with untouched outputs; order- and thread-dependent);
    the engine gives such lookups zero weight, which the oracle reproduces with set_pqeq_clean(1).  The reference-faithful
    oracle mode is pinned against the real reference in tests/test_oracle_golden.py and the size of the difference is reported."""
    kw = dict(QEq_tol=1e-12, NMAXQEq=2000)
    e = _engine("sicnp", (1, 1, 1), pqeq=oa.PQEQ_SICNP, **kw)
    o = _oracle("sicnp", (1, 1, 1), pqeq=oa.PQEQ_SICNP, **kw); o.set_pqeq_clean(1)
    e.QEq(); e.FORCE(); o.qeq(); o.force()
    e.step(5); o.step(5)
    a = e.atoms()
    ie = np.argsort(a["gid"]); io = np.argsort(o.gids())
    assert np.abs(a["pos"][ie] - o.pos()[io]).max() <= 1e-9
    assert q_err(a["q"][ie], o.charges()[io]) <= QTOL
    assert f_err(a["f"][ie], o.forces()[io]) <= FTOL
    # shells move by (small net force)/K per call, clipped to 1e-3 A: measured difference 2e-8 A after 6 calls, it follows the
    # 1e-7 relative CG noise of the charges at this tolerance
    assert np.abs(e.shells()[ie] - o.spos()[io]).max() <= 1e-7
    # the Coulomb term PE(12) = 69 kcal/mol is what is left of +-1e5 kcal/mol of core/shell sums and moves with the shells
    # (2e-8 A apart, above): gate every term on the scale of the total energy instead of its own
    pe_e, pe_o = e.energy()["PE"], o.energy()
    assert np.abs(pe_e - pe_o).max() <= 1e-9 * abs(pe_o[0])
    e.close()


@pytest.mark.parametrize("qeq_mode", [0, 1])
def test_pqeq_md_against_the_reference_on_a_cluster_without_stale_lookups(qeq_mode):
    """PQEq MD against the REFERENCE ITSELF (not the oracle's clean switch): the isolated SiC + O2 cluster of
    tests/golden/sicfrag26_pqeq_efieldx_md8.npz (26 atoms, diameter 11.1 A, 40 A box) has no pair near the 12.5 A cut-off, so the
    reference's beyond-cut-off artefact (module.F90:401) cannot occur -- tests/test_oracle_golden.py asserts pqeq_stale() == 0 on it --
    and its 8-step trajectory with the field along x and moving shells is what a clean PQEq gives.  Charges, forces <= 1e-6 of the
    reference's dump, shell displacements against its trajectory frame."""
    g = np.load(os.path.join(oa.GOLD, "sicfrag26_pqeq_efieldx_md8.npz"))
    e = _engine("sicfrag", (1, 1, 1), pqeq=oa.PQEQ_SICNP, efield=(1, 0.05), QEq_tol=1e-12, NMAXQEq=2000, qeq_mode=qeq_mode)
    e.QEq(); e.FORCE(); e.step(8)
    a = e.atoms()
    o = np.argsort(a["gid"]); go = np.argsort(g["gid"])
    assert (a["gid"][o] == g["gid"][go]).all()
    assert np.abs(a["pos"][o] - g["pos"][go]).max() <= 1e-9
    assert q_err(a["q"][o], g["charge"][go]) <= QTOL
    assert f_err(a["f"][o], g["force"][go]) <= FTOL
    t = str(g["xyz_last"]).split("\n")[2:-1]                    # WriteXYZ with PQEq (fileio.F90:241-355): ..., global id i9, shell displacement 3es20.12
    ids = np.array([int(l[83:92]) for l in t]); sp = np.array([[float(x) for x in l[92:].split()] for l in t])
    so = np.argsort(ids)
    assert (ids[so] == a["gid"][o]).all()
    assert np.linalg.norm(sp, axis=1).max() > 1e-3               # the shells have moved
    assert np.abs(e.shells()[o] - sp[so]).max() <= 1e-8          # clipped 1e-3 A moves along (force / K): follows the charges' 1e-7 relative CG noise
    e.close()


def test_pqeq_md_on_the_periodic_nanoparticle_stays_within_the_documented_bound_of_the_reference():
    """The one designed deviation from the reference, pinned against the REFERENCE ITSELF (Deviations): on the periodic
    547-atom SiC nanoparticle in O2 the reference's beyond-cut-off look-ups (module.F90:401: outputs untouched, the callers pqeq.F90:305-333,
    381-411 then re-use the previous pair's values) do occur once the shells have moved -- 170 times in these 5 steps -- and the engine gives
    them zero weight.  tests/golden/sicnp547_pqeq_md5.npz is the reference's own run (rxmd --pqeq, rxmd.in as shipped: QEq tol 1e-7, 5 steps).
    Size of the artefact alone (restatement with and without it at tight tolerance): |dq| <= 3.9e-3, |df| <= 2.0e-3 kcal/mol/A; here, against the
    reference's run at its own tolerance, both are asserted below 5e-3, so nothing else can hide behind the clean-mode comparisons above.  Step 0 of the same input has no such look-up and is compared at 1e-6
    in test_pqeq_step0_against_the_reference_golden."""
    g = np.load(os.path.join(oa.GOLD, "sicnp547_pqeq_md5.npz"))
    e = _engine("sicnp", (1, 1, 1), pqeq=oa.PQEQ_SICNP, QEq_tol=1e-7, NMAXQEq=500)
    e.QEq(); e.FORCE(); e.step(5)
    a = e.atoms()
    o = np.argsort(a["gid"]); go = np.argsort(g["gid"])
    assert (a["gid"][o] == g["gid"][go]).all()
    dq = np.abs(a["q"][o] - g["charge"][go]).max(); df = np.abs(a["f"][o] - g["force"][go]).max()
    assert np.abs(a["pos"][o] - g["pos"][go]).max() <= 5e-6       # positions after 5 steps of 0.25 fs with forces up to 2e-3 kcal/mol/A apart: measured 1.5e-6 A
    print("PQEq periodic MD vs the reference's own run: max |dq| %.3e  max |df| %.3e kcal/mol/A" % (dq, df))
    # DESIGN 5b measured the artefact alone (faithful against clean restatement at tight tolerance): |dq| 3.9e-3, |df| 2.0e-3.  Against the reference's
    # run at ITS tolerance (1e-7) the CG's exit noise rides on top (charges to ~1e-5 -> forces to ~1e-3): measured here |df| = 3.5e-3
    assert dq <= 5e-3, dq
    assert df <= 5e-3, df
    # ... and the deviation is real, not noise: were it zero, this test would pin nothing
    assert dq > 1e-6 or df > 1e-6
    e.close()


def test_skewed_box_pair_selection_and_stress_match_the_oracle():
    """conf/init.mos2 (gamma = 120 degrees) against the oracle, which reproduces the real reference on this input to dump resolution
    (tests/test_oracle_golden.py): the bonded and 10 A lists hold exactly the reference's pairs (its cell meshes are laid out along the
    lattice vectors and drop pairs a cutoff sphere would hold), hessian row sums, energies, and the stress accumulators with the
    torsion image correction along skewed lattice vectors."""
    kw = dict(QEq_tol=1e-12, NMAXQEq=2000)
    o = _oracle("mos2_tri324", (3, 3, 2), **kw); o.qeq(); o.force()
    e = _engine("mos2_tri324", (3, 3, 2), **kw)
    it, est = e.QEq(); pe = e.FORCE(); a = e.atoms()
    assert (a["gid"] == o.gids()).all()
    st = e.stats()
    assert st["natoms"] + st["nghost_force"] == o.L.rxo_nghost_total(o.w, 0)
    assert (e.debug(2).astype(int) == o.get(103).astype(int)).all()          # bonded neighbour counts, residents and ghosts
    assert (e.debug(6).astype(int) == o.get(104).astype(int)).all()          # 10 A row lengths
    assert np.allclose(e.debug(7), o.get(108), rtol=1e-12)                   # hessian row sums
    assert q_err(a["q"], o.charges()) <= QTOL
    assert f_err(a["f"], o.forces()) <= FTOL
    assert e_err(pe, o.energy()) <= ETOL
    a0, b0 = e.energy()["astr"], o.astr(reset=True)
    assert np.abs(a0 - b0).max() <= 1e-8 * np.abs(b0).max()
    e.step(2); o.step(2)
    a1, b1 = e.energy()["astr"], o.astr(reset=True)
    assert np.abs(a1 - b1).max() <= 1e-8 * np.abs(b1).max()
    e.close()


def test_stress_accumulators_match_the_oracle():
    """astr(1:6): virial over residents+ghosts before the fold (pot.F90:65-72) + kinetic part per step (main.F90:86-94),
    accumulated between reads like PRINTE; the oracle's values are pinned to the reference's printed pressure column"""
    e = _engine("rdx222", (2, 2, 2), QEq_tol=1e-12, NMAXQEq=2000)
    o = _oracle("rdx222", (2, 2, 2), QEq_tol=1e-12, NMAXQEq=2000)
    e.QEq(); e.FORCE(); o.qeq(); o.force()
    a0, b0 = e.energy()["astr"], o.astr(reset=True)
    assert np.abs(a0 - b0).max() <= 1e-8 * np.abs(b0).max()
    e.step(3); o.step(3)
    a1, b1 = e.energy()["astr"], o.astr(reset=True)
    assert np.abs(a1 - b1).max() <= 1e-8 * np.abs(b1).max()
    assert np.abs(e.energy()["astr"]).max() == 0.0      # reading resets
    e.close()


@pytest.mark.parametrize("qeq_mode", [0, 1])
def test_extended_lagrangian_charges_isQEq2(qeq_mode):
    """isQEq = 2: one CG step per MD step from qs = fqs*qsfp + (1-fqs)*q (qeq.F90:51-57); the fictitious charges are
    integrated with the atoms (main.F90:67-68,98).  10 steps against the real reference's dump."""
    g = np.load(os.path.join(oa.GOLD, "rdx168_lex_md10.npz"))
    e = _engine("rdx168", (1, 1, 1), isQEq=2, qeq_mode=qeq_mode)
    e.QEq(); e.FORCE(); e.step(10)
    a = e.atoms()
    o = np.argsort(a["gid"]); go = np.argsort(g["gid"])
    assert np.abs(a["pos"][o] - g["pos"][go]).max() <= 1e-9
    assert q_err(a["q"][o], g["charge"][go]) <= QTOL
    assert f_err(a["f"][o], g["force"][go]) <= FTOL
    e.close()


@pytest.mark.parametrize("win", ["1", "0"])
def test_extended_lagrangian_mode_through_either_matrix_pass(win, monkeypatch):
    """isQEq = 2 (one CG step per MD step): since the 10 A sweep writes the window form itself (round 4) it exists in every mode and the window
    pass runs here too; RXMD_SPMV_WIN=0 is the row pass -- same trajectory either way (the reference's dump of 10 steps)."""
    monkeypatch.setenv("RXMD_SPMV_WIN", win)
    g = np.load(os.path.join(oa.GOLD, "rdx168_lex_md10.npz"))
    e = _engine("rdx168", (1, 1, 1), isQEq=2, qeq_mode=1)
    e.QEq(); e.FORCE(); e.step(10)
    a = e.atoms(); st = e.stats()
    assert st["win_in_use"] == int(win) and (st["natoms"] + 15) // 16 <= st["win_groups"] <= st["natoms"] // 16 + st["cells10"][0] * st["cells10"][1] + 1, st
    o = np.argsort(a["gid"]); go = np.argsort(g["gid"])
    assert np.abs(a["pos"][o] - g["pos"][go]).max() <= 1e-9
    assert q_err(a["q"][o], g["charge"][go]) <= QTOL
    assert f_err(a["f"][o], g["force"][go]) <= FTOL
    e.close()


@pytest.mark.parametrize("mode,kw", [(4, dict(vsfact=0.9)), (5, dict(treq=300.0)), (7, dict(treq=300.0)), (8, dict(treq=300.0))])
def test_velocity_scaling_modes_from_a_restart_file(mode, kw):
    """mdmode 4/5/7/8 on the device (rxmd_hip_thermostat) continued from the reference's own restart file (rxff.bin after 20 NVE
    steps, loaded with set_atoms_rxff), 7 steps with sstep 3, against the reference's dump of the last step"""
    import rxmd_amd
    g = np.load(os.path.join(oa.GOLD, "rdx168_thermo%d.npz" % mode))
    ff = oa.make_system("rdx168")[0]
    lat, vp, step, recs = oa.parse_rxff(g["restart_rxff"])
    e = rxmd_amd.RxmdEngine(ff, lat, QEq_tol=1e-12, NMAXQEq=2000)
    e.set_atoms_rxff(recs[0])
    e.QEq(); e.FORCE()
    for nstep in range(7):
        if nstep % 3 == 0:
            e.thermostat(mode, **kw)
        e.step(1)
    a = e.atoms()
    o = np.argsort(a["gid"]); go = np.argsort(g["gid"])
    assert np.abs(a["pos"][o] - g["pos"][go]).max() <= 1e-9
    assert q_err(a["q"][o], g["charge"][go]) <= QTOL
    assert f_err(a["f"][o], g["force"][go]) <= FTOL
    e.close()


def test_pqeq_electric_field_step0_against_the_reference():
    """rxmd.in `efield 1 0.05`: field force on the cores (EEfield) on top of the PQEq forces, against the real reference (x direction:
    the one the reference's EEfield addresses correctly, see include/rxmd_hip.h)"""
    g = np.load(os.path.join(oa.GOLD, "sicnp547_pqeq_efieldx_0.npz"))
    e = _engine("sicnp", (1, 1, 1), pqeq=oa.PQEQ_SICNP, efield=(1, 0.05), QEq_tol=1e-12, NMAXQEq=2000)
    e.QEq(); e.FORCE(); a = e.atoms()
    o = np.argsort(a["gid"]); go = np.argsort(g["gid"])
    assert q_err(a["q"][o], g["charge"][go]) <= QTOL
    assert f_err(a["f"][o], g["force"][go]) <= FTOL
    e.close()


def test_pqeq_electric_field_md_against_the_clean_oracle():
    """field along z, 3 MD steps: core and shell field forces, momentum removal between kick and drift (main.F90:70-71).  Oracle in
    clean mode = field applied to atom i along dir and no stale table values (the reference-faithful mode is pinned on CPU)."""
    kw = dict(QEq_tol=1e-12, NMAXQEq=2000)
    e = _engine("sicnp", (1, 1, 1), pqeq=oa.PQEQ_SICNP, efield=(3, 0.05), **kw)
    o = _oracle("sicnp", (1, 1, 1), pqeq=oa.PQEQ_SICNP, **kw); o.set_efield(3, 0.05); o.set_pqeq_clean(1)
    e.QEq(); e.FORCE(); o.qeq(); o.force()
    e.step(3); o.step(3)
    a = e.atoms()
    ie = np.argsort(a["gid"]); io = np.argsort(o.gids())
    assert np.abs(a["pos"][ie] - o.pos()[io]).max() <= 1e-9
    assert q_err(a["q"][ie], o.charges()[io]) <= QTOL
    assert f_err(a["f"][ie], o.forces()[io]) <= FTOL
    assert np.abs(e.shells()[ie] - o.spos()[io]).max() <= 1e-7
    e.close()


def test_low_gradient_dispersion_against_the_reference_and_the_oracle():
    """--lg on the reference's conf/init.rdx.lg input: LG ffield format, low-gradient + core terms in the vdW table (init.F90:496-514).
    Step 0 against the real reference, then 5 MD steps against the oracle (itself pinned on that trajectory on CPU)."""
    kw = dict(QEq_tol=1e-12, NMAXQEq=2000, lg=True)
    g = np.load(os.path.join(oa.GOLD, "rdx168_lg_tight.npz"))
    e = _engine("rdx168_lg", (1, 1, 1), **kw)
    e.QEq(); pe = e.FORCE(); a = e.atoms()
    assert (a["gid"] == g["gid"]).all()
    assert q_err(a["q"], g["charge"]) <= QTOL
    assert f_err(a["f"], g["force"]) <= FTOL
    assert abs(pe[0] / 168 - g["mdstep"][0][2]) <= 1e-5 * abs(g["mdstep"][0][2])
    o = _oracle("rdx168_lg", (1, 1, 1), **kw); o.qeq(); o.force()
    assert e_err(pe, o.energy()) <= ETOL
    e.step(5); o.step(5); a = e.atoms()
    assert np.abs(a["pos"] - o.pos()).max() <= 1e-9
    assert q_err(a["q"], o.charges()) <= QTOL
    assert f_err(a["f"], o.forces()) <= FTOL
    e.close()


def test_reference_example3_small_box_pqeq_with_field_against_the_clean_oracle():
    """the system of the reference's examples/3-reaxpq+ (polyethylene 2x3x5: a 14.8 x 14.8 x 12.7 A box under the 12.5 A PQEq cut-off,
    parameter file listing only C and H, field along x), 5 MD steps at tight tolerance against the oracle in clean mode; the oracle's
    reference-faithful mode is pinned to the real reference's frame of this example on CPU (tests/test_oracle_golden.py)"""
    kw = dict(QEq_tol=1e-12, NMAXQEq=2000)
    e = _engine("example3", (2, 3, 5), pqeq=oa.PQEQ_EXAMPLE3, efield=(1, 0.01), **kw)
    o = _oracle("example3", (2, 3, 5), pqeq=oa.PQEQ_EXAMPLE3, **kw); o.set_efield(1, 0.01); o.set_pqeq_clean(1)
    e.QEq(); e.FORCE(); o.qeq(); o.force()
    e.step(5); o.step(5)
    a = e.atoms()
    ie = np.argsort(a["gid"]); io = np.argsort(o.gids())
    assert np.abs(a["pos"][ie] - o.pos()[io]).max() <= 1e-9
    assert q_err(a["q"][ie], o.charges()[io]) <= QTOL
    assert f_err(a["f"][ie], o.forces()[io]) <= FTOL
    assert np.abs(e.shells()[ie] - o.spos()[io]).max() <= 1e-7
    e.close()


@pytest.mark.parametrize("case", ["fes576_md3", "mos2_216_md3", "mos2_tri324_md3", "sic512_md3", "aloslab180_md3", "pbt2272_md2"])
@pytest.mark.parametrize("qeq_mode", [0, 1])
def test_other_force_fields_and_systems_the_reference_ships(case, qeq_mode):
    """mos2_tri324: the one NON-ORTHORHOMBIC input of the reference (conf/init.mos2: hexagonal 2H-MoS2, gamma = 120 degrees, 3 x 3 x 2) -- the
    full H / HHi transforms, the engine's grid measured perpendicular to the cell faces, and the reference's pair selection through its
    lattice-vector cell meshes and QEq ghost shell (RefMesh, engine.h).
    pyrite (10-type ffield), MoS2, zinc-blende SiC, an alumina slab and an amorphous polymer cell (PBT, 2,272 atoms) from the reference's conf/: 2-3 MD steps at tight tolerance
    against the goldens of the real reference (state after the third step: positions, charges, forces)"""
    g = np.load(os.path.join(oa.GOLD, case + ".npz"))
    e = _engine(case, tuple(int(x) for x in g["mc"]), QEq_tol=1e-12, NMAXQEq=2000, qeq_mode=qeq_mode)
    e.QEq(); e.FORCE(); e.step(int(g["nsteps"]))
    a = e.atoms()
    o = np.argsort(a["gid"]); go = np.argsort(g["gid"])
    assert (a["gid"][o] == g["gid"][go]).all()
    assert np.abs(a["pos"][o] - g["pos"][go]).max() <= 1e-9
    assert q_err(a["q"][o], g["charge"][go]) <= QTOL
    assert f_err(a["f"][o], g["force"][go]) <= FTOL
    e.close()


@pytest.mark.parametrize("case,mc,kw", [("rdx222", (2, 2, 2), {}), ("rdx168", (5, 5, 5), {}), ("ice644", (6, 4, 4), {}), ("rdx222", (2, 2, 2), {"isQEq": 2}), ("rdx168", (1, 1, 1), {"qstep": 3})])
def test_bonded_chain_on_its_own_stream_gives_the_same_trajectory(case, mc, kw, monkeypatch):
    """The charge-free part of FORCE (bond orders, bonded terms, assembly of the bonded forces) runs on a stream of its own next to the QEq
    iterations and ENbond (engine.h: bond_stream) when RXMD_BOND_OVERLAP=1; the default is the one-stream order.  Same kernels on the same inputs: what
    differs is that ENbond's force is added behind the bonded forces instead of before them -- the last bit of a sum.  Standalone calls
    (QEq, FORCE) and 8 steps through step(), with every QEq setting that changes what is in flight when the chain starts."""
    res = {}
    for overlap in (True, False):
        monkeypatch.setenv("RXMD_BOND_OVERLAP", "1" if overlap else "0")
        e = _engine(case, mc, QEq_tol=1e-12, NMAXQEq=2000, qeq_mode=1, **kw)
        e.QEq(); pe0 = np.array(e.FORCE()); a0 = e.atoms()
        assert e.stats()["bond_overlap"] == int(overlap)
        e.step(8)
        en = e.energy(); a1 = e.atoms()
        res[overlap] = (a0["f"].copy(), pe0, a1["pos"].copy(), a1["f"].copy(), a1["q"].copy(), np.array(en["PE"]), np.array(en["astr"]))
        e.close()
    on, off = res[True], res[False]
    assert f_err(on[0], off[0]) <= 1e-12 and e_err(on[1], off[1]) <= 1e-12          # (the order of ENbond's addition, and of the hydrogen bonds' acceptor atomics: 1.2e-13 seen)
    assert np.abs(on[2] - off[2]).max() <= 1e-10                      # positions after 8 steps
    assert f_err(on[3], off[3]) <= 1e-8 and q_err(on[4], off[4]) <= 1e-8
    assert e_err(on[5][1:14], off[5][1:14]) <= 1e-9
    assert np.abs(on[6] - off[6]).max() <= 1e-8 * max(np.abs(off[6]).max(), 1.0)        # the stress sums (both chains add to them)


@pytest.mark.parametrize("qeq_mode", [0, 1])
def test_charges_every_third_step_only(qeq_mode):
    """rxmd.in `QEq 1 2000 1.d-12 3` (qstep = 3, main.F90:77): 7 MD steps against the real reference"""
    g = np.load(os.path.join(oa.GOLD, "rdx168_qstep3_md7.npz"))
    e = _engine("rdx168", (1, 1, 1), QEq_tol=1e-12, NMAXQEq=2000, qstep=3, qeq_mode=qeq_mode)
    e.QEq(); e.FORCE(); e.step(7)
    assert e.stats()["qeq_calls"] == 4                              # the pre-loop call + steps 0, 3, 6
    a = e.atoms()
    assert (a["gid"] == g["gid"]).all()
    assert np.abs(a["pos"] - g["pos"]).max() <= 1e-9
    assert q_err(a["q"], g["charge"]) <= QTOL
    assert f_err(a["f"], g["force"]) <= FTOL
    e.close()


def test_row_stride_of_the_10A_list_grows_when_the_density_estimate_is_too_low():
    """The list stride is sized from the mean density; a tiny explicit hint (maxneighbs10) keeps the reference's overflow trap,
    the automatic sizing must recover by growing to what the sweep reports."""
    import rxmd_amd
    from rxmd_amd import system
    ff, names, frac, lat = oa.make_system("sicnp")
    lat3, rec = system.geninit(ff, names, frac, lat, mc=(1, 1, 1))
    e = rxmd_amd.RxmdEngine(ff, lat3, maxneighbs10=64)
    e.set_atoms_rxff(rec)
    with pytest.raises(rxmd_amd.RxmdError) as ex:
        e.QEq()
    assert ex.value.code == -5 and "MAXNEIGHBS10" in str(ex.value)      # RXMD_E_MAXNEIGHBS10, qeq.F90:248-252
    e.close()
    # a box that is mostly vacuum: widen it so that the mean density underestimates the rows of the particle
    lat_wide = list(lat3); lat_wide[0] *= 3.0
    rec2 = rec.copy(); rec2[:, 0] = rec2[:, 0] / 3.0
    e = rxmd_amd.RxmdEngine(ff, lat_wide)
    e.set_atoms_rxff(rec2)
    s0 = e.stats()["n10_stride"]
    e.QEq(); e.FORCE()
    s1 = e.stats()
    assert s1["n10_stride"] > s0 and s1["max_n10"] <= s1["n10_stride"]
    e.close()


def test_forces_and_charges_are_bitwise_reproducible_run_to_run():
    """no result may depend on scheduling: partial sums are combined in fixed orders, the torsion accumulators use LDS atomics whose
    order is fixed by queue and lane order; only the 14 energy scalars and the hydrogen-bond acceptor forces take global atomics.
    RDX 2x2x2 (H-bond acceptor atomics present): charges bitwise, forces to 1e-13 relative; the ice case (Ehb == 0): forces bitwise."""
    res = []
    for rep in range(2):
        e = _engine("rdx222", (2, 2, 2), QEq_tol=1e-12, NMAXQEq=2000)
        e.QEq(); e.FORCE(); e.step(2)
        a = e.atoms(); res.append((a["q"].copy(), a["f"].copy())); e.close()
    assert np.array_equal(res[0][0], res[1][0])
    assert np.abs(res[0][1] - res[1][1]).max() <= 1e-13 * np.abs(res[0][1]).max()
    res = []
    for rep in range(2):
        e = _engine("ice644", (6, 4, 4), QEq_tol=1e-12, NMAXQEq=2000)
        e.QEq(); e.FORCE()
        a = e.atoms(); res.append((a["q"].copy(), a["f"].copy())); e.close()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])


@pytest.mark.parametrize("qeq_mode", [0, 1])
def test_full_size_nve_trajectory_conserves_energy(qeq_mode):
    """BASELINE configs[1] size, 20 MD steps (dt 0.25 fs, QEq tol 1e-7 as in the bench): total energy per atom stays where the
    reference keeps it (its own RDX-168 run moves by 2e-4 kcal/mol/atom over 15 steps), net charge stays zero, no atom is lost"""
    e = _engine("rdx168", (18, 18, 18), qeq_mode=qeq_mode)
    e.QEq(); pe = e.FORCE()
    n = e.natoms
    en0 = e.energy(); te0 = (en0["PE"][0] + en0["KE"]) / n
    e.step(20)
    en1 = e.energy(); te1 = (en1["PE"][0] + en1["KE"]) / n
    assert e.natoms == n
    assert abs(te1 - te0) <= 1e-3
    assert en1["KE"] / n > 1e-5                      # the crystal started at rest and is moving now
    assert abs(en1["qsum"]) <= 1e-6
    e.close()


def test_random_velocities_mdmode_0_and_6():
    """mdmode 0 / 6 (main.F90:54-55 -> INITVELOCITY, init.F90:292-360) on the device: random, so the checks are the properties the reference's
    routine guarantees -- temperature exactly treq, zero total momentum, unit-variance Gaussian components with the SAME variance for every
    element (the reference does not weight by mass) -- plus what the counter-based generator adds: an atom's draw depends on its global id
    only (not on its local index: the same atoms handed over in another order get the same velocities), and a second call draws anew."""
    import rxmd_amd
    from rxmd_amd import system
    UTEMP0 = 503.398008
    ff, names, frac, lat = oa.make_system("rdx168")
    lat3, rec = system.geninit(ff, names, frac, lat, mc=(6, 6, 6))
    n = len(rec)
    lines = open(ff).read().split("\n"); npar = int(lines[1].split()[0]); nso = int(lines[2 + npar].split()[0])
    mass = {t + 1: float(lines[2 + npar + 4 + 4 * t][3:].split()[2]) for t in range(nso)}       # param.F90:102-104: name, rat, Val, mass
    res = []
    for order in (np.arange(n), np.random.default_rng(3).permutation(n)):
        e = rxmd_amd.RxmdEngine(ff, lat3)
        e.set_atoms_rxff(rec[order])
        e.thermostat(6, treq=300.0)
        a = e.atoms(); en = e.energy()
        o = np.argsort(a["gid"])
        res.append((a["gid"][o], a["v"][o], a["type"][o], en["KE"]))
        if len(res) == 1:
            e.thermostat(0, treq=450.0)
            a2 = e.atoms(); en2 = e.energy()
            v2 = a2["v"][np.argsort(a2["gid"])]
        e.close()
    gid, v, typ, ke = res[0]
    # temperature: KE per atom = 1.5 treq in the reference's units (init.F90:355-358), i.e. T = KE/n * UTEMP = treq
    assert abs(ke / n * UTEMP0 * 2.0 / 3.0 - 300.0) <= 1e-9 * 300.0
    assert abs(en2["KE"] / n * UTEMP0 * 2.0 / 3.0 - 450.0) <= 1e-9 * 450.0
    m = np.array([mass[int(t)] for t in typ])
    p = (m[:, None] * v).sum(axis=0)
    assert np.abs(p).max() <= 1e-9 * (m[:, None] * np.abs(v)).sum()          # centre-of-mass velocity removed
    # unit-variance Gaussian before the common scale factor: same variance for every element, kurtosis 3, components uncorrelated
    s_all = v.std()
    for t in (1, 2, 3, 4):
        vt = v[typ == t]
        assert abs(vt.std() / s_all - 1.0) <= 0.03
    z = v.reshape(-1) / s_all
    assert abs(z.mean()) <= 0.01 and abs((z ** 4).mean() - 3.0) <= 0.06 and abs((z ** 3).mean()) <= 0.03
    c = np.corrcoef(v.T)
    assert np.abs(c - np.eye(3)).max() <= 0.02
    # keyed by the global id: another local order, the same velocities
    assert np.array_equal(res[1][0], gid)
    assert np.abs(res[1][1] - v).max() <= 1e-12 * np.abs(v).max()
    # a second call is a new draw
    assert np.abs(np.corrcoef(v2.reshape(-1), v.reshape(-1))[0, 1]) <= 0.02


def test_poisoned_allocations():
    """RXMD_POISON_ALLOC=1 (engine.hip): all engine memory starts as 0xFF bytes and the per-step scratch is re-filled with the pattern before every
    rebuild.  RDX 2 x 2 x 2 and the SiC nanoparticle with PQEq, both QEq algebras, QEq + FORCE + MD steps against the oracle: nothing may read what
    nobody wrote (the reference's allocator does not clear either, module.F90:732-744; what it clears -- pot.F90:20-26, init.F90:117-131 -- the
    kernels clear).  The whole GPU suite was run once under the switch in round 4 (120 passed)."""
    import subprocess, sys
    env = dict(os.environ, RXMD_POISON_ALLOC="1")
    p = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "poison_worker.py")], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0 and "POISON-OK" in p.stdout, p.stderr[-3000:]


def test_bond_tables_grow_on_demand(monkeypatch):
    """The compact bond tables start at 12 bonds per atom slot and grow to what a list build reports (Engine::build_ghosts_and_lists): with a
    capacity of 1,024 bonds (RXMD_BOND_CAP) RDX 2 x 2 x 2 -- ~40,000 bonds with its ghosts -- has to grow them in its first build, and must
    give the same charges, forces and energies as the run that never grew."""
    kw = dict(QEq_tol=1e-12, NMAXQEq=2000)
    ref = _engine("rdx222", (2, 2, 2), qeq_mode=1, **kw); ref.QEq(); pe0 = ref.FORCE(); ref.step(2); a0 = ref.atoms(); ref.close()
    monkeypatch.setenv("RXMD_BOND_CAP", "1024")
    e = _engine("rdx222", (2, 2, 2), qeq_mode=1, **kw); e.QEq(); pe = e.FORCE()
    assert e.stats()["nbonds"] > 1024
    e.step(2); a = e.atoms(); e.close()
    # (to the order of the energy sums and of the acceptor atomics of Ehb, the only additions whose order is not fixed)
    assert e_err(pe, pe0) <= 1e-12
    assert q_err(a["q"], a0["q"]) <= 1e-12 and f_err(a["f"], a0["f"]) <= 1e-11 and np.abs(a["pos"] - a0["pos"]).max() <= 1e-12
