"""Parity where the benchmark runs.

The small golden systems (<= 2,304 atoms) are one or two cut-offs wide: every 10 A row has a ghost partner, the engine's grid has two or
three cells per edge, no workgroup-order remap matters.  These tests compare the HIP path with the oracle in the regime of the
benchmark instead:

 * RDX 6 x 6 x 6 = 36,288 atoms with seeded Gaussian displacements (sigma 0.05 A) and velocities: 6 cell columns per edge, a non-empty
   interior row set (rows without a ghost partner) for the two-launch matrix pass of the multi-rank path, migration every step.
   Lists, bond orders, the delta / delta' / cdbnd / ccbnd intermediates, charges, forces, 14 energies; 3 MD steps; both QEq algebras; and the
   same trajectory through the staged six-stage exchange with RCCL self send/recv, halo overlap on and off.
 * per-atom FORCES of the 979,776-atom crystal (BASELINE configs[1]): by periodicity every interior unit cell of the 18^3 replication
   equals an interior cell of a 4 x 4 x 4 oracle run -- same bonded neighbours in the same relative index order, so the index-ordered
   ccbnd rule (pot.F90:113-144) gives the same result.

How the charges are compared on the perturbed crystal.  On a thermally disordered 36k-atom system the reference's two-vector CG does
not reach QEq_tol 1e-12 at all: the s system converges (|gs|^2 1e-14 after 150 iterations) but the t system (H t = -1) crawls --
|gt|^2 = 6e-2 / 6e-3 / 7e-4 / 6e-5 after 150 / 300 / 600 / 1400 iterations (oracle, measured here), the REAL(4) step length
(qeq.F90:23,133) having cost the directions their conjugacy -- and the exit test can fire by chance on an extremum of Est
(98 of 100 iterations in one step of this very trajectory).  An exit that depends on the 13th digit of Est is no basis for a parity
test, so both sides run a FIXED number of iterations (QEq_tol 1e-300 never fires, NMAXQEq = 100 per call): the same algorithm from the
same start must give the same iterate, to the parity tolerance, at the same iteration.
"""
import numpy as np
import pytest

import oracle_api as oa
from test_gpu_parity import q_err, f_err, e_err, compare_bond_order_taps, QTOL, FTOL, ETOL

pytestmark = pytest.mark.gpu

MC = (6, 6, 6)
KW = dict(QEq_tol=1e-300, NMAXQEq=100)
NSTEPS = 3


def _perturbed_rdx(vprocs=(1, 1, 1)):
    """RDX 6x6x6 as geninit lays it out, every atom displaced by N(0, 0.05 A) per component, velocities N(0, 0.05) (230 K); with vprocs the
    atoms are dealt to the domains of the rank grid the way geninit does (geninit.F90:493-527) -> (ffield, lattice, per-rank dicts, per-rank v)"""
    ff, names, frac, lat = oa.make_system("rdx168")
    lat2, ranks = oa.geninit(names, frac, lat, oa.ffield_names(ff), mc=MC)
    rng = np.random.default_rng(2026)
    n = len(ranks[0]["type"])
    rn = ranks[0]["rnorm"] + rng.normal(0, 0.05, (n, 3)) / np.asarray(lat2[:3])
    # back into [0, 1) like geninit (geninit.F90:476-480) and every COPYATOMS(MODE_MOVE) leave them: a resident outside the box is not
    # a state the reference's driver produces (its QEq ghost shell of exactly rctap would miss partners of such an atom)
    rn = rn - np.floor(rn)
    v = rng.normal(0, 0.05, (n, 3))
    if tuple(vprocs) == (1, 1, 1):
        ranks[0]["rnorm"] = rn
        return ff, lat2, ranks, v
    vp = np.array(vprocs)
    dom = (rn * vp).astype(np.int64)
    sid = dom[:, 0] + dom[:, 1] * vp[0] + dom[:, 2] * vp[0] * vp[1]
    out, vs = [], []
    for p in range(int(vp.prod())):
        sel = np.nonzero(sid == p)[0]
        obox = (1.0 / vp) * np.array([p % vp[0], (p // vp[0]) % vp[1], p // (vp[0] * vp[1])])
        out.append(dict(rnorm=rn[sel] - obox, type=ranks[0]["type"][sel].copy(), gid=ranks[0]["gid"][sel].copy()))
        vs.append(v[sel].copy())
    return ff, lat2, out, vs


@pytest.fixture(scope="module")
def oracle36k():
    ff, lat2, ranks, v = _perturbed_rdx()
    n = len(ranks[0]["type"])
    o = oa.Oracle(ff, lat2, ranks, nbuffer=8 * n, v0=[v], **KW)
    o.qeq(); o.force()
    nbr, bo = o.bonds()
    s0 = dict(gid=o.gids().copy(), q=o.charges().copy(), f=o.forces().copy(), pe=o.energy().copy(), est=o.trace()[-1, 0],
              G=o.L.rxo_nghost_total(o.w, 0), gidG=o.get(106).copy(), nbrcnt=o.get(103).copy(), n10=o.get(104).copy(), hsum=o.get(108).copy(),
              delta=o.get(101).copy(), deltap=o.get(102).copy(), ccused=o.get(109).copy(), cdbnd=o.get(110).copy(), nbr=nbr, bo=bo,
              posG=o.get(100, width=3).copy(), astr=o.astr(reset=True))
    o.step(NSTEPS)
    s1 = dict(gid=o.gids().copy(), pos=o.pos().copy(), vel=o.vel().copy(), q=o.charges().copy(), f=o.forces().copy(), pe=o.energy().copy(),
              ke=o.kinetic())
    return s0, s1


def _engine36k(qeq_mode):
    import rxmd_amd
    ff, lat2, ranks, v = _perturbed_rdx()
    r = ranks[0]
    n = len(r["type"])
    rec = np.zeros((n, 10))
    rec[:, 0:3] = r["rnorm"]; rec[:, 3:6] = v; rec[:, 7] = r["type"] + r["gid"] * 1e-13
    e = rxmd_amd.RxmdEngine(ff, lat2, qeq_mode=qeq_mode, **KW)
    e.set_atoms_rxff(rec)
    return e


def _check_step0(e, s0, it, est, pe):
    a = e.atoms()
    n = len(s0["gid"])
    assert it == KW["NMAXQEq"]
    assert np.array_equal(a["gid"], s0["gid"])
    st = e.stats()
    assert st["natoms"] + st["nghost_force"] == s0["G"]
    # lists: the same ghosts in the same order, the same bonded and 10 A row lengths, the same matrix rows
    assert np.array_equal(e.debug(4).astype(np.int64), s0["gidG"].astype(np.int64))
    assert np.array_equal(e.debug(2).astype(int), s0["nbrcnt"].astype(int))
    assert np.array_equal(e.debug(6).astype(int), s0["n10"].astype(int))
    assert np.allclose(e.debug(7), s0["hsum"], rtol=1e-12)
    # a9 / a10 / a18: delta' (BOPRIM, bo.F90:28-118), delta and corrected bond orders (BOFULL, bo.F90:121-298), cdbnd, ccbnd (pot.F90:113-144)
    cnt, pg, bo = e.bonds()
    ndeep = compare_bond_order_taps(dict(deltap=e.debug(1), delta=e.debug(0), cd=e.debug(8), cc=e.debug(10), cnt=cnt, pg=pg, bo=bo),
                                    dict(deltap=s0["deltap"], delta=s0["delta"], cd=s0["cdbnd"], cc=s0["ccused"], cnt=s0["nbrcnt"][:n], nbr=s0["nbr"][:n],
                                         bo=s0["bo"][:n], gidG=s0["gidG"].astype(np.int64), pos=s0["posG"]), n, np.asarray(e.lattice[:3]))
    assert ndeep > 0.25 * n                      # ccbnd compared index by index on the atoms no torsion across the box reaches
    assert q_err(a["q"], s0["q"]) <= QTOL
    assert f_err(a["f"], s0["f"]) <= FTOL
    assert e_err(pe, s0["pe"]) <= ETOL
    assert abs(est - s0["est"]) <= 1e-9 * abs(est)


def _check_trajectory(e, s1, s0):
    a = e.atoms()
    assert not np.array_equal(s1["gid"], s0["gid"]), "the test needs migration"
    assert np.array_equal(a["gid"], s1["gid"])                      # the reference's local order after three COPYATOMS(MODE_MOVE)
    assert np.abs(a["pos"] - s1["pos"]).max() <= 1e-9
    assert np.abs(a["v"] - s1["vel"]).max() <= 1e-9
    assert q_err(a["q"], s1["q"]) <= QTOL
    assert f_err(a["f"], s1["f"]) <= FTOL
    en = e.energy()
    assert e_err(en["PE"], s1["pe"]) <= 1e-8                        # three steps of roundoff-level trajectory differences
    assert abs(en["KE"] - s1["ke"]) <= 1e-9 * abs(s1["ke"])


@pytest.mark.parametrize("qeq_mode", [0, 1])
def test_perturbed_rdx_36k_against_the_oracle(qeq_mode, oracle36k):
    s0, s1 = oracle36k
    e = _engine36k(qeq_mode)
    it, est = e.QEq(); pe = e.FORCE()
    _check_step0(e, s0, it, est, pe)
    a0 = e.energy()["astr"]
    assert np.abs(a0 - s0["astr"]).max() <= 1e-8 * np.abs(s0["astr"]).max()
    e.step(NSTEPS)
    _check_trajectory(e, s1, s0)
    e.close()


@pytest.mark.parametrize("qeq_mode,overlap,direct", [(1, True, False), (1, False, False), (0, True, False), (1, True, True)])
def test_perturbed_rdx_36k_through_the_multi_rank_path(qeq_mode, overlap, direct, oracle36k, monkeypatch):
    """the vprocs > 1 code path on one GPU (staged six-stage exchange, every message through RCCL send/recv to self, device all-reduces)
    with a REAL interior row set: the matrix pass runs as an interior launch under the (hs,ht) halo and a boundary launch behind it"""
    s0, s1 = oracle36k
    monkeypatch.setenv("RXMD_FORCE_STAGED", "1")
    monkeypatch.setenv("RXMD_FORCE_REMOTE", "1")
    if not overlap:
        monkeypatch.setenv("RXMD_NO_HALO_OVERLAP", "1")
    if direct:                                   # RXMD_HALO_DIRECT=1: ghost values straight from their owners in one grouped exchange (engine.h)
        monkeypatch.setenv("RXMD_HALO_DIRECT", "1")
    e = _engine36k(qeq_mode)
    e.init_rccl(e.rccl_unique_id(), 0, 1)
    it, est = e.QEq(); pe = e.FORCE()
    _check_step0(e, s0, it, est, pe)
    st = e.stats()
    nb = st["n_boundary_rows"]
    assert 0 < nb < 0.9 * st["natoms"], "interior rows expected: %d boundary rows of %d" % (nb, st["natoms"])
    # the matrix passes were window passes (groups of 16 cell-sorted rows; with overlap: interior groups under the halo, boundary groups behind it)
    assert st["win_in_use"] == 1 and (st["natoms"] + 15) // 16 <= st["win_groups"] <= st["natoms"] // 16 + st["cells10"][0] * st["cells10"][1] + 1, st
    e.step(NSTEPS)
    _check_trajectory(e, s1, s0)
    assert e.stats()["win_in_use"] == 1
    e.close()


def test_forces_of_the_979776_atom_crystal_by_periodicity():
    """BASELINE configs[1] at full size: per-atom forces and charges of all 16^3 interior unit cells of RDX 18 x 18 x 18 against an interior
    cell of a 4 x 4 x 4 ORACLE run.  An interior cell has every bonded partner among the residents, in the same relative index order
    (cells are numbered x-major in both systems), so the `nbr < i` rule of ForceBondedTerms selects the same terms."""
    from test_gpu_parity import _engine, _oracle
    kw = dict(QEq_tol=1e-12, NMAXQEq=2000)
    o = _oracle("rdx168", (4, 4, 4), **kw); o.qeq(); o.force()
    qo = o.charges().reshape(4, 4, 4, 168); fo = o.forces().reshape(4, 4, 4, 168, 3)
    assert np.array_equal(o.gids(), np.arange(1, 4 ** 3 * 168 + 1))
    # the oracle's own interior cells agree with each other (what the comparison rests on)
    for c in [(1, 1, 2), (1, 2, 1), (2, 1, 1), (2, 2, 2)]:
        assert np.abs(fo[c] - fo[1, 1, 1]).max() <= 1e-7 and np.abs(qo[c] - qo[1, 1, 1]).max() <= 1e-9      # forces follow the charges' 1e-10 CG noise
    mc = (18, 18, 18)
    e = _engine("rdx168", mc, qeq_mode=1, **kw)
    e.QEq(); e.FORCE(); a = e.atoms()
    assert np.array_equal(a["gid"], np.arange(1, 18 ** 3 * 168 + 1))
    inner = lambda x, w: x.reshape((18, 18, 18, 168) + w)[1:17, 1:17, 1:17].reshape((-1,) + w)
    nc = 16 ** 3
    qref = np.broadcast_to(qo[1, 1, 1], (nc, 168)).reshape(-1); fref = np.broadcast_to(fo[1, 1, 1], (nc, 168, 3)).reshape(-1, 3)
    assert q_err(inner(a["q"], ()), qref) <= QTOL
    # with its own charges: two CG runs that each stop on a 1e-12 relative change of Est agree to ~1e-7 relative in the charges (measured
    # 1e-7; the exit iteration moves with the summation order), and 1e-8 e x ~1e2 kcal/mol/A/e is 1e-6 of an rms force of 1.2
    assert f_err(inner(a["f"], (3,)), fref) <= 5e-6
    # the force kernels alone (protocol step 2): the oracle's charges, which by periodicity are those of EVERY cell of the crystal,
    # injected into the 979,776-atom engine -> the parity tolerance
    e.set_charges(np.tile(qo[1, 1, 1], 18 ** 3))
    e.FORCE(); a = e.atoms(); e.close()
    assert f_err(inner(a["f"], (3,)), fref) <= FTOL


@pytest.mark.parametrize("vp,direct", [((2, 1, 1), False), ((2, 1, 1), True), ((2, 2, 2), False), ((2, 2, 2), True)])
def test_perturbed_rdx_36k_on_several_ranks_against_the_multi_rank_oracle(vp, direct, monkeypatch):
    """the 36,288-atom perturbed crystal as a 2 x 1 x 1 and as a 2 x 2 x 2 decomposition: engine ranks (one GPU, messages host-staged over
    gloo) against the oracle run with the same vprocs -- per-rank local order after three migrations, charges, forces, positions.  Every
    domain has interior rows (the two-launch matrix pass under the halo) and boundary rows; on 2 x 2 x 2 (domains of 39 x 35 x 32 A) ghosts
    reach a rank through the edge / corner forwarding of comm.F90:68-86 (x, then y carrying x's ghosts, then z carrying both) while interior
    groups run under the halo.  `direct`: the owner-to-ghost vector halo (7 peers per rank on 2 x 2 x 2)."""
    import socket
    import torch.multiprocessing as mp
    import mr_worker
    world = vp[0] * vp[1] * vp[2]
    if direct:
        monkeypatch.setenv("RXMD_HALO_DIRECT", "1")
    if world > 2:                                 # 8 processes on ONE GPU: a second hardware queue each would be time-sliced
        monkeypatch.setenv("RXMD_SINGLE_STREAM", "1")
    ff, lat2, ranks, vs = _perturbed_rdx(vp)
    nmax = max(len(r["type"]) for r in ranks)
    o = oa.Oracle(ff, lat2, ranks, vprocs=vp, nbuffer=(10 if world == 2 else 12) * nmax, v0=vs, **KW)
    o.qeq(); o.force(); o.step(NSTEPS)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as m:
        out = m.dict()
        ps = [ctx.Process(target=mr_worker.engine_rank_perturbed, args=(r, world, port, vp, NSTEPS, 1, out)) for r in range(world)]
        [p.start() for p in ps]; [p.join(1200) for p in ps]
        for p in ps:
            if p.is_alive():
                p.kill()
        assert len(out) == world, "a rank died or hung"
        res = [out[r] for r in range(world)]
    moved = 0
    for r, x in enumerate(res):
        assert "error" not in x, x.get("error")
        assert x["err"] == "None"
        assert 0 < x["n_boundary_rows"] < x["natoms"]
        assert np.array_equal(x["gid"], o.gids(r))
        moved += int(not np.array_equal(np.sort(x["gid"]), np.sort(ranks[r]["gid"])))
        assert np.abs(x["pos"] - o.pos(r)).max() <= 1e-9
        assert q_err(x["q"], o.charges(r)) <= QTOL
        assert f_err(x["f"], o.forces(r)) <= FTOL
    assert moved > 0, "the test needs atoms that changed their rank"


@pytest.mark.parametrize("case,mc_small,inner,mc_full,natoms", [("ice644", (6, 4, 4), (2, 1, 1), (60, 35, 40), 2016000), ("sicnp", (3, 3, 3), (1, 1, 1), (12, 12, 12), 945216)])
def test_forces_of_the_full_size_water_and_sicnp_crystals_by_periodicity(case, mc_small, inner, mc_full, natoms):
    """BASELINE configs[2] (perturbed ice Ih 60 x 35 x 40 = 2,016,000 atoms) and configs[4] (SiC nanoparticle + O2 with PQEq, 12^3 = 945,216 atoms) at
    full size, PER-ATOM FORCES: every interior unit cell of the full replication against an interior cell of a small ORACLE run of the same cell
    (the argument of test_forces_of_the_979776_atom_crystal_by_periodicity: an interior cell has all its bonded partners among the residents in the
    same relative index order -- geninit numbers cells x-major in both systems -- so the `nbr < i` rule of ForceBondedTerms, pot.F90:113-144, and the
    gid-ordered torsion booking select the same terms; energies cannot see that rule, forces do).  With the engine's own charges the forces follow the
    charges' CG truncation; with the oracle's charges injected (those of EVERY cell by periodicity; PQEq: shells on their cores at step 0,
    pqeq.F90:361-435) the parity tolerance 1e-6."""
    from test_gpu_parity import _engine, _oracle
    kw = dict(QEq_tol=1e-12, NMAXQEq=2000)
    if case == "sicnp":
        kw["pqeq"] = oa.PQEQ_SICNP
    o = _oracle(case, mc_small, **kw)
    if case == "sicnp":
        o.set_pqeq_clean(1)
    o.qeq(); o.force()
    ns = mc_small[0] * mc_small[1] * mc_small[2]
    n0 = len(o.gids()) // ns
    assert n0 * mc_full[0] * mc_full[1] * mc_full[2] == natoms
    assert np.array_equal(o.gids(), np.arange(1, ns * n0 + 1))
    qo = o.charges().reshape(mc_small + (n0,)); fo = o.forces().reshape(mc_small + (n0, 3))
    frms = np.sqrt((fo ** 2).mean())
    # what the comparison rests on: the oracle's own interior cells agree with each other
    others = [c for c in [(inner[0] + 1, inner[1], inner[2]), (inner[0], inner[1] + 1, inner[2]), (inner[0], inner[1], inner[2] + 1)]
              if all(c[a] < mc_small[a] for a in range(3))]           # (3 x 3 x 3 has ONE interior cell: its face neighbours then -- these two cells show no order effect)
    assert others
    for c in others:
        assert np.abs(fo[c] - fo[inner]).max() <= 2e-7 * max(frms, 1.0) and np.abs(qo[c] - qo[inner]).max() <= 1e-9
    e = _engine(case, mc_full, qeq_mode=1, **kw)
    e.QEq(); e.FORCE(); a = e.atoms()
    assert np.array_equal(a["gid"], np.arange(1, natoms + 1))
    sl = tuple(slice(1, m - 1) for m in mc_full)
    inner_of = lambda x, w: x.reshape(mc_full + (n0,) + w)[sl].reshape((-1,) + w)
    nc = (mc_full[0] - 2) * (mc_full[1] - 2) * (mc_full[2] - 2)
    qref = np.broadcast_to(qo[inner], (nc, n0)).reshape(-1); fref = np.broadcast_to(fo[inner], (nc, n0, 3)).reshape(-1, 3)
    assert q_err(inner_of(a["q"], ()), qref) <= QTOL
    assert f_err(inner_of(a["f"], (3,)), fref) <= 5e-6              # its own charges: two CG runs that stop on a 1e-12 change of Est agree to ~1e-7 in q
    e.set_charges(np.tile(qo[inner], natoms // n0))
    e.FORCE(); a = e.atoms(); e.close()
    assert f_err(inner_of(a["f"], (3,)), fref) <= FTOL


def test_without_the_window_form_every_consumer_falls_back(oracle36k, monkeypatch):
    """RXMD_SPMV_NO_WIN=1: what a list build whose windows do not fit does (more than 448 units in a group) -- the QEq passes run as row passes
    (k_spmv) and ENbond as one wavefront per row with its 32-byte gather (k_nonbond) from the 4-byte entries.  Same step-0 state and the same
    three MD steps as the default against the oracle."""
    s0, s1 = oracle36k
    monkeypatch.setenv("RXMD_SPMV_NO_WIN", "1")
    e = _engine36k(1)
    it, est = e.QEq(); pe = e.FORCE()
    assert e.stats()["win_in_use"] == 0
    _check_step0(e, s0, it, est, pe)
    e.step(NSTEPS)
    _check_trajectory(e, s1, s0)
    e.close()
