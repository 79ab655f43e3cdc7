"""Output path FROM HIP STATE: trajectory frames and checkpoints written by the engine, against the files the
unmodified reference wrote for the same runs (WriteXYZ src/fileio.F90:241-355, WriteBIN src/fileio.F90:558-653)."""
import os
import numpy as np
import pytest

import oracle_api as oa
from test_gpu_parity import _engine

pytestmark = pytest.mark.gpu


def test_xyz_frame_written_from_engine_state_is_the_references_file(tmp_path):
    """10 MD steps of RDX-168 at the default settings (QEq_tol 1e-7), then RxmdEngine.write_xyz: header lines byte for byte, every atom line
    the same element and global id at the same columns, positions to the last printed digit (f12.5), charges to the last printed
    digit (f8.3; the CG exit noise of tol 1e-7 is two orders below it) -- and nearly all lines identical as bytes."""
    g = np.load(os.path.join(oa.GOLD, "rdx168_md10.npz"))
    e = _engine("rdx168", (1, 1, 1))
    e.QEq(); e.FORCE(); e.step(10)
    path = tmp_path / "000000010.xyz"
    e.write_xyz(str(path))
    e.close()
    mine = open(path).read().split("\n"); theirs = str(g["xyz_last"]).split("\n")
    assert mine[:2] == theirs[:2] and len(mine) == len(theirs) and mine[-1] == theirs[-1] == ""
    same = 0
    for a, b in zip(mine[2:-1], theirs[2:-1]):
        assert len(a) == len(b) == 56 and a[:3] == b[:3] and a[47:] == b[47:]
        assert np.abs(np.array([float(x) for x in a[3:39].split()]) - np.array([float(x) for x in b[3:39].split()])).max() <= 1.01e-5
        assert abs(float(a[39:47]) - float(b[39:47])) <= 1.01e-3
        same += a == b
    assert same >= 0.97 * (len(mine) - 3), same


def test_bond_file_and_pdb_frame_written_from_engine_state_are_the_references_files(tmp_path):
    """The same run as above with the reference's --isBondFile --isPDB: RxmdEngine.write_bnd (WriteBND, fileio.F90:27-148: the input of
    rxmd's util/ tools) and write_pdb (WritePDB, fileio.F90:151-238).  Bond file: every line the same global id, type, number of listed
    bonds and the same (partner id, bond order f6.3) pairs -- in the order of the engine's bond list, which walks the cells in another
    order than the reference's linked lists (a permutation inside the line) -- positions to f12.3; pdb: the fixed columns byte for byte, positions to f8.3, the charge column to f6.2, the stress column as printed."""
    g = np.load(os.path.join(oa.GOLD, "rdx168_md10.npz"))
    e = _engine("rdx168", (1, 1, 1))
    e.QEq(); e.FORCE(); e.step(9)
    e.energy()                                   # PRINTE every step (pstep 1): the stress accumulators the last frame sees are those of the last step
    e.step(1)
    pb, pp = tmp_path / "000000010.bnd", tmp_path / "000000010.pdb"
    e.write_bnd(str(pb)); e.write_pdb(str(pp))
    e.close()
    mine = open(pb).read().split("\n"); theirs = str(g["bnd_last"]).split("\n")
    assert len(mine) == len(theirs) == 169 and mine[-1] == theirs[-1] == ""
    same = 0
    for a, b in zip(mine[:-1], theirs[:-1]):
        ta, tb = a.split(), b.split()
        assert len(ta) == len(tb) and ta[0] == tb[0] and ta[4:6] == tb[4:6] and sorted(ta[6::2]) == sorted(tb[6::2]), (a, b)
        assert np.abs(np.array([float(v) for v in ta[1:4]]) - np.array([float(v) for v in tb[1:4]])).max() <= 1.01e-3
        pa = sorted(zip(ta[6::2], ta[7::2])); pb_ = sorted(zip(tb[6::2], tb[7::2]))
        assert max([abs(float(u[1]) - float(v[1])) for u, v in zip(pa, pb_)] + [0.0]) <= 1.01e-3
        same += a[:58] == b[:58] and pa == pb_
    assert same >= 0.97 * 168, same
    mine = open(pp).read().split("\n"); theirs = str(g["pdb_last"]).split("\n")
    assert len(mine) == len(theirs) == 169
    same = 0
    for a, b in zip(mine[:-1], theirs[:-1]):
        assert len(a) == len(b) == 66 and a[:30] == b[:30] and a[60:] == b[60:], (a, b)
        assert np.abs(np.array([float(a[30 + 8 * c:38 + 8 * c]) - float(b[30 + 8 * c:38 + 8 * c]) for c in range(3)])).max() <= 1.01e-3
        assert abs(float(a[54:60]) - float(b[54:60])) <= 1.01e-2
        same += a == b
    assert same >= 0.97 * 168, same


@pytest.mark.parametrize("win", ["1", "0"])
@pytest.mark.parametrize("mode,kw", [(4, dict(vsfact=0.9)), (5, dict(treq=300.0)), (7, dict(treq=300.0)), (8, dict(treq=300.0))])
def test_checkpoint_written_from_engine_state_is_the_references_file(mode, kw, win, tmp_path, monkeypatch):
    """Continue the reference's own restart file (rxff.bin after 20 NVE steps) for 7 steps with velocity scaling every 3rd step, then
    RxmdEngine.write_rxff: the header as bytes (process grid, atom count, step counter 27, lattice), and every record field against
    the rxff.bin the reference wrote at the end of the same run -- normalised positions, velocities, charge, packed type+id, and the
    fictitious charges qsfp / qsfv that main.F90:67-68,98 integrates in every mode."""
    import rxmd_amd
    monkeypatch.setenv("RXMD_SPMV_WIN", win)       # "0": the wavefront-per-row pass keeps the velocity gate it had before the window pass (1e-8)
    g = np.load(os.path.join(oa.GOLD, "rdx168_thermo%d.npz" % mode))
    ff = oa.make_system("rdx168")[0]
    lat, vp, step0, recs = oa.parse_rxff(g["restart_rxff"])
    e = rxmd_amd.RxmdEngine(ff, lat, QEq_tol=1e-12, NMAXQEq=2000)
    e.set_atoms_rxff(recs[0])
    e.QEq(); e.FORCE()
    for nstep in range(7):
        if nstep % 3 == 0:
            e.thermostat(mode, **kw)
        e.step(1)
    path = tmp_path / "rxff.bin"
    e.write_rxff(str(path), current_step=step0 + 7)
    e.close()
    mine = open(path, "rb").read(); theirs = bytes(g["final_rxff"])
    nhead = 4 * (4 + 1 + 1) + 48
    assert len(mine) == len(theirs) and mine[:nhead] == theirs[:nhead]
    _, _, s1, r1 = oa.parse_rxff(np.frombuffer(mine, np.uint8)); _, _, s2, r2 = oa.parse_rxff(g["final_rxff"])
    assert s1 == s2 == 27
    a, b = r1[0], r2[0]
    ta, tb = np.rint(a[:, 7]), np.rint(b[:, 7])                               # atype = type + gid * 1e-13 (main.F90:582-593): same local order
    assert np.array_equal(ta, tb) and np.array_equal(np.rint((a[:, 7] - ta) * 1e13), np.rint((b[:, 7] - tb) * 1e13))
    assert np.abs(a[:, 7] - b[:, 7]).max() <= 4e-16                           # and the packed value itself to the last bit or two
    assert np.abs(a[:, 0:3] - b[:, 0:3]).max() <= 1e-10                       # normalised positions (1e-9 A)
    # (velocities carry the charges' path dependence: a CG that stops one of seven steps at 54 instead of ~70 iterations leaves 1e-9 in q and
    #  1.4e-8 here; the row pass and the window pass stop at different counts, the reference's own spread under re-ordering is of this size)
    assert np.abs(a[:, 3:6] - b[:, 3:6]).max() <= (1e-8 if win == "0" else 5e-8) * np.abs(b[:, 3:6]).max()
    qrms = np.sqrt((b[:, 6] ** 2).mean())
    assert (np.abs(a[:, 6] - b[:, 6]) / np.maximum(np.abs(b[:, 6]), qrms)).max() <= 1e-6
    assert (np.abs(a[:, 8] - b[:, 8]) / np.maximum(np.abs(b[:, 8]), qrms)).max() <= 1e-6
    # qsfv += 0.5 dt Lex_w2 (q - qsfp) with 0.5 dt Lex_w2 = Lex_k / dt = 391 in the reference's time unit: the truncation noise of the charges (two CG
    # runs that each stop on a 1e-12 relative change of Est differ by 1e-9 .. 3e-9 in q, whichever pass: measured 1.8e-7 with the window pass,
    # 7.0e-7 with the row pass in mode 4) times 391 -- an absolute gate at that size, not one relative to the largest qsfv of the frame
    assert np.abs(a[:, 9] - b[:, 9]).max() <= 391.0 * 3e-9


def test_device_resident_minimiser_against_the_references_line_search_and_the_oracle():
    """mdmode 10 on the device-resident primitive (rxmd_hip_minimise = ConjugateGradient of src/cg.F90 restated over HBM-resident positions,
    directions and gradients; every trial point is migrate + QEq + FORCE on the device).
    What the reference itself pins: its first line search.  It prints the bracket (cg.F90:127-129) and every golden-section round
    (cg.F90:262-263); the engine's first CG loop must end on the reference's final step and on the energy it printed there.
    Beyond that loop the reference is no yardstick: LineMinimization hands MigrateVec3D the SEARCH DIRECTION as the vector to migrate,
    together with an uninitialised atype array (cg.F90:226, 299-310), so the compaction of COPYATOMS(MODE_MOVE) (comm.F90:238-252) destroys
    the direction, the atoms hardly move and the energy criterion is met at loop 0 whatever CG_tol is (measured here: |dE| = 3e-7 kcal/mol
    with CG_tol 1e-5 and 1e-6 alike).  The engine implements the algorithm the source describes: the full run must go downhill loop
    after loop and end on a structure whose energy the ORACLE confirms."""
    import rxmd_amd
    g = np.load(os.path.join(oa.GOLD, "rdx168_minimiser.npz"))
    kw = dict(QEq_tol=1e-12, NMAXQEq=2000)
    e = _engine("rdx168", (1, 1, 1), **kw)
    e.QEq(); pe0 = e.FORCE()[0]
    a0 = e.atoms(); f0 = a0["f"].copy(); x0 = a0["pos"].copy()
    assert abs(float(g["bracket"][0]) - 2e-2 / 168) <= 1e-9                  # the reference's bracket: the very first trial step (printed es15.5)
    loops, pe1, ev1 = e.minimise(ftol=1e-5, max_loops=1)
    gs = g["golden_section"]
    assert loops == 1 and ev1 >= 2 * len(gs)
    assert abs(pe1 - gs[-1, 4]) <= 2e-6                                      # energy at the end of the line search = the reference's last printed PEbx
    a1 = e.atoms()
    lat = np.array(e.lattice[:3])
    o1 = np.argsort(a1["gid"]); o0 = np.argsort(a0["gid"])                   # the migration appends an atom that crossed a box face at the end
    d = a1["pos"][o1] - (x0 + gs[-1, 3] * f0)[o0]; d -= lat * np.rint(d / lat)   # the atoms moved by (final dx) * f; modulo the wrap
    assert np.abs(d).max() <= 5e-7                                           # dx itself is decided by 1e-10 energy differences in the last rounds
    e.close()
    # the full minimisation
    e = _engine("rdx168", (1, 1, 1), **kw)
    e.QEq(); e.FORCE()
    loops, pe, ev = e.minimise(ftol=1e-5)
    assert 2 <= loops < 500 and pe < pe1 - 0.1 and pe1 < pe0                # it keeps going downhill where the reference stops
    a = e.atoms()
    assert np.abs(a["v"]).max() == 0.0
    ff = oa.make_system("rdx168")[0]
    rec = e.get_atoms_rxff()
    ranks = [dict(rnorm=rec[:, 0:3].copy(), type=np.rint(rec[:, 7]).astype(np.int32), gid=a["gid"].copy())]
    o = oa.Oracle(ff, e.lattice, ranks, **kw); o.qeq(); o.force()
    assert abs(o.energy()[0] - pe) <= 1e-9 * abs(pe)
    fo = o.forces(); frms = np.sqrt((fo ** 2).mean())
    # two CG charge solves that each stop on a 1e-12 relative energy change agree to ~1e-7 in the charges; on the small forces of a
    # minimised structure (rms 0.55 kcal/mol/A) that is 4e-7 absolute = 1.4e-6 of the rms (measured)
    assert (np.abs(a["f"] - fo).max(axis=1) / np.maximum(np.abs(fo).max(axis=1), frms)).max() <= 5e-6
    e.close()
