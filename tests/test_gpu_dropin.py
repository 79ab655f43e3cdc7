"""Drop-in level 1 of INTEGRATION.md, end to end: the REFERENCE'S OWN driver (src/main.F90 and every other object of the serial
build, unmodified) with its QEq / FORCE calls redirected through bindings/rxmd_hip_mod.F90 to librxmd_hip.so.  oracle/Makefile
builds it as oracle/_ref/rxmd_hipdrv in the container that has the reference sources; here it runs on the GPU box like the
reference would: geninit -> rxmd.in / ffield / DAT/rxff.bin -> MD loop -> MDstep lines and the xyz frame."""
import os, shutil, subprocess, tempfile
import numpy as np
import pytest
import oracle_api as oa

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")


def test_reference_driver_runs_on_the_hip_library():
    drv, gen = os.path.join(REF, "rxmd_hipdrv"), os.path.join(REF, "geninit")
    if not (os.path.exists(drv) and os.path.exists(gen)):
        pytest.skip("oracle/_ref/rxmd_hipdrv was not built (needs the reference sources + amdflang: make -C oracle ref)")
    g = np.load(os.path.join(oa.GOLD, "rdx168_md10.npz"))
    tmp = tempfile.mkdtemp(prefix="dropin_")
    try:
        os.makedirs(os.path.join(tmp, "DAT"))
        shutil.copy(os.path.join(oa.INP, "rdx.xyz"), os.path.join(tmp, "input.xyz"))
        shutil.copy(os.path.join(oa.INP, "ffield_rdx"), os.path.join(tmp, "ffield"))
        shutil.copy(os.path.join(oa.INP, "rxmd.in"), os.path.join(tmp, "rxmd.in"))
        subprocess.run([gen, "-i", "input.xyz", "-f", "ffield", "-o", "DAT", "-mc", "1", "1", "1"], cwd=tmp, check=True, stdout=subprocess.DEVNULL)
        p = subprocess.run([drv, "--ntime_step", "10", "--pstep", "1", "--fstep", "10"], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert "successfully finished" in p.stdout, p.stdout[-3000:]
        rows = [[float(x) for x in l.split()[1:13]] for l in p.stdout.split("\n") if l.startswith("MDstep:")]
        rows = np.array(rows); ref = g["mdstep"][:len(rows), :12]
        assert len(rows) == 10
        # MDstep: step, total / potential / kinetic energy per atom (es13.5), six energy groups (es11.3), temperature, pressure (f8.2)
        assert np.allclose(rows[:, 1:3], ref[:, 1:3], rtol=2e-6)
        assert np.allclose(rows[:, 3], ref[:, 3], rtol=2e-3, atol=1e-9)            # kinetic energy: 6 digits of a 1e-4 number
        assert np.allclose(rows[:, 4:10], ref[:, 4:10], rtol=2e-3, atol=1e-6)
        assert np.allclose(rows[:, 11], ref[:, 11], atol=0.011)                     # pressure [GPa]: the virial came from the engine
        # the trajectory frame the driver wrote after 10 steps against the one the unmodified reference wrote
        mine = open(os.path.join(tmp, "DAT", "000000010.xyz")).read().split("\n")
        theirs = str(g["xyz_last"]).split("\n")
        assert mine[:2] == theirs[:2] and len(mine) == len(theirs)
        for a, b in zip(mine[2:-1], theirs[2:-1]):
            assert a[:3] == b[:3] and a[47:] == b[47:]                              # element, global id
            assert np.allclose([float(x) for x in a[3:39].split()], [float(x) for x in b[3:39].split()], atol=2.1e-5, rtol=0)   # positions, f12.5
            assert abs(float(a[39:47]) - float(b[39:47])) <= 1.1e-3                                                              # charge, f8.3
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def test_reference_driver_extended_lagrangian_charges_on_the_hip_library():
    """--isQEq 2 through the drop-in driver: the fictitious charges qsfp/qsfv live in the reference's module atoms, are integrated by
    its main.F90 (:67-68,98) and must travel into and out of every QEq call (rxmd_hip_put_lex / rxmd_hip_get_lex in QEq_hip);
    10 steps against what the unmodified reference prints for the same run"""
    drv, gen = os.path.join(REF, "rxmd_hipdrv"), os.path.join(REF, "geninit")
    if not (os.path.exists(drv) and os.path.exists(gen)):
        pytest.skip("oracle/_ref/rxmd_hipdrv was not built (needs the reference sources + amdflang: make -C oracle ref)")
    g = np.load(os.path.join(oa.GOLD, "rdx168_lex_md10.npz"))
    tmp = tempfile.mkdtemp(prefix="dropin_")
    try:
        os.makedirs(os.path.join(tmp, "DAT"))
        shutil.copy(os.path.join(oa.INP, "rdx.xyz"), os.path.join(tmp, "input.xyz"))
        shutil.copy(os.path.join(oa.INP, "ffield_rdx"), os.path.join(tmp, "ffield"))
        shutil.copy(os.path.join(oa.INP, "rxmd.in"), os.path.join(tmp, "rxmd.in"))
        subprocess.run([gen, "-i", "input.xyz", "-f", "ffield", "-o", "DAT", "-mc", "1", "1", "1"], cwd=tmp, check=True, stdout=subprocess.DEVNULL)
        p = subprocess.run([drv, "--ntime_step", "10", "--pstep", "1", "--fstep", "10", "--isQEq", "2"], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert "successfully finished" in p.stdout, p.stdout[-3000:]
        rows = np.array([[float(x) for x in l.split()[1:13]] for l in p.stdout.split("\n") if l.startswith("MDstep:")])
        ref = g["mdstep"][:len(rows), :12]
        assert len(rows) == 10
        assert np.allclose(rows[:, 1:3], ref[:, 1:3], rtol=2e-6)                    # total and potential energy per atom, es13.5
        assert np.allclose(rows[:, 4:10], ref[:, 4:10], rtol=2e-3, atol=1e-6)       # the six energy groups, es11.3
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def test_reference_driver_with_pqeq_on_the_hip_library():
    """the same with --pqeq: PQEq(atype,pos,q) -> PQEq_hip, FORCE -> FORCE_hip carrying module atoms' spos in and out"""
    drv, gen = os.path.join(REF, "rxmd_hipdrv"), os.path.join(REF, "geninit")
    if not (os.path.exists(drv) and os.path.exists(gen)):
        pytest.skip("oracle/_ref/rxmd_hipdrv was not built (needs the reference sources + amdflang: make -C oracle ref)")
    g = np.load(os.path.join(oa.GOLD, "sicnp547_pqeq_md5.npz"))
    tmp = tempfile.mkdtemp(prefix="dropin_")
    try:
        os.makedirs(os.path.join(tmp, "DAT"))
        shutil.copy(os.path.join(oa.INP, "sicnp.xyz"), os.path.join(tmp, "input.xyz"))
        shutil.copy(os.path.join(oa.INP, "ffield_sicnp"), os.path.join(tmp, "ffield"))
        shutil.copy(os.path.join(oa.INP, "rxmd.in"), os.path.join(tmp, "rxmd.in"))
        shutil.copy(oa.PQEQ_SICNP, os.path.join(tmp, "pqeq.in"))
        subprocess.run([gen, "-i", "input.xyz", "-f", "ffield", "-o", "DAT", "-mc", "1", "1", "1"], cwd=tmp, check=True, stdout=subprocess.DEVNULL)
        p = subprocess.run([drv, "--ntime_step", "5", "--pstep", "1", "--fstep", "5", "--pqeq", "pqeq.in"], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           text=True, timeout=600)
        assert "successfully finished" in p.stdout, p.stdout[-3000:]
        rows = np.array([[float(x) for x in l.split()[1:13]] for l in p.stdout.split("\n") if l.startswith("MDstep:")])
        ref = g["mdstep"][:len(rows), :12]
        assert len(rows) == 5
        assert np.allclose(rows[0, 1:3], ref[0, 1:3], rtol=2e-6)            # step 0: shells on the cores, no beyond-cutoff lookups
        assert np.allclose(rows[:, 1:3], ref[:, 1:3], rtol=2e-5)            # later steps: the reference's stale-value artefact is inside this
        assert np.allclose(rows[:, 4:10], ref[:, 4:10], rtol=5e-3, atol=1e-5)
        mine = open(os.path.join(tmp, "DAT", "000000005.xyz")).read().split("\n")
        theirs = str(g["xyz_last"]).split("\n")
        assert mine[:2] == theirs[:2] and len(mine) == len(theirs)
        for a, b in zip(mine[2:-1], theirs[2:-1]):
            assert len(a) == len(b) and a[:3] == b[:3] and a[83:92] == b[83:92]
            va = np.array([float(t) for t in a[3:83].split()] + [float(t) for t in a[92:].split()])
            vb = np.array([float(t) for t in b[3:83].split()] + [float(t) for t in b[92:].split()])
            assert np.abs(va[:3] - vb[:3]).max() <= 1e-5                    # positions after 5 steps at the run tolerance 1e-7
            assert abs(va[3] - vb[3]) <= 1e-2 and np.abs(va[4:] - vb[4:]).max() <= 1e-3   # charge, shell displacement (artefact-sized bounds)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def test_reference_minimiser_runs_on_the_hip_library():
    """mdmode 10: the reference's own geometry minimiser (src/cg.F90, unmodified algorithm: bracket, golden section, Polak-Ribiere)
    with its QEq / FORCE calls redirected to the HIP library; the minimised structure must be the one the reference finds"""
    import resource
    drv, gen = os.path.join(REF, "rxmd_hipdrv"), os.path.join(REF, "geninit")
    if not (os.path.exists(drv) and os.path.exists(gen)):
        pytest.skip("oracle/_ref/rxmd_hipdrv was not built (needs the reference sources + amdflang: make -C oracle ref)")
    g = np.load(os.path.join(oa.GOLD, "rdx168_minimiser.npz"))
    tmp = tempfile.mkdtemp(prefix="dropin_")
    try:
        os.makedirs(os.path.join(tmp, "DAT"))
        shutil.copy(os.path.join(oa.INP, "rdx.xyz"), os.path.join(tmp, "input.xyz"))
        shutil.copy(os.path.join(oa.INP, "ffield_rdx"), os.path.join(tmp, "ffield"))
        shutil.copy(os.path.join(oa.INP, "rxmd.in"), os.path.join(tmp, "rxmd.in"))
        with open(os.path.join(tmp, "rxmd.in"), "a") as f:
            f.write("CG_tol 1.d-5\n")
        subprocess.run([gen, "-i", "input.xyz", "-f", "ffield", "-o", "DAT", "-mc", "1", "1", "1"], cwd=tmp, check=True, stdout=subprocess.DEVNULL)
        try:                                  # cg.F90 keeps NBUFFER-sized automatic arrays on the stack; raised in THIS process (inherited), no preexec_fn
            resource.setrlimit(resource.RLIMIT_STACK, (resource.RLIM_INFINITY, resource.RLIM_INFINITY))
        except (ValueError, OSError):
            soft, hard = resource.getrlimit(resource.RLIMIT_STACK)
            resource.setrlimit(resource.RLIMIT_STACK, (hard, hard))
        p = subprocess.run([drv, "--mdmode", "10", "--QEq_tol", "1e-12", "--NMAXQEq", "2000"], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                           timeout=900)
        assert "successfully finished structural optimization" in p.stdout, p.stdout[-3000:]
        mine = open(os.path.join(tmp, "DAT", "000000000.xyz")).read().split("\n")
        theirs = str(g["xyz"]).split("\n")
        assert mine[:2] == theirs[:2] and len(mine) == len(theirs)
        for a, b in zip(mine[2:-1], theirs[2:-1]):
            assert a[:3] == b[:3] and a[47:] == b[47:]
            # the golden-section search stops at a relative step tolerance of 1e-6 (cg.F90:16): the two minima agree to ~1e-4 A
            assert np.allclose([float(x) for x in a[3:39].split()], [float(x) for x in b[3:39].split()], atol=3e-4, rtol=0)
            assert abs(float(a[39:47]) - float(b[39:47])) <= 1.1e-3
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


@pytest.mark.parametrize("ex,extra", [("example1", []), ("example3", ["pqeq1.par"])])
def test_reference_example_workflows_on_the_hip_library(ex, extra):
    """The reference's own examples as shipped (examples/1-reaxff; examples/3-reaxpq+ = PQEq + electric field along x): polyethylene,
    `geninit -mc 2 3 5` (a 14.8 x 14.8 x 12.7 A box, smaller than twice the cut-off), their rxmd.in -- run by the reference's driver on the
    HIP library, against what the unmodified reference prints and writes for the same 20 steps."""
    drv, gen = os.path.join(REF, "rxmd_hipdrv"), os.path.join(REF, "geninit")
    if not (os.path.exists(drv) and os.path.exists(gen)):
        pytest.skip("oracle/_ref/rxmd_hipdrv was not built (needs the reference sources + amdflang: make -C oracle ref)")
    g = np.load(os.path.join(oa.GOLD, ex + ".npz"))
    tmp = tempfile.mkdtemp(prefix="dropin_")
    try:
        os.makedirs(os.path.join(tmp, "DAT"))
        shutil.copy(os.path.join(oa.INP, "example1", "pe_cell.xyz"), os.path.join(tmp, "input.xyz"))
        shutil.copy(os.path.join(oa.INP, "example1", "ffield_pe"), os.path.join(tmp, "ffield"))
        shutil.copy(os.path.join(oa.INP, ex, "rxmd_%s.in" % ex), os.path.join(tmp, "rxmd.in"))
        for f in extra:                                       # the rxmd.in of example 3 names ./pqeq1.par
            shutil.copy(os.path.join(oa.INP, ex, f.replace(".par", "_%s.par" % ex)), os.path.join(tmp, f))
        subprocess.run([gen, "-i", "input.xyz", "-f", "ffield", "-o", "DAT", "-mc", "2", "3", "5"], cwd=tmp, check=True, stdout=subprocess.DEVNULL)
        p = subprocess.run([drv, "--ntime_step", "20", "--pstep", "5", "--fstep", "20"], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert "successfully finished" in p.stdout, p.stdout[-3000:]
        rows = np.array([[float(x) for x in l.split()[1:13]] for l in p.stdout.split("\n") if l.startswith("MDstep:")])
        ref = g["mdstep"][:len(rows), :12]
        assert len(rows) == 4 and list(rows[:, 0]) == [0.0, 5.0, 10.0, 15.0]
        assert np.allclose(rows[0, 1:3], ref[0, 1:3], rtol=2e-6)                     # step 0 to the printed digits
        assert np.allclose(rows[:, 1:3], ref[:, 1:3], rtol=2e-5)                     # the run tolerance (QEq 1e-7) over 15 steps
        assert np.allclose(rows[:, 4:10], ref[:, 4:10], rtol=5e-3, atol=1e-5)
        mine = open(os.path.join(tmp, "DAT", "000000020.xyz")).read().split("\n")
        theirs = str(g["xyz_last"]).split("\n")
        assert mine[:2] == theirs[:2] and len(mine) == len(theirs)
        dq, ds = [], []
        for a, b in zip(mine[2:-1], theirs[2:-1]):
            assert a[:3] == b[:3] and len(a) == len(b)
            if extra:      # PQEq frame (fileio.F90:241-355): position, charge in es formats up to column 83, id, shell displacement behind column 92
                va = np.array([float(t) for t in a[3:83].split()] + [float(t) for t in a[92:].split()])
                vb = np.array([float(t) for t in b[3:83].split()] + [float(t) for t in b[92:].split()])
                assert np.abs(va[:3] - vb[:3]).max() <= 5e-5
                dq.append(abs(va[3] - vb[3])); ds.append(np.abs(va[4:] - vb[4:]).max())
            else:
                assert np.allclose([float(x) for x in a[3:39].split()], [float(x) for x in b[3:39].split()], atol=5e-5, rtol=0)   # positions, f12.5
                assert abs(float(a[39:47]) - float(b[39:47])) <= 1e-2                                                              # charge, f8.3
        if extra:
            # In this 12.7 A box the 12.5 A PQEq cut-off makes the reference re-use the previous pair's table values for many
            # core-shell lookups beyond the cut-off (module.F90:401): single atoms of its frame carry a shell pushed by the
            # 1e-3 A clip and a charge off by ~1e-2 while energies and positions agree.  Bounds are sized to that artefact.
            dq = np.sort(np.array(dq)); ds = np.sort(np.array(ds))
            print("PQEq frame vs reference: |dq| median %.2e 90%% %.2e max %.2e ; |dshell| median %.2e 90%% %.2e max %.2e"
                  % (dq[len(dq) // 2], dq[int(0.9 * len(dq))], dq[-1], ds[len(ds) // 2], ds[int(0.9 * len(ds))], ds[-1]))
            assert dq[len(dq) // 2] <= 2e-3 and dq[-1] <= 3e-2
            assert ds[len(ds) // 2] <= 5e-4 and ds[-1] <= 2.5e-3
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def test_reference_mpi_driver_two_ranks_on_the_hip_library():
    """The reference's examples/2-reaxff-dc as shipped (`geninit -mc 4 3 5 -v 2 1 1`, `mpirun -np 2 rxmd`) with the reference's own
    MPI driver on librxmd_hip.so (oracle/_ref/rxmd_hipdrv_mpi): its COPYATOMS(MODE_MOVE) migrates over MPI, every QEq / FORCE call
    goes to one engine per rank, and the engines' six-stage exchange and CG all-reduces travel through the MPI_SENDRECV /
    MPI_ALLREDUCE callbacks that bindings/rxmd_hip_mod.F90 installs (both ranks share the one GPU of this box, so the host-staged
    transport is the one exercised; with one GPU per rank the same binding hands the engines an RCCL id over MPI_BCAST).
    3 MD steps at tight tolerance against the real-MPI reference run of the same example: MDstep lines and the trajectory frame."""
    drv, gen = os.path.join(REF, "rxmd_hipdrv_mpi"), os.path.join(REF, "geninit")
    mpiexec = shutil.which("mpiexec") or "/opt/conda/bin/mpiexec"
    if not (os.path.exists(drv) and os.path.exists(gen)):
        pytest.skip("oracle/_ref/rxmd_hipdrv_mpi was not built (needs the reference sources + amdflang + MPICH: make -C oracle ref)")
    if not os.path.exists(mpiexec):
        pytest.skip("no mpiexec on this box")
    g = np.load(os.path.join(oa.GOLD, "example2_v211_md3.npz"))
    tmp = tempfile.mkdtemp(prefix="dropin_mpi_")
    try:
        os.makedirs(os.path.join(tmp, "DAT"))
        shutil.copy(os.path.join(oa.INP, "example1", "pe_cell.xyz"), os.path.join(tmp, "input.xyz"))
        shutil.copy(os.path.join(oa.INP, "example1", "ffield_pe"), os.path.join(tmp, "ffield"))
        shutil.copy(os.path.join(oa.INP, "rxmd.in"), os.path.join(tmp, "rxmd.in"))
        subprocess.run([gen, "-i", "input.xyz", "-f", "ffield", "-o", "DAT", "-mc", "4", "3", "5", "-v", "2", "1", "1"], cwd=tmp, check=True, stdout=subprocess.DEVNULL)
        env = dict(os.environ, RXMD_HIP_TRANSPORT="mpi")
        p = subprocess.run([mpiexec, "-np", "2", drv, "--ntime_step", "3", "--pstep", "1", "--fstep", "3", "--vprocs", "2", "1", "1",
                            "--QEq_tol", "1e-12", "--NMAXQEq", "2000"], cwd=tmp, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
        assert "successfully finished" in p.stdout, p.stdout[-3000:]
        assert "host-staged MPI transport over    2 ranks" in p.stdout
        rows = np.array([[float(x) for x in l.split()[1:13]] for l in p.stdout.split("\n") if l.startswith("MDstep:")])
        ref = g["mdstep"][:len(rows), :12]
        assert len(rows) == 3
        assert np.allclose(rows[:, 1:3], ref[:, 1:3], rtol=2e-6)                     # total / potential energy per atom, es13.5
        assert np.allclose(rows[:, 4:10], ref[:, 4:10], rtol=2e-3, atol=1e-6)        # the six energy groups, es11.3
        # the frame both ranks wrote after the third step against the per-rank dumps of the MPI reference
        gid = np.concatenate([g["gid_0"], g["gid_1"]]); pos = np.concatenate([g["pos_0"], g["pos_1"]]); chg = np.concatenate([g["charge_0"], g["charge_1"]])
        want = {int(i): (x, c) for i, x, c in zip(gid, pos, chg)}
        lines = open(os.path.join(tmp, "DAT", "000000003.xyz")).read().split("\n")
        assert int(lines[0].split()[0]) == len(gid)
        seen = 0
        for l in lines[2:]:
            if not l.strip():
                continue
            x = np.array([float(t) for t in l[3:39].split()]); q = float(l[39:47]); i = int(l[47:].split()[0])
            assert np.abs(x - want[i][0]).max() <= 2.1e-5 and abs(q - want[i][1]) <= 1.1e-3
            seen += 1
        assert seen == len(gid)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
