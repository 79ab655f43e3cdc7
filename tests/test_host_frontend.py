"""CPU-only tests: the C-ABI library loads and exports every symbol include/rxmd_hip.h declares, the
host front-end (ffield parser, derived tables, geninit, rxff.bin, rxmd.in) agrees with the oracle and
with files produced by the real reference, and the product fails loudly without a GPU."""
import ctypes as C
import os
import re
import numpy as np
import pytest

import oracle_api as oa
import rxmd_amd
from rxmd_amd import system, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FF_RDX = os.path.join(oa.INP, "ffield_rdx")
FF_WAT = os.path.join(oa.INP, "ffield_water")


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "rxmd_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(rxmd_(?:hip|host)_\w+)\s*\(", hdr))
    assert len(declared) >= 24
    lib = C.CDLL(rxmd_amd.SO_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), "librxmd_hip.so does not export %s" % name
    bound = {s[0] for s in _lib.SYMBOLS}
    assert declared == bound, (declared ^ bound)
    assert rxmd_amd.load_library().rxmd_hip_has_device_code() == 1


def test_struct_layout_matches_header_sizes():
    # config: ptr + 6 doubles + ... ; a drift between header and ctypes mirror would corrupt every call
    L = rxmd_amd.load_library()
    cfg = _lib.RxmdConfig()
    L.rxmd_hip_default_config(C.byref(cfg))
    assert list(cfg.vprocs) == [1, 1, 1] and cfg.isQEq == 1 and cfg.NMAXQEq == 500 and cfg.QEq_tol == 1e-7
    assert cfg.qstep == 1 and cfg.dt_fs == 0.25 and cfg.Lex_fqs == 1.0 and cfg.Lex_k == 2.0 and cfg.lattice[5] == 90.0
    assert cfg.lg == 0 and cfg.pqeq_path is None and cfg.efield_dir == 0 and cfg.reserved1 == 0 and cfg.efield_strength == 0.0
    assert C.sizeof(cfg) == 168 and _lib.RxmdConfig.pqeq_path.offset == 144 and _lib.RxmdConfig.efield_strength.offset == 160   # = gcc's layout of include/rxmd_hip.h


def _host_table(ff, which, npt=None):
    L = rxmd_amd.load_library()
    info = np.zeros(64)
    nb = L.rxmd_host_ffield_table(ff.encode(), None, 6, info.ctypes.data_as(C.c_void_p), 64)
    assert nb > 0
    nso = int(info[0])
    n = None
    if npt is not None:
        n = np.zeros(nso + 2, np.int64); n[1:1 + len(npt)] = npt
    if which == 6:
        return info
    out = np.zeros(nb * 5000 if which < 5 else nb + 1)
    rc = L.rxmd_host_ffield_table(ff.encode(), None if n is None else n.ctypes.data_as(C.c_void_p), which, out.ctypes.data_as(C.c_void_p), out.size)
    assert rc == nb
    return out.reshape(nb, 5000) if which < 5 else out


@pytest.mark.parametrize("case", ["rdx168", "ice644", "rdx168_lg"])
def test_ffield_tables_and_cutoffs_match_oracle(case):
    ff, names, frac, lat = oa.make_system(case)
    lg = case.endswith("_lg")                      # --lg: LG ffield format, low-gradient terms in the vdW table (init.F90:496-514)
    lat2, ranks = oa.geninit(names, frac, lat, oa.ffield_names(ff, lg=lg))
    o = oa.Oracle(ff, lat2, ranks, lg=lg)
    was = rxmd_amd.load_library().rxmd_host_ffield_lg(1 if lg else 0)
    try:
        _tables_and_cutoffs(o, ff, ranks)
    finally:
        rxmd_amd.load_library().rxmd_host_ffield_lg(was)
    if lg:
        from rxmd_amd import system
        assert system.ffield_type_names(ff, lg=True)[1:5] == ["C", "H", "O", "N"]
        lat3, rec = system.geninit(ff, names, frac, lat, lg=True)
        assert np.array_equal(np.floor(rec[:, 7]).astype(int), ranks[0]["type"]) and np.allclose(rec[:, :3], ranks[0]["rnorm"], rtol=0, atol=1e-15)


def _tables_and_cutoffs(o, ff, ranks):
    nso = int(o.info()[15])
    npt = np.bincount(ranks[0]["type"], minlength=nso + 1)[1:]
    for which in range(5):
        t = _host_table(ff, which, npt)
        to = o.table(which)
        assert t.shape == to.shape
        assert np.allclose(t, to, rtol=1e-14, atol=0), which
    rc = _host_table(ff, 5, npt)
    rco = np.zeros(len(rc) - 1); o.L.rxo_get_rc(o.w, rco.ctypes.data_as(C.c_void_p))
    present = rco > 0          # the reference zeroes rc (not rc2) of bond rows whose atom types are absent, init.F90:404-413
    assert present.any() and np.array_equal(rc[:-1][present], rco[present])
    assert rc[-1] == o.info()[0]


def test_ffield_known_values_rdx():
    info = _host_table(FF_RDX, 6)
    assert [int(x) for x in info[:5]] == [7, 18, 62, 23, 9]
    nso = 7
    chi, eta, mass = info[5:5 + nso], info[5 + nso:5 + 2 * nso], info[5 + 2 * nso:5 + 3 * nso]
    assert chi[0] == 5.7254 and eta[0] == 2 * 6.9235 and mass[0] == 12.0 and mass[1] == 1.008   # eta doubled, param.F90:361
    rc = _host_table(FF_RDX, 5, [1, 1, 1, 1, 0, 0, 0])
    assert abs(rc[-1] - 3.16) < 1e-9                  # maxrc of the RDX system (N-N)


def test_geninit_matches_numpy_restatement_and_splits_domains():
    names, frac, lat = system.read_xyz(os.path.join(oa.INP, "rdx.xyz"))
    ffn = oa.ffield_names(FF_RDX)
    for mc, vp in [((1, 1, 1), (1, 1, 1)), ((2, 3, 1), (1, 1, 1)), ((2, 2, 2), (2, 1, 1)), ((2, 2, 2), (2, 2, 2))]:
        lat_o, ranks = oa.geninit(names, frac, lat, ffn, mc=mc, vprocs=vp)
        tot = 0
        for p, r in enumerate(ranks):
            lat_s, rec = system.geninit(FF_RDX, names, frac, lat, mc=mc, vprocs=vp, myid=p)
            assert np.allclose(lat_s, lat_o)
            assert np.array_equal(rec[:, :3], r["rnorm"])
            assert np.array_equal(np.rint(rec[:, 7]).astype(int), r["type"])
            gid = np.rint((rec[:, 7] - np.rint(rec[:, 7])) * 1e13).astype(np.int64)
            assert np.array_equal(gid, r["gid"])
            tot += len(rec)
        assert tot == 168 * mc[0] * mc[1] * mc[2]


def test_rxff_reader_on_file_written_by_the_reference_geninit():
    path = os.path.join(oa.GOLD, "rdx_mc121_v121_rxff.bin")    # reference init/geninit -mc 1 2 1 -v 1 2 1
    names, frac, lat = system.read_xyz(os.path.join(oa.INP, "rdx.xyz"))
    tot = 0
    for p in range(2):
        latf, vp, rec = system.read_rxff(path, p)
        assert vp == [1, 2, 1] and np.allclose(latf, [13.18, 23.14, 10.71, 90, 90, 90])
        lat_s, mine = system.geninit(FF_RDX, names, frac, lat, mc=(1, 2, 1), vprocs=(1, 2, 1), myid=p)
        assert rec.shape == mine.shape
        assert np.array_equal(rec, mine)               # bit-identical records, incl. atype = type + gid*1e-13 + 1e-14
        tot += len(rec)
    assert tot == 336


def test_rxff_write_read_roundtrip(tmp_path):
    names, frac, lat = system.read_xyz(os.path.join(oa.INP, "rdx.xyz"))
    recs = []
    for p in range(2):
        lat_s, r = system.geninit(FF_RDX, names, frac, lat, mc=(2, 1, 1), vprocs=(2, 1, 1), myid=p)
        recs.append(r)
    f = tmp_path / "rxff.bin"
    system.write_rxff(str(f), lat_s, (2, 1, 1), recs, current_step=7)
    for p in range(2):
        l2, vp, r2 = system.read_rxff(str(f), p)
        assert vp == [2, 1, 1] and np.array_equal(r2, recs[p]) and np.allclose(l2, lat_s)


def test_rxmd_in_parser():
    cfg = system.parse_rxmd_in(os.path.join(oa.INP, "rxmd.in"))
    assert cfg["mdmode"] == 1 and cfg["dt"] == 0.25 and cfg["ntime_step"] == 100 and cfg["vprocs"] == (1, 1, 1)
    assert cfg["isQEq"] == 1 and cfg["NMAXQEq"] == 500 and cfg["QEq_tol"] == 1e-7 and cfg["qstep"] == 1
    assert cfg["isXYZ"] is True and cfg["isBinary"] is False


def test_rxmd_in_unknown_keyword_is_fatal(tmp_path):
    p = tmp_path / "rxmd.in"
    p.write_text("mdmode 1\nbogus 3\n")
    with pytest.raises(ValueError):
        system.parse_rxmd_in(str(p))


def test_engine_fails_loudly_without_gpu_or_with_bad_input():
    import torch
    names, frac, lat = system.read_xyz(os.path.join(oa.INP, "rdx.xyz"))
    if not torch.cuda.is_available():
        with pytest.raises(rxmd_amd.RxmdError) as ei:
            rxmd_amd.RxmdEngine(FF_RDX, lat)
        assert ei.value.code == -6 and "no CPU path" in str(ei.value)
    with pytest.raises(rxmd_amd.RxmdError) as ei:
        rxmd_amd.RxmdEngine("/nonexistent/ffield", lat)
    assert ei.value.code == -2
    with pytest.raises(rxmd_amd.RxmdError) as ei:
        rxmd_amd.RxmdEngine(FF_RDX, [10, 10, 10, 60, 60, 150])       # coplanar lattice vectors: no box (skewed boxes as such are fine)
    assert ei.value.code == -1


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "rxmd_amd")):
        if "build" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in txt.lower() or f == "__init__.py" and False, "%s mentions the oracle" % f


@pytest.mark.parametrize("case,steps,pq", [("rdx168_md10", 10, False), ("sicnp547_pqeq_md5", 5, True)])
def test_xyz_frame_is_byte_identical_to_the_reference_writer(case, steps, pq):
    """format_xyz (WriteXYZ, fileio.F90:241-355) fed with the oracle's state after the same number of steps must give the file
    the real reference wrote (kept as text in the golden): f12.5/f8.3 columns, or es20.12 + shell columns with PQEq."""
    import oracle_api as oa
    from rxmd_amd import system
    g = np.load(os.path.join(oa.GOLD, case + ".npz"))
    name = "sicnp" if pq else "rdx168"
    ff, names, frac, lat = oa.make_system(name)
    lat2, ranks = oa.geninit(names, frac, lat, oa.ffield_names(ff))
    o = oa.Oracle(ff, lat2, ranks, pqeq=oa.PQEQ_SICNP if pq else None)
    o.qeq(); o.force(); o.step(steps)
    txt = system.format_xyz(lat2, system.ffield_type_names(ff), o.gids(), o.types(), o.pos(), o.charges(), o.spos() if pq else None)
    ref = str(g["xyz_last"])
    a, b = txt.split("\n"), ref.split("\n")
    assert len(a) == len(b)
    if pq:      # 13 significant digits of a trajectory that the oracle reproduces to ~1e-12: compare numerically, layout exactly
        for x, y in zip(a[2:-1], b[2:-1]):
            assert len(x) == len(y) and x[:3] == y[:3] and x[83:92] == y[83:92]
            assert np.allclose([float(t) for t in x[3:83].split()] + [float(t) for t in x[92:].split()],
                               [float(t) for t in y[3:83].split()] + [float(t) for t in y[92:].split()], rtol=1e-9, atol=1e-12)
        assert a[:2] == b[:2]
    else:
        assert txt == ref


def test_fortran_binding_module_compiles_against_the_reference_modules(tmp_path):
    """bindings/rxmd_hip_mod.F90 (QEq_hip / FORCE_hip in the reference's argument shapes) must compile against the reference's own
    `atoms` module; needs amdflang and the module files of the oracle/_ref serial build (this container only)"""
    import shutil, subprocess
    flang = "/opt/rocm/bin/amdflang"
    mods = os.path.join(ROOT, "oracle", "_ref", "build_ser")
    if not (os.path.exists(flang) and os.path.exists(os.path.join(mods, "atoms.mod"))):
        pytest.skip("amdflang or the reference module files are not here")
    for src in ("rxmd_hip_mod.F90", "smoke_c_abi.F90"):
        r = subprocess.run([flang, "-cpp", "-DNOMPI", "-I" + mods, "-c", os.path.join(ROOT, "bindings", src), "-o", str(tmp_path / (src + ".o")),
                            "-module-dir", str(tmp_path)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]


def test_bench_refuses_a_launcher_mismatch_and_probe_switches():
    """bench.py --gpus N under a launcher of another size is an error, never a silent 1-rank run; work-skipping switches are refused"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="3", RANK="0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert p.returncode != 0 and "must agree" in p.stderr and "{" not in p.stdout
    env = dict(os.environ, RXMD_E4B_PROBE="1"); env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py")], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert p.returncode != 0 and "refusing" in p.stderr and "{" not in p.stdout
    sys.path.insert(0, root)
    import bench
    assert [bench.vprocs_for(n) for n in (1, 2, 4, 8)] == [(1, 1, 1), (2, 1, 1), (2, 2, 1), (2, 2, 2)]


def test_default_library_carries_no_work_skipping_switch():
    """timing probes that skip work, stripped-down kernels and variant taps are compiled only with -DRXMD_EXPERIMENTS (`make experiments` ->
    librxmd_hip_exp.so): the product library has no environment switch that changes what is computed"""
    blob = open(os.path.join(os.path.dirname(os.path.abspath(_lib.__file__)), "librxmd_hip.so"), "rb").read()
    for word in (b"_PROBE", b"RXMD_S10", b"RXMD_LEVEL1_RECORDS", b"RXMD_ISO_", b"RXMD_NO_XCD_SWIZZLE", b"RXMD_QEQ_NO_PREPASS", b"k_spmv_bisect"):
        assert word not in blob, word


def test_readme_lists_the_switches_of_the_library():
    """one table of environment switches (rxmd_amd/csrc/options.def), read once per engine at rxmd_hip_create; README.md shows exactly what the
    libraries themselves print (rxmd_host_describe_options; regenerate with scripts/gen_readme_options.py), and no other translation unit calls getenv"""
    import glob, re, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "scripts"))
    import gen_readme_options as gro
    prod, exp = gro.blocks()
    rows = [l for l in prod.split("\n") if l]
    assert len(rows) >= 20 and all(re.match(r"^\| `RXMD_[A-Z0-9_]+` \| [^|]+ \| .+ \|$", l) for l in rows) and "(exp)" not in prod and "_PROBE" not in prod
    readme = open(os.path.join(root, "README.md")).read()
    a = readme.index("<!-- options:begin"); b = readme.index("<!-- options:end -->")
    assert prod in readme[a:b], "README.md is out of date: python scripts/gen_readme_options.py"
    if exp is not None:                              # the experiments build is present (it is not part of build(); the GPU box gets whatever was built here)
        a = readme.index("<!-- options-exp:begin"); b = readme.index("<!-- options-exp:end -->")
        assert exp in readme[a:b], "README.md is out of date: python scripts/gen_readme_options.py"
    for f in glob.glob(os.path.join(root, "rxmd_amd", "csrc", "*")):
        if f.endswith((".hip", ".cpp", ".h")) and not f.endswith("options.cpp"):
            assert "getenv" not in open(f).read(), f


def test_bench_line_stays_under_the_drivers_tail_and_ends_with_the_legs():
    """bench.py prints the compact form of its record: every key of the driver's contract, `roofline` and `cpu_baseline` without their tables, and `legs`
    (one short record per leg) as the LAST key, under 8 KB -- the driver keeps the tail of stdout, and the long line of round 5 had lost alt / alt_lex / energy_per_atom there.
    Fed with the full record of the round's final run (profiles/); and the N > 1 prediction of DESIGN.md 6 (the self loop + the wire) behaves like a scaling model."""
    import json, sys
    sys.path.insert(0, ROOT)
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r06_final5_bench_default_full_line.json")))
    line = bench.compact_line(full)
    assert len(line) < 8000
    out = json.loads(line)
    assert list(out)[-1] == "legs" and [l["leg"] for l in out["legs"]][:3] == ["headline", "steady (100 steps after 10)", "isQEq 2"]
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in out, k
    r = out["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["launches_timed"] > 0 and r["traffic"] > 0
    assert out["cpu_baseline"]["kind"] == "reference" and out["cpu_baseline"]["cores"] >= 1
    assert all(len(json.dumps(l)) < 400 for l in out["legs"])
    # the prediction: more split axes = more wire and more all-reduce latency, nothing else; exposed halo only beyond the interior part of the pass
    p2, p4, p8 = (bench.predict_ms_per_step(vp, 36.55, 400262, 0.87) for vp in ((2, 1, 1), (2, 2, 1), (2, 2, 2)))
    assert p2["ms_per_step"] < p4["ms_per_step"] < p8["ms_per_step"] and p2["self_loop_part_ms"] == p8["self_loop_part_ms"]
    assert abs(p8["wire_ms"] - 3 * p2["wire_ms"]) < 2e-3 and p8["vector_halo_exposed_ms_per_iteration"] == 0.0
    assert bench.predict_ms_per_step((2, 2, 2), 36.55, 400262, 0.10)["vector_halo_exposed_ms_per_iteration"] > 0.0
