#!/usr/bin/env python3
"""A/B of environment switches on the isQEq 2 leg (extended-Lagrangian charges: one CG step per MD step, the non-CG kernels ARE the step), RDX 979,776 atoms,
alternating engines in one process.  usage: python3 scripts/gpu_ab_lex.py <reps> "ENV=1[,ENV2=1]" "-" ..."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rxmd_amd
from rxmd_amd import system
INP = os.path.join(ROOT, "tests", "golden", "inputs")
names, frac, lat = system.read_xyz(os.path.join(INP, "rdx.xyz"))
ff = os.path.join(INP, "ffield_rdx")
lat_s, rec = system.geninit(ff, names, frac, lat, mc=(18, 18, 18))
reps = int(sys.argv[1]); variants = sys.argv[2:]
for rep in range(reps):
    for v in variants:
        sets = [] if v == "-" else [kv.split("=") for kv in v.split(",")]
        for k, val in sets: os.environ[k] = val
        e = rxmd_amd.RxmdEngine(ff, lat_s, QEq_tol=1e-7, NMAXQEq=500, device=0, qeq_mode=1, isQEq=2, dt_fs=0.25)
        e.set_atoms_rxff(rec); e.QEq(); e.FORCE(); e.step(3); e.reset_timers()
        t0 = time.perf_counter(); e.step(20); dt = (time.perf_counter() - t0) / 20 * 1e3
        st = e.stats()
        print("%-34s isQEq 2: %.2f ms/step   lists %.2f  qeq %.2f  force %.2f  (nonbond %.2f bonded %.2f; e4b %.2f e3b %.2f ehb %.2f)" % (
            v, dt, st["ms_lists"] / 20, st["ms_qeq"] / 20, st["ms_force"] / 20, st["ms_nonbond"] / 20, st["ms_bonded"] / 20, st["ms_k_e4b"] / 20, st["ms_k_e3b"] / 20, st["ms_k_ehb"] / 20), flush=True)
        e.close()
        for k, _ in sets: os.environ.pop(k)
