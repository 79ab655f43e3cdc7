#!/usr/bin/env python3
"""Torsions visited twice (default until round 6) against torsions visited once (RXMD_E4B_ONCE=1), RDX 979,776 atoms: the same state, the same
process, engines alternating; FORCE on ONE stream (RXMD_NO_BOND_OVERLAP=1: every kernel alone on the GPU) and with the bonded chain on its own stream.
Prints per FORCE call: the torsion kernel (+ delivery), the bonded section, FORCE as a whole; and the largest force difference between the two forms.
usage: python3 scripts/gpu_ab_e4b.py [cells] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rxmd_amd
from rxmd_amd import system
INP = os.path.join(ROOT, "tests", "golden", "inputs")
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 18
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
names, frac, lat = system.read_xyz(os.path.join(INP, "rdx.xyz"))
ff = os.path.join(INP, "ffield_rdx")
lat_s, rec = system.geninit(ff, names, frac, lat, mc=(cells, cells, cells))
forces = {}
VARIANTS = [tuple(v.split(":")) for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else ("0:4:0", "0:1:1", "1:4:0", "1:1:1"))]   # visits-once : wavefronts per workgroup : persistent grid
for overlap in ("1", "0"):
    os.environ["RXMD_NO_BOND_OVERLAP"] = overlap
    for rep in range(reps):
        for once, wpb, pers in VARIANTS:
            os.environ["RXMD_E4B_ONCE"] = once; os.environ["RXMD_E4B_WPB"] = wpb; os.environ["RXMD_E4B_PERSIST"] = pers
            e = rxmd_amd.RxmdEngine(ff, lat_s, QEq_tol=1e-7, NMAXQEq=500, device=0, qeq_mode=1)
            e.set_atoms_rxff(rec); e.QEq(); e.FORCE(); e.reset_timers()
            for _ in range(10): e.FORCE()
            st = e.stats()
            print("one stream %s  visits %s  wavefronts per workgroup %s  persistent %s: k_e4b %.3f ms  k_e3b %.3f  k_ehb %.3f  assemble %.3f  bonded %.3f  force %.3f" % (
                overlap, "1" if once == "1" else "2", wpb, pers, st["ms_k_e4b"] / 10, st["ms_k_e3b"] / 10, st["ms_k_ehb"] / 10, st["ms_k_assemble"] / 10, st["ms_bonded"] / 10, st["ms_force"] / 10), flush=True)
            if rep == 0 and overlap == "1" and once not in forces:
                a = e.atoms(); o = np.argsort(a["gid"]); forces[once] = a["f"][o].copy(); forces["pe" + once] = np.array(e.energy()["PE"])
            e.close()
    if overlap == "1":
        d = np.abs(forces["0"] - forces["1"])
        print("largest |f(one visit) - f(two visits)| %.3e kcal/mol/A (largest |f| %.3e);  PE difference %.3e" % (d.max(), np.abs(forces["0"]).max(), np.abs(forces["pe0"] - forces["pe1"]).max()), flush=True)
