#!/usr/bin/env python3
"""Where the torsion kernel's time goes (experiments build: RXMD_E4B_PROBE 1 = set-up only, 2 = set-up + enumeration, 0 = all), RDX 979,776 atoms.
usage: RXMD_HIP_LIB=rxmd_amd/librxmd_hip_exp.so python3 scripts/gpu_e4b_phases.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rxmd_amd
from rxmd_amd import system
INP = os.path.join(ROOT, "tests", "golden", "inputs")
names, frac, lat = system.read_xyz(os.path.join(INP, "rdx.xyz"))
ff = os.path.join(INP, "ffield_rdx")
lat_s, rec = system.geninit(ff, names, frac, lat, mc=(18, 18, 18))
for probe in (sys.argv[1].split(",") if len(sys.argv) > 1 else ("0", "1", "2", "0")):
    os.environ["RXMD_E4B_PROBE"] = probe
    e = rxmd_amd.RxmdEngine(ff, lat_s, QEq_tol=1e-7, NMAXQEq=500, device=0, qeq_mode=1)
    e.set_atoms_rxff(rec); e.QEq(); e.FORCE(); e.reset_timers()
    for _ in range(5): e.FORCE()
    st = e.stats()
    print("visits %s  probe %s: k_e4b %.3f ms  k_e3b %.3f  k_ehb %.3f" % ("2" if os.environ.get("RXMD_E4B_ONCE", "1") == "0" else "1", probe, st["ms_k_e4b"] / 5, st["ms_k_e3b"] / 5, st["ms_k_ehb"] / 5), flush=True)
    if probe == "5" and os.environ.get("RXMD_E3Q_COUNT"):
        pe = e.energy()["PE"]; print("   angle kernel: evaluate calls per FORCE %.0f, angles %.0f (%.1f per call; %.2f calls per wavefront of 16 atoms; %.2f angles per atom)" % (pe[5], pe[6], pe[6] / max(pe[5], 1), pe[5] / (len(rec) / 16), pe[6] / len(rec)), flush=True)
    if probe == "5":
        pe = e.energy()["PE"]; print("   evaluate calls per FORCE %.0f, entries %.0f (%.1f per call; %.2f calls per wavefront of 8 atoms)" % (pe[8], pe[9], pe[9] / max(pe[8], 1), pe[8] / (len(rec) / 8)), flush=True)
    e.close()
