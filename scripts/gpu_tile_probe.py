#!/usr/bin/env python3
"""Timing probe of a half-storage matrix pass over 3-D tiles (qeq.hip: k_spmv_tile_probe, debug tap 105; experiments build), RDX 979,776 atoms.
usage: RXMD_HIP_LIB=rxmd_amd/librxmd_hip_exp.so python3 scripts/gpu_tile_probe.py"""
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
e = rxmd_amd.RxmdEngine(ff, lat_s, QEq_tol=1e-7, NMAXQEq=500, device=0, qeq_mode=1)
e.set_atoms_rxff(rec); e.QEq()
for rep in range(3):
    t = e.debug(105, cap=8)
    print("tile probe: half-storage pass %.4f ms (without the transposed LDS adds %.4f) + halo gather %.4f ms = %.4f ms ; the real window pass in the same process %.4f ms ; launch error %d" % (t[0], t[4], t[1], t[0] + t[1], t[2], int(t[3])), flush=True)
e.close()
