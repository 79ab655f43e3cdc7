"""window pass with 256 / 384 entries in flight on the water configuration (rows of ~357 entries) and on RDX, one process each (debug tap 104)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
import rxmd_amd
from rxmd_amd import system
os.environ["RXMD_ISO_REPS"] = "40"; os.environ["RXMD_PLACE_TRIES"] = "1"
for wl in ("water", "rdx"):
    ff, names, frac, lat, cells, wname, pqeq = bench.make_workload(wl, bench.ATOMS_PER_GPU_CELLS)
    lat_super, rec = system.geninit(ff, names, frac, lat, mc=cells)
    e = rxmd_amd.RxmdEngine(ff, lat_super, qeq_mode=1)
    e.set_atoms_rxff(rec); e.QEq()
    st = e.stats()
    for rep in range(2):
        iso = e.debug(104, cap=24)
        print("%s (%d atoms, n10 %.0f, longest row %d): window pass 256 in flight %.4f ms, 384 in flight %.4f ms, row pass %.4f ms" % (wl, st["natoms"], st["nnz10"] / st["natoms"], st["max_n10"], iso[0], iso[2], iso[1]), flush=True)
    e.close()
