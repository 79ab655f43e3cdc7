#!/usr/bin/env python3
"""Are the forces the same BITS from engine to engine?  RDX 2x2x2 and 5x5x5, two-visit torsions, angle kernel in both forms, torsion instances."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rxmd_amd
from rxmd_amd import system
import oracle_api as oa
for case, mc in (("rdx222", (2, 2, 2)), ("rdx168", (5, 5, 5))):
    ff, names, frac, lat = oa.make_system(case)
    lat3, rec = system.geninit(ff, names, frac, lat, mc=mc)
    ref = {}
    for once in ("0", "1"):
        for q, slots in (("3", ""), ("3", ""), ("3", "32"), ("3", "4"), ("0", ""), ("0", ""), ("0", "32")):
            os.environ["RXMD_E3B_QUEUE"] = q; os.environ["RXMD_E4B_ONCE"] = once
            if slots: os.environ["RXMD_E4B_SLOTS"] = slots
            else: os.environ.pop("RXMD_E4B_SLOTS", None)
            e = rxmd_amd.RxmdEngine(ff, lat3, QEq_tol=1e-12, NMAXQEq=2000); e.set_atoms_rxff(rec)
            e.QEq(); e.FORCE(); f = e.atoms()["f"].copy(); e.close()
            key = (once, q)
            if key not in ref: ref[key] = f
            d = np.abs(f - ref[key])
            print("%s once %s queue %s slots %-3s: differs from the first of its kind in %d of %d components, max %.3e" % (case, once, q, slots or "-", int((d > 0).sum()), d.size, d.max()), flush=True)
