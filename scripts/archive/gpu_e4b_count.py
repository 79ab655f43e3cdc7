"""RXMD_E4B_PROBE=5: k_e4b counts the batches it evaluates and the torsions in them (PE(8), PE(9) of one FORCE call are the counters)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rxmd_amd
from rxmd_amd import system
INP = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "inputs")
ff = os.path.join(INP, "ffield_rdx")
names, frac, lat = system.read_xyz(os.path.join(INP, "rdx.xyz"))
c = int(sys.argv[1]) if len(sys.argv) > 1 else 18
lat_super, rec = system.geninit(ff, names, frac, lat, mc=(c, c, c), vprocs=(1, 1, 1), myid=0)
eng = rxmd_amd.RxmdEngine(ff, lat_super, qeq_mode=1)
eng.set_atoms_rxff(rec)
eng.QEq()
pe = eng.FORCE()
n = len(rec)
print("atoms %d  pe8 %g  pe9 %g  (x natoms: batches %g entries %g; per wavefront of two atoms: %.3f batches, %.2f entries, fill %.3f)" %
      (n, pe[8], pe[9], pe[8] * n, pe[9] * n, pe[8] * n / (n / 2), pe[9] * n / (n / 2), pe[9] / max(pe[8], 1e-300) / 64.0))
print("raw", pe[8], pe[9])
