"""window pass, synthetic timing probe (debug tap 103) next to the stripped-down forms of the row kernel (tap 102)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from test_gpu_parity import _engine
e = _engine("rdx168", (18, 18, 18), qeq_mode=1)
e.QEq(); e.FORCE()
st = e.stats(); gb = (st["nnz10"] * 12 + st["natoms"] * 56) / 1e9
for rep in range(3):
    iso = e.debug(104, cap=4); print('real kernels back to back: window pass %.4f ms, row pass %.4f ms' % (iso[0], iso[1]))
    w = e.debug(103, cap=12)
    ms = e.debug(102, cap=8)
    print("bit-word window probe: nt loads %.4f / with ghost sums %.4f | plain loads %.4f / %.4f || 16-bit slot window probe: %.4f, rows through rows_sorted %.4f, real windows through win_k %.4f, both %.4f ms (largest window %d units) || row kernel: streams only %.4f | + gathers %.4f | full %.4f" % (w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], int(w[8]), ms[0], ms[1], ms[5]))
e.close()
