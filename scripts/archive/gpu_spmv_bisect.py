"""stripped-down forms of the matrix pass, timed in isolation (debug tap 102): streams only / + gathers / + ghost sums and reductions / + tail"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from test_gpu_parity import _engine
for case, mc in (("rdx168", (18, 18, 18)), ("ice644", (60, 35, 40))):
    e = _engine(case, mc, qeq_mode=1)
    e.QEq(); e.FORCE()
    st = e.stats(); gb = (st["nnz10"] * 12 + st["natoms"] * 56) / 1e9
    for rep in range(2):
        ms = e.debug(102, cap=8)
        print(case, "streams only %.4f ms (%.2f TB/s) | + gathers %.4f | + ghost sums, 4 reductions %.4f | + tail operands %.4f | + two row stores %.4f | + partials %.4f | one 32-B store instead %.4f" % (ms[0], gb / ms[0], ms[1], ms[2], ms[3], ms[4], ms[5], ms[6]))
    pr = e.debug(100, cap=16); print("plain read probe ms:", [round(pr[2 * g], 4) for g in range(4)])
    e.close()
