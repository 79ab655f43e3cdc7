"""real window pass / row pass: back to back (debug tap 104) against the same kernels inside the CG loop of MD steps"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from test_gpu_parity import _engine
e = _engine("rdx168", (18, 18, 18), qeq_mode=1)
e.QEq(); e.FORCE()
def loop(tag, nsteps=4):
    s0 = e.stats(); e.step(nsteps); s1 = e.stats()
    nl = s1["spmv_launches"] - s0["spmv_launches"]
    print("%-28s in the CG loop: %.4f ms per pass (%d passes, %.1f per step)" % (tag, (s1["ms_qeq_spmv"] - s0["ms_qeq_spmv"]) / nl, nl, nl / nsteps))
for rep in range(3):
    os.environ["RXMD_ISO_REPS"] = "100"
    iso = e.debug(104, cap=12); print('back to back, alternating: window pass %.4f ms, row pass %.4f ms (variants: %.4f %.4f)' % (iso[0], iso[1], iso[2], iso[3]))
    os.environ["RXMD_SPMV_WIN"] = "1"; loop("window pass")
    os.environ["RXMD_SPMV_WIN"] = "0"; loop("row pass")
e.close()
