"""k_list10 with single stores switched off (experiments library, RXMD_LIST_PROBE bits 4 / 8 / 16: no slot / entry / value store): where is its time?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from test_gpu_parity import _engine
for probe in ("0", "4", "8", "16", "28"):
    os.environ["RXMD_LIST_PROBE"] = probe
    os.environ["RXMD_SPMV_WIN"] = "0"; os.environ["RXMD_NONBOND_WIN"] = "0"
    e = _engine("rdx168", (18, 18, 18), qeq_mode=1, NMAXQEq=2)
    e.QEq(); e.FORCE(); e.step(2); e.reset_timers(); e.step(6)
    st = e.stats()
    print("probe", probe, "k_list10 %.3f ms  lists %.3f ms" % (st["ms_k_list10"] / 6, st["ms_lists"] / 6), flush=True)
    e.close()
