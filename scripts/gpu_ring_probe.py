"""ring matrix pass in isolation (debug tap 101) under the environment settings given as arguments "K=V,K=V" ..."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from test_gpu_parity import _engine
e = _engine("rdx168", (18, 18, 18), qeq_mode=1)
e.QEq(); e.FORCE()
nnz = e.stats()["nnz10"]; n = e.natoms
gb = (nnz * 12 + n * 56) / 1e9
for spec in sys.argv[1:]:
    for kv in spec.split(","):
        if kv:
            k, v = kv.split("="); os.environ[k] = v
    ms = e.debug(101, cap=8)[0]
    print("%-60s %.4f ms  %.2f TB/s" % (spec, ms, gb / ms))
    for kv in spec.split(","):
        if kv:
            os.environ.pop(kv.split("=")[0])
