import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rxmd_amd
from rxmd_amd import system
INP = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "inputs")
ff = os.path.join(INP, "ffield_rdx")
names, frac, lat = system.read_xyz(os.path.join(INP, "rdx.xyz"))
c = int(os.environ.get("CELLS", "18"))
lat_s, rec = system.geninit(ff, names, frac, lat, mc=(c, c, c))
e = rxmd_amd.RxmdEngine(ff, lat_s, NMAXQEq=int(os.environ.get("ITERS", "15")), QEq_tol=0.0, maxneighbs10=int(os.environ.get("S10", "0")))
e.set_atoms_rxff(rec)
try:
    e.QEq()
except Exception as ex:
    print("qeq raised", ex)
st = e.stats()
ms = st["ms_qeq_spmv"] / max(st["spmv_launches"], 1)
print(json.dumps({"tag": os.environ.get("TAG", ""), "spmv_ms": ms, "GBs": (st["nnz10"] * 12 + st["natoms"] * 56) / ms / 1e6, "launches": st["spmv_launches"], "ms_lists": st["ms_lists"]}))

if os.environ.get("PROBE"):
    o = e.debug(100, cap=16)
    for g in range(4):
        print("stream probe grid", [2048, 8192, 32768, 131072][g], "ms", round(o[2 * g], 4), "GB/s", round(o[2 * g + 1] / o[2 * g] / 1e6))
print("stride", st["n10_stride"])
