"""diagnostic: where do the ccbnd / cdbnd taps of the engine differ from the oracle's (residents / ghosts, by exchange stage)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import oracle_api as oa
from test_gpu_parity import _engine, _oracle
kw = dict(QEq_tol=1e-12, NMAXQEq=2000)
for case, mc in (("rdx222", (2, 2, 2)), ("rdx168", (3, 3, 3))):
    o = _oracle(case, mc, **kw); o.qeq(); o.force()
    e = _engine(case, mc, **kw); e.QEq(); e.FORCE()
    n = len(o.gids()); G = o.L.rxo_nghost_total(o.w, 0)
    cc_o, cd_o = o.get(109), o.get(110)
    cc_e, cd_e = e.debug(10), e.debug(8)
    pos = o.get(100, width=3)
    d = np.abs(cc_e - cc_o)
    bad = np.nonzero(d > 1e-8)[0]
    print(case, mc, "n", n, "G", G, "bad cc", len(bad), "of which residents", (bad < n).sum(), "max", d.max())
    dd = np.abs(cd_e - cd_o); print("  cd max diff", dd.max(), "bad", (dd > 1e-8).sum())
    nb_o = o.get(103)
    lat = np.array(e.lattice[:3])
    for i in bad[:12]:
        depth = np.max(np.maximum(-pos[i], pos[i] - lat))
        print("   i", i, "cc_e", cc_e[i], "cc_o", cc_o[i], "cd_e", cd_e[i], "cd_o", cd_o[i], "nbrs", nb_o[i], "depth outside box %.2f A" % depth)
    if len(bad):
        depth = np.max(np.maximum(-pos[bad], pos[bad] - lat), axis=1)
        print("  depth of bad ghosts: min %.2f max %.2f" % (depth.min(), depth.max()))
        good = np.nonzero((d <= 1e-8) & (np.arange(G) >= n) & (np.abs(cc_o) > 1e-6))[0]
        dg = np.max(np.maximum(-pos[good], pos[good] - lat), axis=1)
        print("  ghosts with matching nonzero cc: %d, depth max %.2f" % (len(good), dg.max() if len(good) else -1))
    e.close()
