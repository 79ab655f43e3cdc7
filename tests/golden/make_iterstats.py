#!/usr/bin/env python3
"""tests/golden/rdx222_md<N>_dq.npz: how far the REFERENCE's own charges at QEq_tol 1e-7 are from the converged solution, step by step.

TEST INFRASTRUCTURE.  RDX 2 x 2 x 2, N = 110 or 1000 MD steps at the default rxmd.in settings through the bit-path oracle (oracle/rxmd_oracle.c) -- which
reproduces the reference's iteration count of every one of these steps (asserted below against tests/golden/rdx222_md<N>.npz, the reference's own
run) -- and after each step a second oracle solves the same geometry at QEq_tol 1e-12 from the same charges: dq[s] = max |q_1e-7 - q_1e-12|.
The engine's charges are held against this distribution in tests/test_gpu_parity.py::test_iteration_statistics_over_1000_steps.
Usage: python tests/golden/make_iterstats.py [N]      (N = 110: about 5 minutes; N = 1000: about 45)
"""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_api as oa

if __name__ == "__main__":
    NS = int(sys.argv[1]) if len(sys.argv) > 1 else 110
    ff, names, frac, lat = oa.make_system("rdx222")
    lat2, ranks = oa.geninit(names, frac, lat, oa.ffield_names(ff), mc=(2, 2, 2))
    o = oa.Oracle(ff, lat2, ranks); o.qeq(); o.force()
    g = np.load(os.path.join(HERE, "rdx222_md%d.npz" % NS))
    its, dqs = [o.L.rxo_qeq_iters(o.w)], []
    L = np.asarray(lat2[:3])
    for s in range(NS):
        o.step(1); its.append(o.L.rxo_qeq_iters(o.w))
        t = oa.Oracle(ff, lat2, [dict(rnorm=o.pos() / L, type=o.types(), gid=o.gids())], QEq_tol=1e-12, NMAXQEq=2000, q0=[o.charges()]); t.qeq()
        dqs.append(np.abs(t.charges() - o.charges()).max())
    ref = [int(x) for x in g["qeq_iters"]]
    same = next((k for k in range(len(its)) if its[k] != ref[k]), len(its))
    print("the oracle reproduces the reference's iteration count for the first %d of %d calls" % (same, len(its)))
    assert same >= min(len(its), 300), "the oracle left the reference's bit path early"
    np.savez_compressed(os.path.join(HERE, "rdx222_md%d_dq.npz" % NS), qeq_iters=np.array(its), dq_tight=np.array(dqs))
    print("mean iterations, steps 11..N: %.2f ; max dq %.3e, 90th percentile %.3e" % (np.mean(its[11:]), max(dqs), np.percentile(dqs, 90)))
