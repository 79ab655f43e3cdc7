#!/usr/bin/env python3
"""Parity of several builds of the library on small systems, each in its own process under a time limit: which variant faults, which is wrong.
usage: python3 scripts/gpu_variants_check.py [tag[:ENV=1,...] ...]      tag = default | <tag of rxmd_amd/librxmd_hip_<tag>.so>"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np
import oracle_api as oa, rxmd_amd
from rxmd_amd import system
for case, mc in (("rdx168", (1, 1, 1)), ("rdx222", (2, 2, 2)), ("rdx168", (4, 4, 4)), ("ice644", (6, 4, 4))):
    ff, names, frac, lat = oa.make_system(case)
    lat2, ranks = oa.geninit(names, frac, lat, oa.ffield_names(ff), mc=mc)
    o = oa.Oracle(ff, lat2, ranks, QEq_tol=1e-12, NMAXQEq=2000); o.qeq(); o.force(); o.step(2)
    lat3, rec = system.geninit(ff, names, frac, lat, mc=mc)
    e = rxmd_amd.RxmdEngine(ff, lat3, QEq_tol=1e-12, NMAXQEq=2000, device=0)
    e.set_atoms_rxff(rec); e.QEq(); e.FORCE(); e.step(2)
    a = e.atoms(); qo, fo = o.charges(), o.forces()
    qerr = (np.abs(a["q"] - qo) / np.maximum(np.abs(qo), np.sqrt((qo ** 2).mean()))).max()
    ferr = (np.abs(a["f"] - fo).max(axis=1) / np.maximum(np.abs(fo).max(axis=1), np.sqrt((fo ** 2).mean()))).max()
    pe = e.energy()["PE"]; eerr = max(abs(x - y) / abs(y) for x, y in zip(pe, o.energy()) if abs(y) > 1e-6)
    print("   %%-8s %%-10s natoms %%6d  q %%.1e  f %%.1e  E %%.1e  %%s" %% (case, mc, len(rec), qerr, ferr, eerr, "ok" if qerr < 1e-6 and ferr < 1e-6 and eerr < 1e-8 else "WRONG"), flush=True)
    e.close()
''' % (ROOT, ROOT)
for spec in (sys.argv[1:] or ["default"]):
    tag, _, envs = spec.partition(":")
    env = dict(os.environ)
    if tag != "default":
        env["RXMD_HIP_LIB"] = os.path.join(ROOT, "rxmd_amd", "librxmd_hip_%s.so" % tag)
    for kv in filter(None, envs.split(",")):
        k, _, v = kv.partition("="); env[k] = v
    print("==", spec, flush=True)
    try:
        p = subprocess.run([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
        out = p.stdout
        print(out if p.returncode == 0 else ("   rc %d\n" % p.returncode) + "\n".join(out.split("\n")[:12] + ["   ..."] + out.split("\n")[-6:]), flush=True)
    except subprocess.TimeoutExpired as ex:
        print("   TIMEOUT after 300 s\n", (ex.stdout or "")[-1500:], flush=True)
