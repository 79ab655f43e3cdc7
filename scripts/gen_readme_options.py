#!/usr/bin/env python3
"""Rewrite the generated parts of README.md's switch tables from the library's own table (rxmd_amd/csrc/options.def -> rxmd_host_describe_options):
the product library prints the product rows, the experiments build (make -C rxmd_amd/csrc experiments) prints its extra rows marked (exp).
No GPU needed.  tests/test_host_frontend.py::test_readme_lists_the_switches_of_the_library fails when README.md and the libraries differ."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def describe(so):
    lib = C.CDLL(so)
    lib.rxmd_host_describe_options.restype = C.c_int; lib.rxmd_host_describe_options.argtypes = [C.c_char_p, C.c_int]
    n = lib.rxmd_host_describe_options(None, 0)
    buf = C.create_string_buffer(n + 1); lib.rxmd_host_describe_options(buf, n + 1)
    return buf.value.decode()


def blocks():
    prod = describe(os.path.join(ROOT, "rxmd_amd", "librxmd_hip.so"))
    exp_so = os.path.join(ROOT, "rxmd_amd", "librxmd_hip_exp.so")
    exp = None
    if os.path.exists(exp_so):
        exp = "".join(l + "\n" for l in describe(exp_so).split("\n") if l and "(exp)" in l)
    return prod, exp


if __name__ == "__main__":
    p = os.path.join(ROOT, "README.md")
    s = open(p).read()
    prod, exp = blocks()
    head = "\n| variable | default | effect |\n|---|---|---|\n"
    for tag, txt in (("options", prod), ("options-exp", exp)):
        if txt is None:
            print("README.md: %s left as it is (librxmd_hip_exp.so not built)" % tag); continue
        b0 = "<!-- %s:begin (generated: python scripts/gen_readme_options.py) -->" % tag; b1 = "<!-- %s:end -->" % tag
        a, b = s.index(b0), s.index(b1)
        s = s[:a] + b0 + head + txt + s[b:]
        print("README.md: %s, %d switches" % (tag, txt.count("\n")))
    open(p, "w").write(s)
