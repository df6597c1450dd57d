#!/usr/bin/env python3
"""Derives the diagnostic builds of the step kernel from the shipped sources (which carry no diagnostic code):
    python profiles/tools/make_diag_variants.py            -> build/diag/libtrm_{compute_only,memory_only}.so
  compute_only   k_column without global traffic: plausible register inputs instead of the loads, stores behind a
                 condition that is never true (the arithmetic stays alive)
  memory_only    k_column's loads and stores with no arithmetic in between
Each edit asserts that it matched exactly once, so a change of the kernel that breaks a variant fails loudly here.
Use with TRM_LIBRARY=build/diag/libtrm_<variant>.so (terrarium.jl_amd/_capi.py) and profiles/tools/ab_step.py."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "terrarium.jl_amd", "csrc")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math".split()


def edit(text, old, new):
    assert text.count(old) == 1, f"diag variant: expected exactly one match of {old!r}"
    return text.replace(old, new)


LOADS = """    c.U = ldg(v.U, cb0);
    c.sat = ldg(v.sat, cb0);
    c.psi = RICHARDS ? ldg(v.psi, cb0) : NF(0);
"""
TLOADS = """        c.T = ldg(v.T, cb0);
        c.liq = ldg(v.liq, cb0);
"""


def compute_only(t):
    t = edit(t, LOADS, """    c.U = NF(1.0e6) + NF(cb0) * NF(3.0);
    c.sat = NF(0.5) + NF(cb0 & 31) * NF(0.01);
    c.psi = NF(-1) - NF(cb0 & 15) * NF(0.1);
""")
    t = edit(t, TLOADS, "        c.T = NF(2) + NF(cb0 & 7);\n        c.liq = NF(1);\n")
    t = edit(t, "    if (ln.act) {\n        const unsigned cb = block_local(cb0), ib = block_local(ib0);\n        stg(v.U, cb, n.U);",
             "    if (ln.act && n.U == NF(-12345.678)) {\n        const unsigned cb = block_local(cb0), ib = block_local(ib0);\n        stg(v.U, cb, n.U);")
    return t


def memory_only(t):
    return edit(t, "    // ---- boundary inputs of the column ---", """    if (PROG == PROG_EULER) {
        if (ln.act) {
            stg(v.U, cb0, c.U + a.dt); stg(v.T, cb0, c.T + a.dt); stg(v.liq, cb0, c.liq + a.dt);
            if (RICHARDS) { stg(v.sat, cb0, c.sat + a.dt); stg(v.psi, cb0, c.psi + a.dt); }
            if (a.write_kf) stg(v.Kf, cb0, c.U + c.sat);
        }
        return;
    }
    // ---- boundary inputs of the column ---""")


def main():
    out = os.path.join(ROOT, "build", "diag")
    os.makedirs(out, exist_ok=True)
    for name, fn in (("compute_only", compute_only), ("memory_only", memory_only)):
        work = os.path.join(out, "src_" + name)
        shutil.rmtree(work, ignore_errors=True)
        shutil.copytree(SRC, os.path.join(work, "terrarium.jl_amd", "csrc"))
        shutil.copytree(os.path.join(ROOT, "include"), os.path.join(work, "include"))
        path = os.path.join(work, "terrarium.jl_amd", "csrc", "trm_column.hpp")
        with open(path) as f:
            text = f.read()
        with open(path, "w") as f:
            f.write(fn(text))
        so = os.path.join(out, f"libtrm_{name}.so")
        subprocess.check_call(["make", "-C", os.path.join(work, "terrarium.jl_amd", "csrc"), "-s", "-j8", "OUT=" + so, "OBJDIR=" + os.path.join(work, "obj"),
                               "EXTRA=" + " ".join(sys.argv[1:])])
        print("built", so)


if __name__ == "__main__":
    main()
