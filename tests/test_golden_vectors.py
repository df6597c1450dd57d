"""Committed golden vectors (tests/golden/step_vectors.npz) against the oracle (CPU) and the HIP
library (GPU).  fp64 heat / Richards(BrooksCorey) cases are compared bit for bit."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_golden_vectors as G
import workloads as W

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "step_vectors.npz"))
EXACT = {"c1_single_column_heat", "c2_n72_heat", "c3_n145_richards"}


def check(name, get, exact_override=None):
    w, nsteps = G.build_case(name)
    exact = name in EXACT if exact_override is None else exact_override
    tol = 1e-10 if w["dtype"] == np.float64 else 1e-4
    for f in W.compared_fields(w):
        a, b = get(f), GOLD[f"{name}/{f}"]
        if exact:
            assert np.array_equal(a, b), (name, f)
        else:
            assert np.max(np.abs(a.astype(np.float64) - b) / np.maximum(1.0, np.abs(b))) <= tol, (name, f)


@pytest.mark.parametrize("name", sorted(G.CASES))
def test_oracle_reproduces_golden_vectors(name):
    w, nsteps = G.build_case(name)
    orc = W.setup_oracle(w)
    orc.run(w["dt"], nsteps)
    check(name, orc.get, exact_override=True)  # the oracle is deterministic: every case bit for bit
    assert orc.status() == int(GOLD[f"{name}/status"]) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["fused", "unfused"])
@pytest.mark.parametrize("name", sorted(G.CASES))
def test_device_matches_golden_vectors(name, kernel):
    w, nsteps = G.build_case(name)
    dev = W.setup_device(w)
    dev.set_option("step_kernel", kernel)
    dev.step(w["dt"], nsteps, finalize=True)
    check(name, dev.get)
    assert dev.status() == 0
