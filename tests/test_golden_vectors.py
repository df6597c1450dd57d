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


# ---- reference-side fixtures (tests/golden/make_reference_fixtures.jl) ------------------------------------------------------------
# Produced by running the Julia script above on a machine with Terrarium.jl; NONE are committed yet (no Julia in the build image),
# so these tests skip.  When `tests/golden/reference_<case>__<field>__<steps>.bin` files exist they pin the oracle and the HIP
# library against the REFERENCE itself: tolerance 1e-10 relative (the reference may multiply by reciprocal spacings where the
# restatement divides: <= 1 ulp per operation), reported per field.
import glob
import json

REF_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REF_MANIFEST = json.load(open(os.path.join(REF_DIR, "reference_inputs", "manifest.json")))
REF_FILES = sorted(glob.glob(os.path.join(REF_DIR, "reference_*__*__*.bin")))


def reference_workload(name):
    import make_reference_inputs as R
    return R.build_case(name)[0]


def reference_vectors(name):
    out = {}
    for path in REF_FILES:
        case, field, steps = os.path.basename(path)[len("reference_"):-len(".bin")].split("__")
        if case == name:
            Nh = REF_MANIFEST[name]["Nh"]
            out.setdefault(int(steps), {})[field] = np.fromfile(path, dtype="<f8").reshape(-1, Nh)
    return out


def test_reference_inputs_are_the_workloads_of_this_repository():
    """The committed inputs of the fixture kit are what tests/workloads.py builds today (regenerate them if this fails)."""
    import make_reference_inputs as R
    for name, entry in REF_MANIFEST.items():
        w, steps = R.build_case(name)
        assert entry["Nh"] == w["Nh"] and entry["Nz"] == w["Nz"] and entry["dt"] == w["dt"] and tuple(entry["steps"]) == tuple(steps)
        for k, v in w["fields"].items():
            a = np.fromfile(os.path.join(REF_DIR, "reference_inputs", entry["fields"][k]["file"]), dtype="<f8").reshape(entry["fields"][k]["shape"])
            assert np.array_equal(a, np.asarray(v, dtype=np.float64)), (name, k)


@pytest.mark.skipif(not REF_FILES, reason="no reference-side fixtures: run tests/golden/make_reference_fixtures.jl with Terrarium.jl")
@pytest.mark.parametrize("name", sorted(REF_MANIFEST))
def test_oracle_matches_reference_fixtures(name):
    vectors = reference_vectors(name)
    if not vectors:
        pytest.skip("no fixtures for this case")
    w = reference_workload(name)
    orc, done = W.setup_oracle(w), 0
    for steps in sorted(vectors):
        for _ in range(steps - done):
            orc.timestep(w["dt"], True)
        done = steps
        for field, ref in vectors[steps].items():
            a = np.atleast_2d(orc.get(field))
            assert np.max(np.abs(a - ref) / np.maximum(1.0, np.abs(ref))) <= 1e-10, (name, field, steps)


@pytest.mark.gpu
@pytest.mark.skipif(not REF_FILES, reason="no reference-side fixtures: run tests/golden/make_reference_fixtures.jl with Terrarium.jl")
@pytest.mark.parametrize("name", sorted(REF_MANIFEST))
def test_device_matches_reference_fixtures(name):
    vectors = reference_vectors(name)
    if not vectors:
        pytest.skip("no fixtures for this case")
    w = reference_workload(name)
    dev, done = W.setup_device(w), 0
    for steps in sorted(vectors):
        dev.step(w["dt"], steps - done, finalize=True)
        done = steps
        for field, ref in vectors[steps].items():
            a = np.atleast_2d(dev.get(field))
            assert np.max(np.abs(a - ref) / np.maximum(1.0, np.abs(ref))) <= 1e-10, (name, field, steps)
