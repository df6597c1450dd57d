"""Generate tests/golden/step_vectors.npz: seeded inputs and expected outputs of the SoilModel /
LandModel step for small cases of BASELINE.json's configurations.

The reference (Julia) cannot run in the build image, so the expected outputs come from the CPU
oracle (oracle/terrarium_oracle.hpp), which is itself pinned by the reference's known-answer tests
(tests/test_oracle_known_answers.py).  The fixture freezes those outputs: a later change of either
the oracle or the HIP library that moves a single bit of the bit-exact cases fails
tests/test_golden_vectors.py.  Data only -- inputs and outputs, no reference source text.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]

import workloads as W  # noqa: E402

CASES = {
    # name: (config, hydraulics, Nz, columns, nsteps, dtype)
    "c1_single_column_heat": ("heat", "default", 20, 1, 50, np.float64),
    "c2_n72_heat": ("heat", "default", 30, 12, 50, np.float64),
    "c3_n145_richards": ("richards", "default", 32, 12, 50, np.float64),
    "c3_n145_richards_vg": ("richards", "vg", 32, 12, 50, np.float64),
    "c4_n145_land": ("land", "vg", 32, 12, 30, np.float64),
    "c5_f32_land": ("land", "vg", 64, 12, 20, np.float32),
    "c4_n145_land_vegetation": ("landveg", "vg", 32, 12, 30, np.float64),
}


def build_case(name):
    config, hydraulics, Nz, ncol, nsteps, dtype = CASES[name]
    lat, lon = W.columns_from_mask("N145" if "n145" in name else "N72")
    sel = np.linspace(0, lat.size - 1, ncol).astype(int)
    return W.make_workload(config, lat[sel], lon[sel], Nz, dtype=dtype, hydraulics=hydraulics), nsteps


def main():
    out = {}
    for name in CASES:
        w, nsteps = build_case(name)
        orc = W.setup_oracle(w)
        orc.run(w["dt"], nsteps)
        for f in W.compared_fields(w):
            out[f"{name}/{f}"] = orc.get(f)
        out[f"{name}/status"] = np.array(orc.status())
    np.savez_compressed(os.path.join(HERE, "step_vectors.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
