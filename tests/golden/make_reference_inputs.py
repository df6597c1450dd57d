"""Inputs of the reference-side fixture kit: the synthetic cases of tests/workloads.py (BASELINE configs C1 - C4 on a handful of
columns) written as raw little-endian float64 arrays + a manifest, so that tests/golden/make_reference_fixtures.jl can set up
the SAME cases on Terrarium.jl without re-implementing this repository's seeded generators.  Data only.
    python tests/golden/make_reference_inputs.py      -> tests/golden/reference_inputs/{manifest.json, <case>__<name>.bin}
Array layouts: 3-D fields [Nz][Nh] (row 0 = bottom layer: `interior(field)[:, 1, :]` transposed), per-column arrays [Nh],
thickness [Nz] with index 0 = surface layer (`get_spacing`)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import workloads as W  # noqa: E402

CASES = {
    # name: (config, hydraulics, mask, Nz, columns, steps dumped)
    "c1_single_column_heat": ("heat", "default", "N72", 20, 1, (1, 100)),
    "c2_n72_heat": ("heat", "default", "N72", 30, 16, (1, 100)),
    "c3_n145_richards": ("richards", "default", "N145", 32, 16, (1, 100)),
    "c3_n145_richards_vg": ("richards", "vg", "N145", 32, 16, (1, 100)),
    "c4_n145_land": ("land", "default", "N145", 32, 16, (1, 50)),
    "c4_n145_land_vg": ("land", "vg", "N145", 32, 16, (1, 50)),
}
OUT = os.path.join(HERE, "reference_inputs")


def build_case(name):
    config, hydraulics, mask, Nz, ncol, steps = CASES[name]
    lat, lon = W.columns_from_mask(mask)
    sel = np.linspace(0, lat.size - 1, ncol).astype(int)
    return W.make_workload(config, lat[sel], lon[sel], Nz, hydraulics=hydraulics), steps


def main():
    os.makedirs(OUT, exist_ok=True)
    manifest = {}
    for name in CASES:
        w, steps = build_case(name)
        entry = dict(config=w["config"], Nz=int(w["Nz"]), Nh=int(w["Nh"]), dt=float(w["dt"]), steps=list(steps), params={k: (int(v) if isinstance(v, (int, np.integer)) else float(v)) for k, v in w["params"].items()},
                     dx=1.0 / w["Nh"], fields={}, bcs={}, inputs={}, compared=W.compared_fields(w))
        def dump(key, arr):
            a = np.ascontiguousarray(np.broadcast_to(np.asarray(arr, dtype=np.float64), arr.shape if hasattr(arr, "shape") else ()), dtype="<f8")
            fn = f"{name}__{key}.bin"
            a.tofile(os.path.join(OUT, fn))
            return dict(file=fn, shape=list(a.shape))
        entry["thickness"] = dump("thickness", np.asarray(w["thickness"]))
        for k, v in w["fields"].items():
            entry["fields"][k] = dump("field_" + k, np.asarray(v, dtype=np.float64))
        for (var, side), (kind, v) in w["bcs"].items():
            entry["bcs"][f"{var}:{side}"] = dict(kind=kind, **dump(f"bc_{var}_{side}", np.broadcast_to(np.asarray(v, dtype=np.float64), (w["Nh"],)).copy()))
        for k, v in w["inputs"].items():
            entry["inputs"][k] = dump("input_" + k, np.broadcast_to(np.asarray(v, dtype=np.float64), (w["Nh"],)).copy())
        manifest[name] = entry
    json.dump(manifest, open(os.path.join(OUT, "manifest.json"), "w"), indent=1, sort_keys=True)
    print("wrote", len(manifest), "cases to", OUT)


if __name__ == "__main__":
    main()
