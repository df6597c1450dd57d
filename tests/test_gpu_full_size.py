"""Full-size checks at BASELINE.json's configurations (N145 mask x 32 levels; 0.1-degree-sized shard in
fp32): size-independent properties plus the oracle on a sample of columns -- columns are laterally
independent, so every sampled column of the full-size GPU run must equal the oracle run on that
column alone."""
import numpy as np
import pytest

import workloads as W
import terrarium_jl_amd as trm

pytestmark = pytest.mark.gpu


def sample_workload(w, sel):
    ws = dict(w)
    ws["Nh"] = sel.size
    ws["fields"] = {k: (v[..., sel] if np.ndim(v) else v) for k, v in w["fields"].items()}
    ws["bcs"] = {k: (kind, (val[sel] if np.ndim(val) else val)) for k, (kind, val) in w["bcs"].items()}
    ws["inputs"] = {k: (v[sel] if np.ndim(v) else v) for k, v in w["inputs"].items()}
    return ws


@pytest.mark.parametrize("config,hydraulics,exact", [("richards", "default", True), ("heat", "default", True),
                                                      ("land", "vg", False), ("land", "default", False)])
def test_n145_full_size_against_sampled_oracle(config, hydraulics, exact):
    lat, lon = W.columns_from_mask("N145")
    assert lat.size == 56951
    w = W.make_workload(config, lat, lon, 32, hydraulics=hydraulics)
    nsteps = 40
    dev = W.setup_device(w)
    dev.step(w["dt"], nsteps, finalize=True)
    assert dev.status() == 0
    rng = np.random.default_rng(11)
    sel = np.unique(np.concatenate([[0, 1, 2, 31, 32, 63, 64, 65, lat.size - 2, lat.size - 1],
                                    rng.integers(0, lat.size, 400)]))
    orc = W.setup_oracle(sample_workload(w, sel))
    # dx of the sampled oracle grid must be the full grid's (it enters the flux-BC scaling Az / V)
    orc = _oracle_with_dx(sample_workload(w, sel), 1.0 / lat.size)
    orc.run(w["dt"], nsteps)
    for name in W.compared_fields(w):
        a = dev.get(name)[..., sel]
        b = orc.get(name)
        assert np.all(np.isfinite(a)), name
        if exact:
            assert np.array_equal(a, b), name
        else:
            assert np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))) < 1e-10, name


def _oracle_with_dx(ws, dx):
    import oracle
    p = oracle.default_params(**ws["params"])
    o = oracle.Oracle(ws["Nh"], ws["thickness"], p, dtype=ws["dtype"], dx=dx)
    for name, v in ws["fields"].items():
        o.set(name, v)
    for (var, side), (kind, value) in ws["bcs"].items():
        o.set_bc(var, side, kind, value)
    for name, v in ws["inputs"].items():
        o.set(name, v)
    o.initialize()
    return o


def test_n145_fused_equals_unfused_and_conserves_water():
    lat, lon = W.columns_from_mask("N145")
    w = W.make_workload("richards", lat, lon, 32)
    a, b = W.setup_device(w), W.setup_device(w)
    b.set_option("step_kernel", "unfused")
    water = lambda d: d.reduce("saturation_water_ice", "volume_integral_z")[0] * 0.49 + d.reduce("surface_excess_water", "sum")[0]
    m0 = water(a)
    a.step(w["dt"], 30, finalize=True)
    b.step(w["dt"], 30, finalize=True)
    for n in W.compared_fields(w):
        assert np.array_equal(a.get(n), b.get(n)), n
    # no-flux top and bottom: total water (pore water + surface excess) is conserved by the step
    assert water(a) == pytest.approx(m0, rel=1e-12)
    sat = a.saturation_water_ice
    assert sat.min() >= 0.0 and sat.max() <= 1.0
    # permutation equivariance: columns are independent, so reversing their order reverses the result
    wr = W.make_workload("richards", lat[::-1].copy(), lon[::-1].copy(), 32)
    wr["fields"] = {k: np.ascontiguousarray(v[..., ::-1]) for k, v in w["fields"].items()}
    wr["bcs"] = {k: (kind, np.ascontiguousarray(val[::-1])) for k, (kind, val) in w["bcs"].items()}
    r = W.setup_device(wr)
    r.step(w["dt"], 30, finalize=True)
    for n in ("temperature", "saturation_water_ice", "water_table"):
        assert np.array_equal(r.get(n)[..., ::-1], a.get(n)), n


@pytest.mark.parametrize("hydraulics,num_columns", [("vg", 203125), ("default", 203125), ("default", 812500)])
def test_c5_shard_fp32_properties(hydraulics, num_columns):
    """0.1-degree-sized shard (fp32, 64 levels): 812 500 columns = one GPU's full C5 share (BASELINE config 5:
    ~6.5 M columns over 8 GPUs), and an odd quarter of it (203 125).  With the default hydraulics the step runs in the
    packed two-columns-per-lane kernel: it must equal the scalar kernel bit for bit at this size too."""
    lat, lon = W.synthetic_columns(num_columns)
    w = W.make_workload("land", lat, lon, 64, dtype=np.float32, hydraulics=hydraulics)
    dev = W.setup_device(w)
    dev.step(w["dt"], 10, finalize=True)
    assert dev.status() == 0
    if hydraulics == "default":
        ref = W.setup_device(w)
        ref.set_option("packed_f32", 0)
        ref.step(w["dt"], 10, finalize=True)
        for n in W.compared_fields(w):
            assert np.array_equal(dev.get(n), ref.get(n), equal_nan=True), n
        ref.close()
        del ref
    for n in ("temperature", "saturation_water_ice", "pressure_head", "skin_temperature", "ground_heat_flux"):
        assert np.all(np.isfinite(dev.get(n))), n
    sel = np.unique(np.concatenate([[0, 1, 63, 64, 65, num_columns - 2, num_columns - 1], np.arange(0, num_columns, num_columns // 100)]))
    orc = _oracle_with_dx(sample_workload(w, sel), 1.0 / lat.size)
    orc.run(w["dt"], 10)
    for n in ("temperature", "saturation_water_ice", "internal_energy", "skin_temperature", "pressure_head", "ground_heat_flux"):
        a, b = dev.get(n)[..., sel].astype(np.float64), orc.get(n).astype(np.float64)
        assert np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))) < 1e-4, n
    dev.close()


def test_bench_contract_json_line():
    """bench.py prints ONE JSON line with the driver's keys, the roofline object and the CPU baseline."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "20", "--warmup", "3", "--cpu-seconds", "2", "--spinup-ms", "20"],
                         capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["metric"] == "column-steps/sec" and d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 3
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "f64" and d["scaling"] == "weak"
    assert "N145" in d["config"]["workload"] and d["config"]["status_flags"] == 0
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.2 < r["frac"] < 1.0
    assert r["traffic"] is None or r["traffic"] > 1e8
    h = d["roofline_hbm_resident"]      # the same kernel on a state 3.6x the Infinity Cache, and the C5 shard, in the same run
    assert h["columns_per_launch"] == 8 * 56951 and h["state_bytes_per_launch"] > 2 * 256 * 2**20 and 0.2 < h["frac"] < 1.0
    assert h["also"][0]["columns_per_launch"] == 812500 and 0.2 < h["also"][0]["frac"] < 1.0
    m = d["multistep"]                  # temporal blocking, reported separately from the per-step roofline
    assert m["steps_per_launch"] == 50 and m["status_flags"] == 0 and m["column_steps_per_s"] > d["value"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 1e5 and "sample" in c
    assert c["streamed_GBps"] > 1.0 and "julia" in c["julia_probe"] and len(c["thread_scan"]) >= 1
    # statistics of the line: median over repeated timed regions, min / max beside it
    assert d["config"]["repeats"] == 10 and d["config"]["ms_per_step_min"] <= d["ms_per_step"] <= d["config"]["ms_per_step_max"]
    assert d["config"]["steps_per_launch"] == 1 and r["kernel_ms_min"] <= r["kernel_ms"] <= d["ms_per_step"] * 1.02
    # value = whole-job columns x steps / wall time
    assert abs(d["value"] - 56951 * 20 / (d["ms_per_step"] * 1e-3 * 20)) / d["value"] < 1e-6
