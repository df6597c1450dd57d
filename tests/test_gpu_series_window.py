"""Windowed time series (input_sources.jl:142-171, TerrariumRastersExt.jl:96-121): a record streamed through a
fixed device window by trm_series_append / trm_series_trim_before -- the H2D copies on a side stream -- must give the
all-resident series' results bit for bit: Euler, Heun, the multi-step program, every time rule that can be windowed."""
import numpy as np
import pytest

import workloads as W
import terrarium_jl_amd as trm

pytestmark = pytest.mark.gpu


def small_columns(n):
    lat, lon = W.columns_from_mask("N72")
    sel = np.linspace(0, lat.size - 1, n).astype(int)
    return lat[sel], lon[sel]


def _record(w, nt, dt_nodes, seed):
    rng = np.random.default_rng(seed)
    t = dt_nodes * np.arange(nt) + np.cumsum(rng.uniform(0.0, 0.2 * dt_nodes, nt))      # irregular nodes
    ph = 2 * np.pi * t[:, None] / 86400.0 - w["lon"][None, :]
    return t, ph


@pytest.mark.parametrize("mode", ["euler", "heun", "multistep"])
@pytest.mark.parametrize("config,dtype", [("land", np.float64), ("heat", np.float32)])
def test_200_levels_through_a_16_level_window(config, dtype, mode):
    lat, lon = small_columns(77)
    w = W.make_workload(config, lat, lon, 24, dtype=dtype)
    nt, W_LEVELS, steps, chunk = 200, 16, 640, 13
    t, ph = _record(w, nt, 3.1 * w["dt"], 3)                    # ~3 steps per level: 640 steps cross every level of the record
    if config == "land":
        series = [("air_temperature", "linear", w["T0"][None, :] + 5.0 * np.sin(ph)), ("surface_shortwave_down", "raster", np.maximum(0.0, 600.0 * np.sin(ph))),
                  ("rainfall", "clamp", 1.0e-7 * (1 + np.cos(ph)))]
        steps = 60                                               # (the explicit Richards scheme of this synthetic state stays finite)
        t = t * (60.0 / 640.0)
    else:
        series = [(("temperature", "top"), "linear", w["T0"][None, :] + 10.0 * np.sin(ph)), (("internal_energy", "bottom"), "clamp", 0.05 + 0.02 * np.cos(ph))]
    a, b = W.setup_device(w), W.setup_device(w)
    for d in (a, b):
        d.set_option("steps_per_launch", 5 if mode == "multistep" else 1)
    def attach(d, target, rule, vals, n):
        if isinstance(target, tuple):
            d.set_bc_series(target[0], target[1], "value" if target[0] == "temperature" else "flux", t[:n], vals[:n], rule)
        else:
            d.set_forcing_series(target, t[:n], vals[:n], rule)
    nxt = {}
    for target, rule, vals in series:
        attach(a, target, rule, vals, nt)
        attach(b, target, rule, vals, W_LEVELS)
        b.series_window(target, W_LEVELS)          # (a resident series is never trimmed: the window is declared)
        nxt[target] = W_LEVELS
    step = (lambda d, n, fin: d.step_heun(w["dt"], n, finalize=fin)) if mode == "heun" else (lambda d, n, fin: d.step(w["dt"], n, finalize=fin))
    step(a, steps, True)
    done = 0
    while done < steps:
        tnow = b.clock()[0]
        b.series_trim_before(tnow)
        cover = np.inf
        for target, rule, vals in series:
            info = b.series_info(target)
            room = W_LEVELS - info["levels"]
            if room > 0 and nxt[target] < nt:
                hi = min(nt, nxt[target] + room)
                b.series_append(target, t[nxt[target]:hi], vals[nxt[target]:hi])
                nxt[target] = hi
                info = b.series_info(target)
            assert info["capacity"] == W_LEVELS and info["levels"] <= W_LEVELS          # the ring never grows
            if nxt[target] < nt:
                cover = min(cover, info["t_last"])
        k = min(chunk, steps - done, int(np.floor((cover - tnow) / w["dt"] + 1e-9)) if np.isfinite(cover) else steps)
        assert k >= 1
        step(b, k, done + k == steps)
        done += k
    assert all(n == nt for n in nxt.values()) or config == "land"
    assert a.clock() == b.clock() and a.status() == b.status()
    for n in W.compared_fields(w) + ["tend_internal_energy"] + [s[0] for s in series if not isinstance(s[0], tuple)]:
        assert np.array_equal(a.get(n), b.get(n), equal_nan=True), n


def test_windowed_field_time_series_through_the_host_mirror():
    """FieldTimeSeries(...).windowed(n) as input / boundary value of trm.initialize: run! and Simulation feed the device
    window between library calls; same numbers as the resident series, and the oracle's."""
    import oracle
    grid = trm.ColumnGrid(trm.ExponentialSpacing(N=20), 60)
    rng = np.random.default_rng(8)
    T0 = rng.uniform(-5.0, 15.0, 60)
    times = 900.0 * np.arange(40)
    vals = T0[None, :] + 8.0 * np.sin(2 * np.pi * times[:, None] / 86400.0 + rng.uniform(0, 6, 60)[None, :])
    def make(window):
        fts = trm.FieldTimeSeries(times, vals, "linear")
        bc = trm.PrescribedSurfaceTemperature("Ts", fts.windowed(window) if window else fts)
        return trm.initialize(trm.SoilModel(grid), trm.Heun(dt=300.0), boundary_conditions=bc, initializers=dict(temperature=T0[None, :] * np.ones((20, 1)), saturation_water_ice=0.8))
    a, b = make(None), make(6)
    trm.run(a, steps=110)
    trm.run(b, steps=110)
    assert b.state.series_info(("temperature", "top"))["capacity"] == 6
    o = oracle.Oracle(60, grid.thickness, oracle.default_params())
    o.set("temperature", T0[None, :] * np.ones((20, 1))); o.set("saturation_water_ice", 0.8)
    o.set_bc_series("temperature", "top", "value", times, vals, "linear")
    o.initialize()
    for _ in range(110):
        o.timestep_heun(300.0, False)
    for n in ("internal_energy", "temperature", "liquid_water_fraction"):
        assert np.array_equal(a.state.get(n), b.state.get(n)), n
        assert np.array_equal(b.state.get(n), o.get(n)), n


def test_append_argument_errors_and_growth():
    lat, lon = small_columns(10)
    w = W.make_workload("heat", lat, lon, 20)
    d = W.setup_device(w)
    t = 100.0 * np.arange(4)
    v = np.ones((4, 10))
    with pytest.raises(trm.TerrariumHipError):
        d.series_append("air_temperature", t, v)                       # no such series yet
    d.set_bc_series("temperature", "top", "value", t, v, "linear")
    with pytest.raises(trm.TerrariumHipError):
        d.series_append(("temperature", "top"), t, v)                  # does not continue the series
    d.series_append(("temperature", "top"), t + 400.0, 2 * v)          # nothing trimmed: the ring grows
    info = d.series_info(("temperature", "top"))
    assert info == dict(levels=8, capacity=8, t_first=0.0, t_last=700.0)
    d.set_clock(650.0, 0)
    d.update_inputs()
    d.series_trim_before(650.0)
    assert d.series_info(("temperature", "top"))["levels"] == 2 and d.series_info(("temperature", "top"))["t_first"] == 600.0
    # a fully resident series in the same context keeps its record, and the windowed one refuses a clock before its head
    d.set_forcing_series("air_temperature", t, v, "linear")
    d.series_trim_before(650.0)
    assert d.series_info("air_temperature")["levels"] == 4
    d.set_clock(100.0, 0)
    with pytest.raises(trm.TerrariumHipError, match="before the levels it still holds"):
        d.update_inputs()
    with pytest.raises(trm.TerrariumHipError, match="before the levels it still holds"):
        d.step(w["dt"], 1)
    d.set_clock(650.0, 0)
    d.step(w["dt"], 1)
    d.set_bc_series("internal_energy", "bottom", "flux", t, v, "cyclical")
    with pytest.raises(trm.TerrariumHipError):
        d.series_append(("internal_energy", "bottom"), t + 400.0, v)   # cyclical series cannot be windowed
