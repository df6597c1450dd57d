"""Time series input sources on the device (trm_set_forcing_series / trm_set_bc_series) against the oracle's
restatement of FieldTimeSeriesInputSource + update_inputs! (src/input_output/input_sources.jl:142-171): one
trm_step call spans many forcing intervals, the series is evaluated at the clock every step."""
import numpy as np
import pytest

import workloads as W
import terrarium_jl_amd as trm
from test_gpu_parity import assert_fields_match, small_columns, TOL64, TOL32

pytestmark = pytest.mark.gpu
DAY = 86400.0


def forcing_series(w, times):
    """SURVEY 8(d) forcing, sampled at `times` (diurnal cycle), [nt][Nh] per input."""
    lon, T0, u = w["lon"], w["T0"], w["u"]
    ph = 2 * np.pi * np.asarray(times)[:, None] / DAY - lon[None, :]
    return dict(air_temperature=T0[None, :] + 5.0 * np.sin(ph),
                surface_shortwave_down=np.maximum(0.0, 600.0 * np.sin(ph)),
                windspeed=(1.0 + 2.0 * np.abs(u))[None, :] * (1.0 + 0.3 * np.cos(ph)),
                rainfall=1.0e-8 * (u > 0.5)[None, :] * (np.sin(ph) > 0))


@pytest.mark.parametrize("kernel", ["fused", "unfused"])
def test_heat_with_upper_boundary_series(kernel):
    """The reference's periodic upper boundary (soil_heat_global.jl:72-89 style) as a Value series on temperature:
    40 steps of 300 s in ONE call across 600 s nodes.  fp64 heat-only => bit-exact."""
    lat, lon = small_columns(130)
    w = W.make_workload("heat", lat, lon, 20)
    times = np.arange(0.0, 40 * 300.0 + 1, 600.0)
    vals = w["T0"][None, :] + 10.0 * np.sin(2 * np.pi * times[:, None] / DAY - lon[None, :])
    orc, dev = W.setup_oracle(w), W.setup_device(w)
    dev.set_option("step_kernel", kernel)
    for o in (orc, dev):
        o.set_bc_series("temperature", "top", "value", times, vals)
    orc.run(w["dt"], 40)
    dev.step(w["dt"], 40, True)
    assert_fields_match(dev, orc, W.compared_fields(w), True, 0.0, "bc series ")
    # the series was really used: the boundary moved away from its t = 0 value
    w0 = W.make_workload("heat", lat, lon, 20)
    ref0 = W.setup_oracle(w0)
    ref0.run(w["dt"], 40)
    assert not np.array_equal(ref0.get("temperature"), orc.get("temperature"))


@pytest.mark.parametrize("hydraulics,dtype,tol", [("default", np.float64, TOL64), ("vg", np.float64, TOL64), ("default", np.float32, TOL32)])
def test_land_with_forcing_series(hydraulics, dtype, tol):
    """LandModel driven by four forcing series with different node spacings and time indexing modes, 30 steps of 60 s
    in one call; the other inputs stay constant."""
    lat, lon = small_columns(97)
    w = W.make_workload("land", lat, lon, 32 if dtype == np.float64 else 64, dtype=dtype, hydraulics=hydraulics)
    orc, dev = W.setup_oracle(w), W.setup_device(w)
    modes = dict(air_temperature="linear", surface_shortwave_down="clamp", windspeed="cyclical", rainfall="linear")
    nodes = dict(air_temperature=np.arange(0.0, 2401.0, 400.0), surface_shortwave_down=np.array([100.0, 450.0, 900.0, 1200.0]),
                 windspeed=np.arange(0.0, 500.0, 90.0), rainfall=np.array([0.0, 3600.0]))
    for name, mode in modes.items():
        vals = forcing_series(w, nodes[name])[name]
        for o in (orc, dev):
            o.set_forcing_series(name, nodes[name], vals, mode)
    orc.run(w["dt"], 30)
    dev.step(w["dt"], 30, True)
    assert_fields_match(dev, orc, W.compared_fields(w) + ["air_temperature", "windspeed", "surface_shortwave_down"], False, tol, "forcing series ")


def test_series_equals_host_interpolation():
    """Independent of the oracle: stepping with a device series == stepping one step at a time with the inputs
    interpolated on the host by v2 * f + v1 * (1 - f) (the formula in include/terrarium_hip.h)."""
    lat, lon = small_columns(64)
    w = W.make_workload("land", lat, lon, 32)
    a, b = W.setup_device(w), W.setup_device(w)
    times = np.array([0.0, 250.0, 700.0, 1800.0])
    vals = forcing_series(w, times)["air_temperature"]
    a.set_forcing_series("air_temperature", times, vals)
    a.step(w["dt"], 20, True)
    for n in range(20):
        t = n * w["dt"]
        k = min(np.searchsorted(times, t, side="right") - 1, times.size - 2)
        if t in times[1:-1]:
            v = vals[list(times).index(t)]
        else:
            f = (1.0 / (times[k + 1] - times[k])) * (t - times[k])
            v = vals[k + 1] * f + vals[k] * (1.0 - f)
        b.set_forcing("air_temperature", v)
        b.step(w["dt"], 1, n == 19)
    for name in W.compared_fields(w):
        assert np.array_equal(a.get(name), b.get(name), equal_nan=True), name


@pytest.mark.parametrize("config", ["heat", "land"])
def test_heun_with_series(config):
    """Heun evaluates the stage's inputs at t + dt (heun.jl:37-71: the stage's clock has ticked)."""
    lat, lon = small_columns(40)
    w = W.make_workload(config, lat, lon, 20)
    orc, dev = W.setup_oracle(w), W.setup_device(w)
    times = np.arange(0.0, 3001.0, 500.0)
    for o in (orc, dev):
        if config == "heat":
            vals = w["T0"][None, :] + 10.0 * np.sin(2 * np.pi * times[:, None] / 3000.0 - lon[None, :])
            o.set_bc_series("temperature", "top", "value", times, vals)
        else:
            o.set_forcing_series("air_temperature", times, forcing_series(w, times * 20)["air_temperature"])
            o.set_forcing_series("surface_shortwave_down", times, forcing_series(w, times * 20)["surface_shortwave_down"], "clamp")
    n = 8
    for k in range(n):   # finalize once, as the library does (LandModel's skin temperature is updated in place)
        orc.timestep_heun(w["dt"], k == n - 1)
    dev.step_heun(w["dt"], n, True)
    assert_fields_match(dev, orc, W.compared_fields(w), config == "heat", TOL64, "heun series ")


def test_series_argument_checks():
    lat, lon = small_columns(8)
    w = W.make_workload("land", lat, lon, 8)
    dev = W.setup_device(w)
    ok = np.zeros((3, 8))
    with pytest.raises(trm.TerrariumHipError):
        dev.set_forcing_series("air_temperature", [0.0, 2.0, 1.0], ok)          # not increasing
    with pytest.raises(trm.TerrariumHipError):
        dev.set_forcing_series("temperature", [0.0, 1.0, 2.0], ok)              # not an input field
    with pytest.raises(trm.TerrariumHipError):
        dev.set_bc_series("temperature", "top", "noflux", [0.0, 1.0, 2.0], ok)  # a series needs a valued kind
    dev.set_forcing_series("air_temperature", [5.0], np.full((1, 8), 3.5))      # a single node = a constant
    dev.update_inputs()
    assert np.all(dev.get("air_temperature") == 3.5)
    dev.clear_series()
    dev.set_forcing("air_temperature", 1.25)
    dev.step(w["dt"], 2, True)
    assert np.all(dev.get("air_temperature") == 1.25)


def test_integrator_with_field_time_series():
    """Host mirror: a FieldTimeSeries given as a boundary value and as an InputSource keeps run! a single library
    call and matches the per-step host evaluation of the same piecewise-linear function."""
    grid = trm.ColumnGrid(trm.ExponentialSpacing(N=20), 5)
    times = np.arange(0.0, 7201.0, 900.0)
    T_ub = lambda t: 2.0 + 8.0 * np.interp(t, times, np.sin(2 * np.pi * times / 7200.0))
    def make(value):
        bcs = trm.merge_boundary_conditions(trm.PrescribedSurfaceTemperature("T_ub", value))
        return trm.initialize(trm.SoilModel(grid), trm.ForwardEuler(), boundary_conditions=bcs,
                              initializers=dict(temperature=1.0, saturation_water_ice=1.0))
    a = make(trm.FieldTimeSeries.from_function(T_ub, times))
    b = make(T_ub)
    assert not a._has_time_dependence() and b._has_time_dependence()
    trm.run(a, steps=24, dt=300.0)
    trm.run(b, steps=24, dt=300.0)
    assert a.clock == b.clock
    # np.interp and the device formula round differently in the last place; the trajectories agree to 1e-12
    assert np.allclose(a.state.get("temperature"), b.state.get("temperature"), rtol=0, atol=1e-11)
    assert np.ptp(a.state.get("temperature")[-1]) == 0 and abs(a.state.get("temperature")[-1, 0] - 1.0) > 0.1


@pytest.mark.parametrize("kernel", ["fused", "unfused"])
def test_heun_with_time_dependent_boundary_function(kernel):
    """under Heun a functional boundary value is evaluated at t for the state and at t + dt for the stage
    (heun.jl:52-59).  The integrator hands f over as a two-node series per step; the oracle steps the same function
    sampled on the step grid (every evaluation hits a node)."""
    import oracle
    Nh, Nz, dt, nsteps = 40, 20, 300.0, 12
    lon = np.linspace(0, 2 * np.pi, Nh, endpoint=False)
    f = lambda t: 2.0 + 8.0 * np.sin(2 * np.pi * t / 7200.0 - lon)
    grid = trm.ColumnGrid(trm.ExponentialSpacing(N=Nz), Nh)
    integ = trm.initialize(trm.SoilModel(grid), trm.Heun(dt=dt), boundary_conditions=trm.PrescribedSurfaceTemperature("Ts", f),
                           initializers=dict(temperature=1.0, saturation_water_ice=0.7))
    integ.state.set_option("step_kernel", kernel)
    for _ in range(nsteps):
        trm.timestep(integ)
    orc = oracle.Oracle(Nh, grid.thickness, oracle.default_params())
    orc.set("temperature", 1.0)
    orc.set("saturation_water_ice", 0.7)
    nodes = dt * np.arange(nsteps + 2)
    orc.set_bc_series("temperature", "top", "value", nodes, np.stack([f(t) for t in nodes]))
    orc.update_inputs()
    orc.initialize()
    for _ in range(nsteps):
        orc.timestep_heun(dt, True)
    for name in ("temperature", "internal_energy", "liquid_water_fraction"):
        assert np.array_equal(integ.state.get(name), orc.get(name)), name
    # and it differs from holding f(t) over both stages (what round 1 did)
    held = trm.initialize(trm.SoilModel(grid), trm.Heun(dt=dt), boundary_conditions={("temperature", "top"): ("value", f(0.0))},
                          initializers=dict(temperature=1.0, saturation_water_ice=0.7))
    for n in range(nsteps):
        held.state.set_bc("temperature", "top", "value", f(n * dt))
        held.state.step_heun(dt, 1, True)
    assert not np.array_equal(held.state.get("temperature"), integ.state.get("temperature"))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_raster_input_source_on_the_device(dtype):
    """RasterInputSource (ext/TerrariumRastersExt): a time-indexed raster on the full grid, gathered through the mask's
    index map and interpolated on the device by every step with the extension's rule (nodes exactly, linear in between,
    flat beyond the ends) -- device == oracle == the rule evaluated in numpy."""
    import oracle
    from terrarium_jl_amd.io import RasterInputSource
    rng = np.random.default_rng(8)
    mask = rng.random((12, 20)) > 0.5
    grid = trm.ColumnRingGrid(trm.ExponentialSpacing(N=10), mask, dtype=dtype)
    Nh = grid.num_columns
    times = 3600.0 * np.arange(4)
    Tair = (5.0 + 10.0 * rng.random((4, 12, 20))).astype(dtype)
    sw = (400.0 * rng.random((4, 12, 20))).astype(dtype)
    land = trm.LandModel(grid)
    inits = dict(temperature=2.0, saturation_water_ice=0.6)
    integ = trm.initialize(land, trm.ForwardEuler(dt=450.0), initializers=inits,
                           inputs=trm.InputSources(trm.InputSource(RasterInputSource(grid, Tair, "air_temperature", times=times)),
                                                   trm.InputSource(RasterInputSource(grid, sw[0], "surface_shortwave_down")),
                                                   trm.InputSource(RasterInputSource(grid, sw, "surface_longwave_down", times=times, reftime=-1800.0))))
    st = integ.state
    cols = grid.gather(Tair)
    lw = grid.gather(sw)

    def rule(t, tt, x):
        if t <= tt[0]: return x[0]
        if t >= tt[-1]: return x[-1]
        r = int(np.searchsorted(tt, t, side="left")); l = int(np.searchsorted(tt, t, side="right")) - 1
        if l == r: return x[r]
        return (x[l].astype(np.float64) + (t - tt[l]) * (x[r] - x[l]).astype(np.float64) / (tt[r] - tt[l])).astype(dtype)

    assert np.array_equal(st.air_temperature, cols[0]) and np.array_equal(st.surface_shortwave_down, grid.gather(sw[0]))
    o = oracle.Oracle(Nh, grid.thickness, oracle.default_params(flow=0, seb=1), dtype=dtype, dx=grid.dx)
    o.set("temperature", 2.0); o.set("saturation_water_ice", 0.6)
    o.set_forcing_series("air_temperature", times, cols, "raster")
    o.set_forcing_series("surface_longwave_down", times + 1800.0, lw, "raster")
    o.set("surface_shortwave_down", grid.gather(sw[0]))
    o.update_inputs(); o.initialize()
    for n in range(30):      # 30 x 450 s: past the last node
        trm.timestep(integ)
        o.timestep(450.0, True)
        t_eval = n * 450.0   # inputs are evaluated at the pre-tick time
        assert np.array_equal(st.air_temperature, rule(t_eval, times, cols)), n
        assert np.array_equal(st.surface_longwave_down, rule(t_eval, times + 1800.0, lw)), n
    tol = 1e-10 if dtype == np.float64 else 1e-4
    for name in ("temperature", "skin_temperature", "ground_heat_flux", "air_temperature"):
        a, b = st.get(name).astype(np.float64), o.get(name).astype(np.float64)
        assert np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))) <= tol, name
