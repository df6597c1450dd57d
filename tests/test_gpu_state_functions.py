"""State-dependent user forcings and boundary values (src/forcings.jl:13-19 `forcing(i, j, k, grid, clock, fields)`,
src/boundary_conditions.jl:25-28 `getbc(..., clock, fields)`): `trm.StateFunction` evaluates the user's function on the
device -- zero-copy torch views of the library's buffers, on the library's stream -- before every step.  Checked against the
ORACLE stepped by hand with the same function evaluated in numpy on the oracle's fields."""
import numpy as np
import pytest

import terrarium_jl_amd as trm

pytestmark = pytest.mark.gpu


def oracle_like(integ, land=False):
    """An oracle with the integrator's flattened parameters and grid."""
    import oracle
    p, grid = integ.state.params, integ.state.grid
    names = [n for n, _ in oracle.ParamsD._fields_ if hasattr(p, n)]
    o = oracle.Oracle(grid.Nh, grid.thickness, oracle.default_params(**{n: getattr(p, n) for n in names}), dx=grid.dx)
    if land:
        o.set_land_model(True)
    return o


def columns(n):
    rng = np.random.default_rng(11)
    return rng.uniform(-1.0, 1.0, n)


def stepper_of(name, dt):
    return trm.Heun(dt=dt) if name == "heun" else trm.ForwardEuler(dt=dt)


def oracle_step(o, heun, dt, finalize, evaluate):
    """One step of the oracle with `evaluate(oracle)` where the reference's tendency kernels would evaluate the user's function:
    at the state, and under Heun at the stage as well (heun.jl:37-71: its clock has ticked)."""
    if heun:
        o.timestep_heun_by_hand(dt, finalize, at_state=evaluate, at_stage=evaluate)
    else:
        evaluate(o)
        o.timestep(dt, finalize)


@pytest.mark.parametrize("stepper", ["euler", "heun"])
def test_relaxation_forcing_of_the_water_content_follows_the_state(stepper):
    """vwc_forcing = -(sat - target) * rate per cell: a nudging term, the textbook state-dependent Forcing.  Under Heun the
    function is evaluated at both stages (the stage's own per-cell forcing buffer: trm_stage_field_device_ptr)."""
    Nz, Nh, dt = 24, 130, 60.0
    u = columns(Nh)
    grid = trm.ColumnGrid(trm.ExponentialSpacing(N=Nz), Nh)
    zc = grid.z_centers()
    par = dict(target=0.6, rate=1.0 / 7200.0)
    F = trm.StateFunction(lambda f, clock, p: (f.saturation_water_ice - p["target"]) * (-p["rate"]), parameters=par)
    model = trm.SoilModel(grid, soil=trm.SoilEnergyWaterCarbon(hydrology=trm.SoilHydrology(vertical_flow=trm.RichardsEq(), vwc_forcing=F)))
    T_init = (3.0 + 2.0 * u)[None, :] - 0.05 * zc[:, None]
    sat = np.clip(np.minimum(1.0, 0.8 - 0.05 * zc)[:, None] * (1.0 + 0.1 * u)[None, :], 0.05, 1.0)
    integ = trm.initialize(model, stepper_of(stepper, dt), boundary_conditions=trm.PrescribedSurfaceTemperature("Ts", 5.0 + u),
                           initializers=dict(temperature=T_init, saturation_water_ice=sat))
    o = oracle_like(integ)
    o.set("temperature", T_init)
    o.set("saturation_water_ice", sat)
    o.set_bc("temperature", "top", "value", 5.0 + u)
    o.initialize()
    trm.run(integ, steps=25)
    for n in range(25):
        oracle_step(o, stepper == "heun", dt, n == 24, lambda q: q.set("vwc_forcing", (q.get("saturation_water_ice") - par["target"]) * (-par["rate"])))
    assert integ.state.clock() == (25 * dt, 25) and integ.state.status() == 0
    for name in ("saturation_water_ice", "internal_energy", "temperature", "pressure_head", "liquid_water_fraction"):
        assert np.array_equal(integ.state.get(name), o.get(name)), name
    # the forcing did something: without it the column mean stays where the fluxes put it
    plain = trm.initialize(trm.SoilModel(grid, soil=trm.SoilEnergyWaterCarbon(hydrology=trm.SoilHydrology(vertical_flow=trm.RichardsEq()))), stepper_of(stepper, dt),
                           boundary_conditions=trm.PrescribedSurfaceTemperature("Ts", 5.0 + u), initializers=dict(temperature=T_init, saturation_water_ice=sat))
    trm.run(plain, steps=25)
    assert np.max(np.abs(plain.state.saturation_water_ice - integ.state.saturation_water_ice)) > 1e-4


@pytest.mark.parametrize("stepper", ["euler", "heun"])
def test_boundary_value_that_reads_the_top_cell(stepper):
    """top temperature value = 0.5 (T_top + 10): a boundary condition in discrete form reading `fields`; timestep! and run!
    Under Heun the stage's boundary values are its own array (trm_stage_bc_device_ptr), evaluated from the stage's temperature."""
    Nz, Nh, dt = 16, 77, 300.0
    u = columns(Nh)
    grid = trm.ColumnGrid(trm.ExponentialSpacing(N=Nz), Nh)
    bc = trm.PrescribedSurfaceTemperature("Ts", trm.StateFunction(lambda f, clock, p: 0.5 * (f.temperature[:, -1] + 10.0)))
    T_init = (2.0 + 4.0 * u)[None, :] * np.ones((Nz, 1))
    integ = trm.initialize(trm.SoilModel(grid), stepper_of(stepper, dt), boundary_conditions=bc, initializers=dict(temperature=T_init, saturation_water_ice=0.8))
    o = oracle_like(integ)
    o.set("temperature", T_init)
    o.set("saturation_water_ice", 0.8)
    o.initialize()
    trm.timestep(integ)
    trm.run(integ, steps=11)
    for n in range(12):
        oracle_step(o, stepper == "heun", dt, True, lambda q: q.set_bc("temperature", "top", "value", 0.5 * (q.get("temperature")[-1] + 10.0)))
    for name in ("internal_energy", "temperature", "liquid_water_fraction"):
        assert np.array_equal(integ.state.get(name), o.get(name)), name
    assert np.all(integ.state.temperature[-1] != T_init[-1])


@pytest.mark.parametrize("stepper", ["euler", "heun"])
def test_atmospheric_input_that_follows_the_skin_temperature(stepper):
    """LandModel input air_temperature = skin temperature + 2 K: an input source driven by the land state (the coupling
    direction of speedy_dry_land.jl:45-66, here as a function of `fields`); the clock reaches the function -- under Heun twice
    per step: (t, n) at the state and (t + dt, n + 1) at the stage, whose clock has ticked (heun.jl:52)"""
    Nz, Nh, dt = 20, 90, 60.0
    u = columns(Nh)
    grid = trm.ColumnGrid(trm.ExponentialSpacing(N=Nz), Nh)
    zc = grid.z_centers()
    seen = []

    def air(f, clock, p):
        seen.append((clock.time, clock.iteration))
        return f.skin_temperature + 2.0

    hp = trm.ConstantSoilHydraulics(swrc=trm.VanGenuchten(alpha=2.0, n=2.0), unsat_hydraulic_cond=trm.UnsatKVanGenuchten())
    land = trm.LandModel(grid, soil=trm.SoilEnergyWaterCarbon(hydrology=trm.SoilHydrology(vertical_flow=trm.RichardsEq(), hydraulic_properties=hp)))
    T_init = (5.0 + u)[None, :] - 0.02 * zc[:, None]
    sat = np.clip(np.minimum(1.0, 0.8 - 0.05 * zc)[:, None] * (1.0 + 0.05 * u)[None, :], 0.05, 1.0)
    other = dict(air_pressure=101325.0, windspeed=1.0 + 2.0 * np.abs(u), specific_humidity=2.0e-3, surface_shortwave_down=300.0 + 100.0 * u,
                 surface_longwave_down=300.0, rainfall=1.0e-8 * (u > 0))
    integ = trm.initialize(land, stepper_of(stepper, dt), initializers=dict(temperature=T_init, saturation_water_ice=sat),
                           inputs=dict(air_temperature=trm.StateFunction(air), **other))
    o = oracle_like(integ, land=True)
    o.set("temperature", T_init)
    o.set("saturation_water_ice", sat)
    for k, v in other.items():
        o.set(k, v)
    o.set("air_temperature", 0.0)
    o.initialize()
    assert np.array_equal(integ.state.skin_temperature, o.get("skin_temperature"))
    trm.run(integ, steps=15)
    for n in range(15):
        oracle_step(o, stepper == "heun", dt, n == 14, lambda q: q.set("air_temperature", np.ravel(q.get("skin_temperature")) + 2.0))
    if stepper == "heun":
        assert seen == [x for n in range(15) for x in ((n * dt, n), (n * dt + dt, n + 1))]
    else:
        assert seen == [(n * dt, n) for n in range(15)]
    for name in ("skin_temperature", "internal_energy", "saturation_water_ice", "ground_heat_flux", "sensible_heat_flux"):
        a, b = integ.state.get(name), o.get(name)
        assert np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))) < 1e-10, name


def test_two_call_heun_equals_the_one_call_when_nothing_is_changed_at_the_stage():
    """trm_heun_predict + trm_heun_correct = trm_step_heun bit for bit (every Heun path: the fused one-launch programs and the
    reference-order kernels), with a constant top boundary value, a bottom flux and a boundary series evaluated at t and t + dt;
    a stage boundary array that has been handed out is refreshed from the state's by the one-call form; misuse is refused."""
    import workloads as W
    lat, lon = W.synthetic_columns(150)
    for config, Nz in (("richards", 24), ("land", 40), ("heat", 100)):
        w = W.make_workload(config, lat, lon, Nz)
        a, b, c = W.setup_device(w), W.setup_device(w), W.setup_device(w)
        tt = np.array([0.0, 400.0, 900.0, 2000.0])
        for d in (a, b, c):
            d.set_bc("internal_energy", "bottom", "flux", 0.05 + 0.01 * w["u"])
            if config != "land":
                d.set_bc_series("temperature", "top", "value", tt, w["T0"][None, :] + np.sin(tt / 300.0)[:, None])
        c.set_option("step_kernel", "unfused")
        b.stage_bc_device_array("internal_energy", "bottom")        # (handed out, never written: must not disturb anything)
        for n in range(6):
            a.step_heun(w["dt"], 1, finalize=(n == 5))
            b.heun_predict(w["dt"])
            b.heun_correct(w["dt"], finalize=(n == 5))
            c.step_heun(w["dt"], 1, finalize=(n == 5))
        assert a.clock() == b.clock() == c.clock()
        for name in W.compared_fields(w):
            assert np.array_equal(a.get(name), b.get(name), equal_nan=True), (config, name)
            assert np.array_equal(a.get(name), c.get(name), equal_nan=True), (config, name)
        assert a.status() == b.status() == c.status()
        b.step_heun(w["dt"], 2)          # the one-call form after the two-call form: the stage array follows the state's again
        a.step_heun(w["dt"], 2)
        assert np.array_equal(a.get("internal_energy"), b.get("internal_energy"))
    with pytest.raises(trm.TerrariumHipError, match="trm_heun_predict first"):
        a.heun_correct(w["dt"])
    a.heun_predict(w["dt"])
    with pytest.raises(trm.TerrariumHipError, match="dt differs"):
        a.heun_correct(2 * w["dt"])


def test_a_forcing_that_reads_an_auxiliary_field_sees_the_stages_own_under_heun():
    """The reference evaluates a Forcing inside compute_tendencies!(stage), AFTER compute_auxiliary!(stage) (heun.jl:54,
    state_variables.jl:72-80): the ground heat flux / hydraulic conductivity it reads there are the STAGE's.  A boundary-value
    function is evaluated by fill_halo_regions!, BEFORE compute_auxiliary!(stage): it finds the state's copies (copyto!,
    heun.jl:45).  Both against the oracle stepped by hand in the reference's order (trm_heun_stage_auxiliary)."""
    Nz, Nh, dt = 20, 70, 60.0
    u = columns(Nh)
    grid = trm.ColumnGrid(trm.ExponentialSpacing(N=Nz), Nh)
    zc = grid.z_centers()
    hp = trm.ConstantSoilHydraulics(swrc=trm.VanGenuchten(alpha=2.0, n=2.0), unsat_hydraulic_cond=trm.UnsatKVanGenuchten())

    # a sink of soil water proportional to the ground heat flux of the column and to the conductivity of the cell's lower face
    def sink(f, clock, p):
        return -1.0e-9 * f.ground_heat_flux[:, None] - 1.0e-3 * f.hydraulic_conductivity[:, :Nz]

    land = trm.LandModel(grid, soil=trm.SoilEnergyWaterCarbon(hydrology=trm.SoilHydrology(vertical_flow=trm.RichardsEq(), hydraulic_properties=hp,
                                                                                         vwc_forcing=trm.StateFunction(sink))))
    T_init = (5.0 + u)[None, :] - 0.02 * zc[:, None]
    sat = np.clip(np.minimum(1.0, 0.8 - 0.05 * zc)[:, None] * (1.0 + 0.05 * u)[None, :], 0.05, 1.0)
    inputs = dict(air_temperature=8.0 + 3.0 * u, air_pressure=101325.0, windspeed=1.0 + 2.0 * np.abs(u), specific_humidity=2.0e-3,
                  surface_shortwave_down=300.0 + 100.0 * u, surface_longwave_down=300.0, rainfall=1.0e-8 * (u > 0))
    integ = trm.initialize(land, trm.Heun(dt=dt), initializers=dict(temperature=T_init, saturation_water_ice=sat), inputs=inputs)
    o = oracle_like(integ, land=True)
    o.set("temperature", T_init)
    o.set("saturation_water_ice", sat)
    for k, v in inputs.items():
        o.set(k, v)
    o.set("vwc_forcing", np.zeros((Nz, Nh)))
    o.initialize()

    def sink_np(q):
        q.set("vwc_forcing", -1.0e-9 * np.ravel(q.get("ground_heat_flux"))[None, :] - 1.0e-3 * q.get("hydraulic_conductivity")[:Nz])
    trm.run(integ, steps=10)
    for n in range(10):
        # at the state the function runs before update_state!(state): the auxiliaries it reads are the previous step's (both sides)
        o.timestep_heun_by_hand(dt, n == 9, at_state=sink_np, at_stage_tendencies=sink_np)
    for name in ("saturation_water_ice", "internal_energy", "ground_heat_flux", "hydraulic_conductivity", "skin_temperature"):
        a, b = integ.state.get(name), o.get(name)
        assert np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))) < 1e-10, name
    # ... and the stale variant (the function at the stage evaluated before compute_auxiliary!(stage)) is measurably different
    o2 = oracle_like(integ, land=True)
    o2.set("temperature", T_init); o2.set("saturation_water_ice", sat)
    for k, v in inputs.items():
        o2.set(k, v)
    o2.set("vwc_forcing", np.zeros((Nz, Nh)))
    o2.initialize()
    for n in range(10):
        o2.timestep_heun_by_hand(dt, n == 9, at_state=sink_np, at_stage=sink_np)
    assert np.max(np.abs(o2.get("saturation_water_ice") - o.get("saturation_water_ice"))) > 1e-12


def test_a_boundary_value_that_reads_an_auxiliary_field_sees_the_states_copy_at_the_stage():
    Nz, Nh, dt = 20, 50, 60.0
    u = columns(Nh)
    grid = trm.ColumnGrid(trm.ExponentialSpacing(N=Nz), Nh)
    zc = grid.z_centers()
    top = trm.StateFunction(lambda f, clock, p: 4.0 + 2.0e5 * f.hydraulic_conductivity[:, Nz - 1])      # (K of the top cell's lower face)
    model = trm.SoilModel(grid, soil=trm.SoilEnergyWaterCarbon(hydrology=trm.SoilHydrology(vertical_flow=trm.RichardsEq())))
    T_init = (3.0 + 2.0 * u)[None, :] - 0.05 * zc[:, None]
    sat = np.clip(np.minimum(1.0, 0.8 - 0.05 * zc)[:, None] * (1.0 + 0.1 * u)[None, :], 0.05, 1.0)
    integ = trm.initialize(model, trm.Heun(dt=dt), boundary_conditions=trm.PrescribedSurfaceTemperature("Ts", top),
                           initializers=dict(temperature=T_init, saturation_water_ice=sat))
    o = oracle_like(integ)
    o.set("temperature", T_init)
    o.set("saturation_water_ice", sat)
    o.set_bc("temperature", "top", "value", np.zeros(Nh))
    o.initialize()
    bc = lambda q: q.set_bc("temperature", "top", "value", 4.0 + 2.0e5 * q.get("hydraulic_conductivity")[Nz - 1])
    trm.run(integ, steps=8)
    for n in range(8):
        o.timestep_heun_by_hand(dt, n == 7, at_state=bc, at_stage=bc)       # (the clone carries the state's hydraulic_conductivity)
    for name in ("temperature", "internal_energy", "saturation_water_ice"):
        assert np.array_equal(integ.state.get(name), o.get(name)), name


def test_a_predicted_stage_is_dropped_by_whatever_else_touches_the_context():
    """trm_heun_predict, then a step / an upload / a new clock / a boundary condition / a restore: trm_heun_correct refuses
    (the stage describes another state) instead of averaging its tendencies in; read-only calls in between are fine."""
    import workloads as W
    lat, lon = W.synthetic_columns(60)
    w = W.make_workload("richards", lat, lon, 20)
    d = W.setup_device(w)
    d.step_heun(w["dt"], 1)
    d.save_state()
    spoilers = [lambda: d.step(w["dt"], 1), lambda: d.step_heun(w["dt"], 1), lambda: d.set("temperature", d.get("temperature")),
                lambda: d.set_clock(0.0, 0), lambda: d.set_bc("internal_energy", "bottom", "flux", 0.01), lambda: d.restore_state(),
                lambda: d.compute_auxiliary(), lambda: d.set_option("write_kf_every_step", 1)]
    for spoil in spoilers:
        d.heun_predict(w["dt"])
        spoil()
        with pytest.raises(trm.TerrariumHipError, match="trm_heun_predict first"):
            d.heun_correct(w["dt"])
    ref = W.setup_device(w)
    ref.step_heun(w["dt"], 1)
    ref.set_bc("internal_energy", "bottom", "flux", 0.01)      # (one of the spoilers: a boundary condition is not part of the snapshot)
    d.restore_state()
    d.heun_predict(w["dt"])
    d.get("temperature"); d.status(); d.reduce("temperature", "max"); d.clock(); d.get_rows("temperature", 0, 2)      # read-only
    d.heun_stage_auxiliary()
    d.heun_stage_auxiliary()                                                                                          # (idempotent)
    d.heun_correct(w["dt"])
    ref.step_heun(w["dt"], 1)
    for name in W.compared_fields(w):
        assert np.array_equal(d.get(name), ref.get(name), equal_nan=True), name
    with pytest.raises(trm.TerrariumHipError, match="trm_heun_predict first"):
        d.heun_stage_auxiliary()


def test_library_loaded_before_torch_shares_one_hip_runtime():
    """The host mirror loads the library BEFORE anything imported torch (an example script, a user's session): torch must still
    find the GPU afterwards -- one HIP / HSA runtime in the process (`_capi._share_hip_runtime_with_torch`)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import numpy as np, terrarium_jl_amd as trm\n"
            "assert 'torch' not in sys.modules\n"
            "st = trm.DeviceState(trm.ColumnGrid(trm.ExponentialSpacing(N=8), 5), trm._capi.default_params())\n"
            "st.set('temperature', 3.0)\n"
            "import torch\n"
            "assert torch.cuda.is_available()\n"
            "T = torch.as_tensor(st.device_array('temperature'), device='cuda')\n"
            "assert float(T[:, :8].min()) == 3.0 and float(T[:, :8].max()) == 3.0\n"
            "maps = open('/proc/self/maps').read().split('\\n')\n"
            "print(len({l.split()[-1] for l in maps if 'libamdhip64' in l}), len({l.split()[-1] for l in maps if 'libhsa-runtime64' in l}))\n") % root
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split()[-2:] == ["1", "1"], out.stdout


def test_simulation_driver_with_a_state_function_and_an_aligned_last_step():
    """Simulation(integrator; dt, stop_time) (model_integrator.jl:39-66) over a boundary value that reads the state: the driver
    evaluates it before every step, the aligned (shortened) last step included; the oracle is stepped by hand the same way"""
    Nz, Nh, dt = 12, 33, 600.0
    u = columns(Nh)
    grid = trm.ColumnGrid(trm.ExponentialSpacing(N=Nz), Nh)
    bc = trm.PrescribedSurfaceTemperature("Ts", trm.StateFunction(lambda f, clock, p: 0.25 * f.temperature[:, -1] + (4.0 + 1.0e-4 * clock.time)))
    T_init = (1.0 + 3.0 * u)[None, :] * np.ones((Nz, 1))
    integ = trm.initialize(trm.SoilModel(grid), trm.ForwardEuler(dt=dt), boundary_conditions=bc, initializers=dict(temperature=T_init, saturation_water_ice=0.6))
    sim = trm.Simulation(integ, dt=dt, stop_time=4.5 * dt)          # 4 steps of 600 s and one of 300 s
    seen = []
    sim.add_callback(lambda s: seen.append((s.iteration, s.time)), trm.IterationInterval(2), name="progress")
    trm.run_simulation(sim)
    assert sim.time == 4.5 * dt and sim.iteration == 5 and seen == [(0, 0.0), (2, 2 * dt), (4, 4 * dt)]
    o = oracle_like(integ)
    o.set("temperature", T_init)
    o.set("saturation_water_ice", 0.6)
    o.initialize()
    for step_dt in (dt, dt, dt, dt, 0.5 * dt):
        o.set_bc("temperature", "top", "value", 0.25 * o.get("temperature")[-1] + (4.0 + 1.0e-4 * o.clock()[0]))
        o.timestep(step_dt, True)
    for name in ("internal_energy", "temperature", "liquid_water_fraction"):
        assert np.array_equal(integ.state.get(name), o.get(name)), name
