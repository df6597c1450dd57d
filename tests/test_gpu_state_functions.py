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


def test_relaxation_forcing_of_the_water_content_follows_the_state():
    """vwc_forcing = -(sat - target) * rate per cell: a nudging term, the textbook state-dependent Forcing"""
    Nz, Nh, dt = 24, 130, 60.0
    u = columns(Nh)
    grid = trm.ColumnGrid(trm.ExponentialSpacing(N=Nz), Nh)
    zc = grid.z_centers()
    par = dict(target=0.6, rate=1.0 / 7200.0)
    F = trm.StateFunction(lambda f, clock, p: (f.saturation_water_ice - p["target"]) * (-p["rate"]), parameters=par)
    model = trm.SoilModel(grid, soil=trm.SoilEnergyWaterCarbon(hydrology=trm.SoilHydrology(vertical_flow=trm.RichardsEq(), vwc_forcing=F)))
    T_init = (3.0 + 2.0 * u)[None, :] - 0.05 * zc[:, None]
    sat = np.clip(np.minimum(1.0, 0.8 - 0.05 * zc)[:, None] * (1.0 + 0.1 * u)[None, :], 0.05, 1.0)
    integ = trm.initialize(model, trm.ForwardEuler(dt=dt), boundary_conditions=trm.PrescribedSurfaceTemperature("Ts", 5.0 + u),
                           initializers=dict(temperature=T_init, saturation_water_ice=sat))
    o = oracle_like(integ)
    o.set("temperature", T_init)
    o.set("saturation_water_ice", sat)
    o.set_bc("temperature", "top", "value", 5.0 + u)
    o.initialize()
    trm.run(integ, steps=25)
    for n in range(25):
        o.set("vwc_forcing", (o.get("saturation_water_ice") - par["target"]) * (-par["rate"]))
        o.timestep(dt, n == 24)
    assert integ.state.clock() == (25 * dt, 25) and integ.state.status() == 0
    for name in ("saturation_water_ice", "internal_energy", "temperature", "pressure_head", "liquid_water_fraction"):
        assert np.array_equal(integ.state.get(name), o.get(name)), name
    # the forcing did something: without it the column mean stays where the fluxes put it
    plain = trm.initialize(trm.SoilModel(grid, soil=trm.SoilEnergyWaterCarbon(hydrology=trm.SoilHydrology(vertical_flow=trm.RichardsEq()))), trm.ForwardEuler(dt=dt),
                           boundary_conditions=trm.PrescribedSurfaceTemperature("Ts", 5.0 + u), initializers=dict(temperature=T_init, saturation_water_ice=sat))
    trm.run(plain, steps=25)
    assert np.max(np.abs(plain.state.saturation_water_ice - integ.state.saturation_water_ice)) > 1e-4


def test_boundary_value_that_reads_the_top_cell():
    """top temperature value = 0.5 (T_top + 10): a boundary condition in discrete form reading `fields`; timestep! and run!"""
    Nz, Nh, dt = 16, 77, 300.0
    u = columns(Nh)
    grid = trm.ColumnGrid(trm.ExponentialSpacing(N=Nz), Nh)
    bc = trm.PrescribedSurfaceTemperature("Ts", trm.StateFunction(lambda f, clock, p: 0.5 * (f.temperature[:, -1] + 10.0)))
    T_init = (2.0 + 4.0 * u)[None, :] * np.ones((Nz, 1))
    integ = trm.initialize(trm.SoilModel(grid), trm.ForwardEuler(dt=dt), boundary_conditions=bc, initializers=dict(temperature=T_init, saturation_water_ice=0.8))
    o = oracle_like(integ)
    o.set("temperature", T_init)
    o.set("saturation_water_ice", 0.8)
    o.initialize()
    trm.timestep(integ)
    trm.run(integ, steps=11)
    for n in range(12):
        o.set_bc("temperature", "top", "value", 0.5 * (o.get("temperature")[-1] + 10.0))
        o.timestep(dt, True)
    for name in ("internal_energy", "temperature", "liquid_water_fraction"):
        assert np.array_equal(integ.state.get(name), o.get(name)), name
    assert np.all(integ.state.temperature[-1] != T_init[-1])


def test_atmospheric_input_that_follows_the_skin_temperature():
    """LandModel input air_temperature = skin temperature + 2 K: an input source driven by the land state (the coupling
    direction of speedy_dry_land.jl:45-66, here as a function of `fields`); the clock reaches the function"""
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
    integ = trm.initialize(land, trm.ForwardEuler(dt=dt), initializers=dict(temperature=T_init, saturation_water_ice=sat),
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
        o.set("air_temperature", np.ravel(o.get("skin_temperature")) + 2.0)
        o.timestep(dt, n == 14)
    assert seen == [(n * dt, n) for n in range(15)]
    for name in ("skin_temperature", "internal_energy", "saturation_water_ice", "ground_heat_flux", "sensible_heat_flux"):
        a, b = integ.state.get(name), o.get(name)
        assert np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))) < 1e-10, name


def test_heun_refuses_a_state_function():
    grid = trm.ColumnGrid(trm.ExponentialSpacing(N=8), 4)
    bc = trm.PrescribedSurfaceTemperature("Ts", trm.StateFunction(lambda f, clock, p: f.temperature[:, -1]))
    with pytest.raises(NotImplementedError):
        trm.initialize(trm.SoilModel(grid), trm.Heun(dt=10.0), boundary_conditions=bc)


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
