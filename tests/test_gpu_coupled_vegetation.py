"""LandModel coupled to vegetation on the device (stage 2: canopy interception, canopy
evapotranspiration, plant available water from the soil, the 0-D prognostics stepped with the soil) through the C ABI against
the coupled oracle (pinned in tests/test_oracle_canopy.py by the reference's canopy unit tests and its "Coupled
vegetation-soil" test).  exp / pow paths: 1e-10 relative in fp64, 2e-4 in fp32."""
import numpy as np
import pytest

import oracle
import terrarium_jl_amd as trm
import workloads as W

pytestmark = pytest.mark.gpu

VEG_AUX = ("balanced_leaf_area_index", "phenology_factor", "leaf_area_index", "canopy_water_conductance", "leaf_to_air_co2_ratio",
           "net_assimilation", "leaf_respiration", "gross_primary_production", "autotrophic_respiration", "net_primary_production",
           "soil_moisture_limiting_factor", "plant_available_water")
CANOPY_AUX = ("canopy_water_interception", "canopy_water_removal", "saturation_canopy_water", "rainfall_ground", "evaporation_canopy",
              "transpiration", "evaporation_ground")
SURFACE = ("skin_temperature", "ground_heat_flux", "latent_heat_flux", "sensible_heat_flux", "surface_net_radiation", "infiltration",
           "surface_runoff")
PROG = ("temperature", "saturation_water_ice", "internal_energy", "surface_excess_water", "canopy_water", "carbon_vegetation",
        "vegetation_area_fraction")
TEND = ("tend_canopy_water", "tend_carbon_vegetation", "tend_vegetation_area_fraction", "tend_internal_energy", "tend_saturation_water_ice")


def land_with_vegetation(grid):
    """land_model_tests.jl:39-45"""
    swrc = trm.VanGenuchten(alpha=2.0, n=2.0)
    hp = trm.ConstantSoilHydraulics(swrc=swrc, unsat_hydraulic_cond=trm.UnsatKVanGenuchten())
    soil = trm.SoilEnergyWaterCarbon(hydrology=trm.SoilHydrology(vertical_flow=trm.RichardsEq(), hydraulic_properties=hp))
    return trm.LandModel(grid, soil=soil, vegetation=trm.VegetationCarbon())


def make_pair(n, dtype, seed=3, stepper=trm.ForwardEuler, dt=0.5, N=20, rain=True):
    rng = np.random.default_rng(seed)
    grid = trm.ColumnGrid(trm.ExponentialSpacing(dz_max=1.0, N=N), n, dtype=dtype)
    zc = grid.z_centers()
    T0 = (5.0 - 0.02 * zc)[:, None] + rng.uniform(-1, 1, n)[None, :]
    sat0 = np.clip(np.minimum(1.0, 0.8 - 0.05 * zc)[:, None] * (1 + 0.05 * rng.uniform(-1, 1, n))[None, :], 0.05, 1.0)
    inits = dict(temperature=T0, saturation_water_ice=sat0, carbon_vegetation=rng.uniform(1.0, 2.0, n),
                 vegetation_area_fraction=rng.uniform(0.05, 0.9, n), canopy_water=rng.uniform(0.0, 1.0e-4, n))
    inputs = dict(SAI=rng.uniform(0.0, 1.0, n), rainfall=rng.uniform(0.0, 2.0e-7, n) if rain else 0.0, air_temperature=rng.uniform(2.0, 20.0, n),
                  specific_humidity=rng.uniform(1.0e-3, 5.0e-3, n), windspeed=rng.uniform(0.0, 4.0, n), CO2=rng.uniform(300.0, 500.0, n),
                  daily_leaf_respiration=rng.uniform(0.0, 1e-6, n))
    integ = trm.initialize(land_with_vegetation(grid), stepper(dt=dt), initializers=inits, inputs=inputs)
    o = oracle.Oracle(n, grid.thickness, oracle.default_params(flow=1, seb=1, swrc=1, unsat_k=1, vg_alpha=2.0, vg_n=2.0), dtype=dtype)
    o.enable_vegetation()
    for k, v in {**inits, **inputs}.items():
        o.set(k, v)
    o.initialize()
    return integ, o


def assert_close(st, o, names, dtype, tol=None):
    tol = tol or (1e-10 if np.dtype(dtype) == np.float64 else 2e-4)
    for n in names:
        a, b = st.get(n).astype(np.float64), o.get(n).astype(np.float64)
        assert np.all(np.isfinite(a)) and np.all(np.isfinite(b)), n
        scale = np.maximum(np.abs(b), np.abs(b).max() * 1e-6 + 1e-300)
        err = np.abs(a - b) / scale
        assert err.max() <= tol, (n, float(err.max()))


# test/coupled_models/land_model_tests.jl:38-71 "LandModel: Coupled vegetation-soil"
def test_reference_coupled_vegetation_soil():
    grid = trm.ColumnGrid(trm.ExponentialSpacing(dz_max=1.0, N=50))
    land = land_with_vegetation(grid)
    assert isinstance(land.surface_hydrology.evapotranspiration, trm.PALADYNCanopyEvapotranspiration)
    assert isinstance(land.surface_hydrology.canopy_interception, trm.PALADYNCanopyInterception)
    integ = trm.initialize(land, trm.ForwardEuler(), initializers=dict(temperature=lambda x, z: 5.0 - 0.02 * z,
                           saturation_water_ice=lambda x, z: min(1, 0.8 - 0.05 * z), carbon_vegetation=0.1))
    st = integ.state
    trm.timestep(integ, 60.0)
    for name in ("saturation_water_ice", "internal_energy", "ground_heat_flux", "carbon_vegetation"):
        assert np.all(np.isfinite(st.get(name))), name
    assert st.status() == 0
    # the same step in the oracle
    o = oracle.Oracle(1, grid.thickness, oracle.default_params(flow=1, seb=1, swrc=1, unsat_k=1, vg_alpha=2.0, vg_n=2.0))
    o.enable_vegetation()
    zc = grid.z_centers()
    o.set("temperature", 5.0 - 0.02 * zc); o.set("saturation_water_ice", np.minimum(1.0, 0.8 - 0.05 * zc)); o.set("carbon_vegetation", 0.1)
    o.initialize()
    o.timestep(60.0)
    assert_close(st, o, PROG + SURFACE + VEG_AUX + CANOPY_AUX, np.float64)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_coupled_update_state_parity(dtype):
    integ, o = make_pair(333, dtype)
    st = integ.state
    st.update_state(True)
    o.update_state(True)
    assert_close(st, o, VEG_AUX + CANOPY_AUX + SURFACE + TEND[:3], dtype)
    for n in TEND[3:]:     # flux differences: cancellation makes the small entries noisy in fp32, compare against the field's scale
        a, b = st.get(n).astype(np.float64), o.get(n).astype(np.float64)
        assert np.abs(a - b).max() <= (1e-10 if dtype == np.float64 else 2e-4) * np.abs(b).max(), n
    # identities of the canopy water budget hold on the device's own numbers
    I, R, E = st.canopy_water_interception, st.canopy_water_removal, st.evaporation_canopy
    assert np.array_equal(st.tend_canopy_water, I - E - R)
    assert np.array_equal(st.rainfall_ground, st.rainfall - I + R)
    Q = (st.evaporation_ground + st.evaporation_canopy) + st.transpiration
    p = o.params
    assert np.allclose(st.latent_heat_flux, np.asarray((p.Llg * p.rho_a) * Q.astype(np.float64), dtype=dtype), rtol=1e-6 if dtype == np.float32 else 1e-15)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("stepper", [trm.ForwardEuler, trm.Heun])
def test_coupled_steps_parity(dtype, stepper):
    integ, o = make_pair(257, dtype, stepper=stepper)
    heun = stepper is trm.Heun
    for _ in range(25):
        trm.timestep(integ)
        (o.timestep_heun if heun else o.timestep)(0.5)
    assert integ.state.status() == 0 and o.status() == 0
    assert_close(integ.state, o, PROG + SURFACE + VEG_AUX + CANOPY_AUX, dtype, tol=None if dtype == np.float64 else 1e-3)
    assert trm.current_time(integ) == 12.5


def test_coupled_batch_equals_single_steps_and_reference_order_kernels():
    a, _ = make_pair(200, np.float64, seed=5)
    b, _ = make_pair(200, np.float64, seed=5)
    c, _ = make_pair(200, np.float64, seed=5)
    a.state.set_option("steps_per_launch", 8)      # (the resident multi-step program does not apply: silently one launch pair per step)
    trm.run(a, steps=16)
    for n in range(16):
        b.state.step(0.5, 1, finalize=(n == 15))
    c.state.set_option("step_kernel", "unfused")
    trm.run(c, steps=16)
    for n in PROG + SURFACE + VEG_AUX + CANOPY_AUX:
        assert np.array_equal(a.state.get(n), b.state.get(n)), n
        assert np.array_equal(a.state.get(n), c.state.get(n)), n


def test_coupled_process_interface():
    """update_state! + explicit_step! + closure! one by one == the fused launch pair, bit for bit."""
    a, _ = make_pair(150, np.float64, seed=7)
    b, _ = make_pair(150, np.float64, seed=7)
    st = a.state
    st.update_state(True)
    g = {n: st.get(n) for n in ("tend_canopy_water", "tend_carbon_vegetation", "tend_vegetation_area_fraction")}
    w0, c0, nu0 = st.canopy_water, st.carbon_vegetation, st.vegetation_area_fraction
    st.explicit_step(0.5)
    assert np.array_equal(st.canopy_water, w0 + g["tend_canopy_water"] * 0.5)
    assert np.array_equal(st.carbon_vegetation, c0 + g["tend_carbon_vegetation"] * 0.5)
    assert np.array_equal(st.vegetation_area_fraction, nu0 + g["tend_vegetation_area_fraction"] * 0.5)
    st.closure()
    b.state.step(0.5, 1, finalize=False)
    for n in PROG:
        assert np.array_equal(st.get(n), b.state.get(n)), n
    # the 0-D tendencies are ASSIGNED by compute_tendencies! (canopy_interception.jl:214, carbon_dynamics.jl:184), the soil's accumulate
    st.update_state(True)
    g = {n: st.get(n) for n in TEND}
    st.compute_tendencies()
    for n in TEND[:3]:
        assert np.array_equal(st.get(n), g[n]), n
    assert np.allclose(st.get("tend_internal_energy"), 2 * g["tend_internal_energy"], rtol=1e-15)


def test_coupled_needs_land_model_and_dry_canopy_is_inert():
    grid = trm.ColumnGrid(trm.ExponentialSpacing(N=10), 4)
    soil_only = trm.initialize(trm.SoilModel(grid))
    with pytest.raises(trm.TerrariumHipError):
        soil_only.state.set_vegetation(trm.flatten_vegetation(trm.VegetationCarbon()), "coupled")
    # no rain, empty store: nothing is intercepted, evaporated from the canopy or removed; all the (zero) rain reaches the ground
    integ, _ = make_pair(64, np.float64, rain=False)
    st = integ.state
    st.set("canopy_water", 0.0)
    trm.run(integ, steps=5)
    for n in ("canopy_water", "canopy_water_interception", "canopy_water_removal", "evaporation_canopy", "rainfall_ground", "saturation_canopy_water"):
        assert np.all(st.get(n) == 0.0), n
    assert np.all(st.transpiration > 0) and np.all(np.isfinite(st.latent_heat_flux))


@pytest.mark.parametrize("N", [40, 70])
def test_coupled_deep_and_64_lane_columns(N):
    """Nz = 40 takes the 64-lane column kernel and the 64-level pitch of the cooperative 0-D kernel; Nz = 70 is deeper than
    the fused kernels go: reference-order kernels and the one-thread-per-column 0-D kernel."""
    integ, o = make_pair(130, np.float64, seed=11, N=N)
    for _ in range(10):
        trm.timestep(integ)
        o.timestep(0.5)
    assert integ.state.status() == 0 and o.status() == 0
    assert_close(integ.state, o, PROG + SURFACE + VEG_AUX + CANOPY_AUX, np.float64)
    st = integ.state
    st.compute_plant_available_water()
    assert np.allclose(st.soil_moisture_limiting_factor, o.get("soil_moisture_limiting_factor"), rtol=1e-12)
    with pytest.raises(trm.TerrariumHipError):
        st.set("root_fraction", 0.0)          # static: derived from the root distribution parameters


def test_bench_workload_with_vegetation_matches_oracle():
    """The `c4vgveg` workload of bench.py (tests/workloads.py "landveg") on a sample of the N72 columns, device vs oracle."""
    lat, lon = W.columns_from_mask("N72")
    w = W.make_workload("landveg", lat[::40], lon[::40], 32, hydraulics="vg")
    orc, dev = W.setup_oracle(w), W.setup_device(w)
    orc.run(w["dt"], 30)
    dev.step(w["dt"], 30, True)
    assert dev.status() == 0 and orc.status() == 0
    assert_close(dev, orc, PROG + SURFACE + VEG_AUX + CANOPY_AUX, np.float64)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_coupled_heun_fused_equals_reference_order_kernels(dtype):
    """Heun of the coupled model in four launches (the soil stages in registers, the 0-D processes evaluated at the state and at
    the stage) against the reference-order kernels on a full copy of the state: bit for bit, with a forcing series so that the
    stage sees its inputs at t + dt."""
    a, _ = make_pair(190, dtype, seed=13, stepper=trm.Heun)
    b, _ = make_pair(190, dtype, seed=13, stepper=trm.Heun)
    b.state.set_option("step_kernel", "unfused")
    rng = np.random.default_rng(5)
    times = np.array([0.0, 2.0, 4.0, 8.0])
    for integ in (a, b):
        integ.state.set_forcing_series("air_temperature", times, rng.uniform(2.0, 20.0, (4, 190)))
        integ.state.set_forcing_series("rainfall", times, rng.uniform(0.0, 2.0e-7, (4, 190)))
        rng = np.random.default_rng(5)
    for n in range(12):
        a.state.step_heun(0.5, 1, finalize=(n % 4 == 3))
        b.state.step_heun(0.5, 1, finalize=(n % 4 == 3))
    assert a.state.status() == b.state.status() == 0
    for n in PROG + SURFACE + VEG_AUX + CANOPY_AUX + TEND[:3]:
        assert np.array_equal(a.state.get(n), b.state.get(n)), n
