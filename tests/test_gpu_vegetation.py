"""Vegetation processes on the device through the C ABI against the vegetation oracle (pinned by the
reference's test/vegetation unit tests in tests/test_oracle_vegetation.py).  exp / log / pow / sqrt paths: 1e-10 relative
in fp64, 1e-4 in fp32; everything discrete (switches, thresholds) exact."""
import numpy as np
import pytest

import oracle
import terrarium_jl_amd as trm

pytestmark = pytest.mark.gpu

AUX = ("balanced_leaf_area_index", "phenology_factor", "leaf_area_index", "canopy_water_conductance", "leaf_to_air_co2_ratio",
       "net_assimilation", "leaf_respiration", "gross_primary_production", "autotrophic_respiration", "net_primary_production")
PROG = ("carbon_vegetation", "vegetation_area_fraction")


def forcing(n, rng):
    return dict(air_temperature=rng.uniform(-12.0, 40.0, n), air_pressure=rng.uniform(7.0e4, 1.03e5, n),
                specific_humidity=rng.uniform(5e-4, 1.2e-2, n), surface_shortwave_down=np.maximum(0.0, rng.uniform(-200.0, 900.0, n)),
                CO2=rng.uniform(280.0, 560.0, n), soil_moisture_limiting_factor=rng.uniform(0.0, 1.0, n),
                daily_leaf_respiration=rng.uniform(0.0, 1e-4, n), ground_temperature=rng.uniform(-5.0, 25.0, n))


def make_pair(n, dtype, seed=1, stepper=trm.ForwardEuler):
    rng = np.random.default_rng(seed)
    f = forcing(n, rng)
    C0, nu0 = rng.uniform(0.2, 25.0, n), rng.uniform(0.0, 0.9, n)
    grid = trm.ColumnGrid(trm.ExponentialSpacing(N=10), n, dtype=dtype)
    integ = trm.initialize(trm.VegetationModel(grid), stepper(dt=1800.0), inputs=f,
                           initializers=dict(carbon_vegetation=C0, vegetation_area_fraction=nu0))
    o = oracle.VegetationOracle(n, dtype=dtype)
    for k, v in f.items():
        o.set(k, v)
    o.set("carbon_vegetation", C0); o.set("vegetation_area_fraction", nu0)
    return integ, o


def assert_close(st, o, names, dtype):
    tol = 1e-10 if np.dtype(dtype) == np.float64 else 2e-4
    for n in names:
        a, b = st.get(n).astype(np.float64), o.get(n).astype(np.float64)
        assert np.all(np.isfinite(a) == np.isfinite(b)), n
        scale = np.maximum(np.abs(b), 1e-30)
        err = np.abs(a - b) / scale
        assert np.nanmax(np.where(np.abs(b) > 1e-300, err, 0.0)) <= tol, (n, float(np.nanmax(err)))
        assert np.array_equal(a == 0.0, b == 0.0), n                    # the switches (no light, too cold, LAI = 0) are exact


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("stepper", [trm.ForwardEuler, trm.Heun])
def test_vegetation_model_parity(dtype, stepper):
    integ, o = make_pair(777, dtype, stepper=stepper)
    heun = stepper is trm.Heun
    for _ in range(30):
        trm.timestep(integ)
        o.timestep(1800.0, True, heun)
    assert_close(integ.state, o, PROG + AUX, dtype)
    assert trm.current_time(integ) == o.time() == 30 * 1800.0


def test_vegetation_batch_equals_per_step_bitwise():
    a, _ = make_pair(300, np.float64, seed=3)
    b, _ = make_pair(300, np.float64, seed=3)
    trm.run(a, steps=40)                        # one launch: the 0-D column stays in registers
    for n in range(40):
        b.state.step(1800.0, 1, finalize=(n == 39))
    for n in PROG + AUX + ("tend_carbon_vegetation", "tend_vegetation_area_fraction"):
        assert np.array_equal(a.state.get(n), b.state.get(n)), n


def test_vegetation_process_interface():
    """update_state! / compute_auxiliary! / compute_tendencies! / explicit_step! one by one == the fused step."""
    a, o = make_pair(200, np.float64, seed=5)
    b, _ = make_pair(200, np.float64, seed=5)
    st = a.state
    st.update_state(True)
    o.compute_auxiliary(); o.compute_tendencies()
    assert_close(st, o, AUX + ("tend_carbon_vegetation", "tend_vegetation_area_fraction"), np.float64)
    st.explicit_step(1800.0)
    b.state.step(1800.0, 1, finalize=False)
    for n in PROG + AUX:
        assert np.array_equal(st.get(n), b.state.get(n)), n
    # compute_tendencies! ASSIGNS the vegetation tendencies (carbon_dynamics.jl:184, vegetation_dynamics.jl:150), reset_tendencies! clears them
    st.reset_tendencies(); st.compute_auxiliary(); st.compute_tendencies()
    g1 = st.get("tend_carbon_vegetation")
    st.compute_tendencies()
    assert np.array_equal(st.get("tend_carbon_vegetation"), g1) and np.any(g1 != 0)
    st.reset_tendencies()
    assert np.all(st.get("tend_carbon_vegetation") == 0)


def test_vegetation_switches_and_defaults():
    """Known answers of test/vegetation/photosynthesis_tests.jl:268-299 and the input defaults, through the C ABI."""
    grid = trm.ColumnGrid(trm.ExponentialSpacing(N=10), 4)
    integ = trm.initialize(trm.VegetationModel(grid), inputs=dict(air_temperature=np.array([-5.0, 20.0, 20.0, 20.0]), air_pressure=1.0e5,
                                                                  surface_shortwave_down=np.array([50.0, 50.0, 0.0, 50.0]), CO2=400.0),
                           initializers=dict(carbon_vegetation=np.array([11.0, 0.0, 11.0, 11.0]), vegetation_area_fraction=0.3))
    st = integ.state
    assert np.all(st.CO2 == 400.0) and np.all(st.soil_moisture_limiting_factor == 1.0) and np.all(st.ground_temperature == 10.0)
    st.compute_auxiliary()
    An, Rd = st.net_assimilation, st.leaf_respiration
    assert An[0] == 0 and Rd[0] == 0          # T_air < -3
    assert An[1] == 0 and Rd[1] == 0          # LAI = 0
    assert An[2] == 0 and Rd[2] == 0          # no light
    assert An[3] > 0 and Rd[3] > 0 and np.isfinite(An[3])
    assert np.array_equal(st.leaf_area_index, st.balanced_leaf_area_index) and np.all(st.phenology_factor == 1.0)
    assert st.balanced_leaf_area_index[0] == 11.0 / ((2.0 / 10.0) + 2.0)


# test/vegetation/plant_available_water_tests.jl:38-77 and root_distribution_tests.jl:5-15 through the C ABI
def test_plant_available_water_and_root_fractions():
    p = trm._capi.default_params()
    p.por_mineral = 0.5
    grid = trm.ColumnGrid(trm.UniformSpacing(dz=0.2, N=10), 3)
    st = trm.DeviceState(grid, p)
    st.set_vegetation(trm._capi.default_vegetation_params(), "standalone")
    rf = st.root_fraction
    assert np.allclose(rf.sum(axis=0), 1.0, rtol=1e-14) and np.all(rf > 0) and np.all(np.diff(rf[:, 0]) > 0)   # more roots near the surface
    for sat, liq, expected in ((1.0, 1.0, 1.0), (0.0, 1.0, 0.0), (1.0, 0.0, 0.0), (0.2, 1.0, 0.25)):
        st.set("saturation_water_ice", sat)
        st.set("liquid_water_fraction", liq)
        st.compute_plant_available_water()
        assert np.allclose(st.plant_available_water, expected, atol=1e-15)
        assert np.allclose(st.soil_moisture_limiting_factor, (st.plant_available_water * rf).sum(axis=0), rtol=1e-14, atol=1e-16)
    # the oracle's scalar formula on a random profile
    rng = np.random.default_rng(3)
    sat = rng.random((10, 3))
    st.set("saturation_water_ice", sat)
    st.set("liquid_water_fraction", 1.0)
    st.compute_plant_available_water()
    ref = np.vectorize(lambda w: oracle.veg_scalar("plant_available_water", w))(sat * 0.5)
    assert np.array_equal(st.plant_available_water, ref)
