"""GPU parity: the HIP library (through its C ABI) against the CPU oracle on the
same seeded inputs.  Bars: paths made only of + - * / sqrt fma are
BIT-EXACT (heat-only; Richards with the reference-default BrooksCorey + linear K,
whose powers Julia evaluates by compensated squaring); paths through generic
pow / exp (van Genuchten, surface energy balance) agree to 1e-10 * max(1, |x|)
in fp64 after the stated number of steps; fp32 to 1e-4 relative."""
import numpy as np
import pytest

import workloads as W
import terrarium_jl_amd as trm

pytestmark = pytest.mark.gpu

TOL64 = 1.0e-10   # fp64 tolerance for pow/exp paths, relative to max(1, |x|)
TOL32 = 1.0e-4    # fp32 tolerance


def small_columns(n, name="N72"):
    lat, lon = W.columns_from_mask(name)
    sel = np.linspace(0, lat.size - 1, n).astype(int)
    return lat[sel], lon[sel]


def assert_fields_match(dev, orc, names, exact, tol, label=""):
    for n in names:
        a, b = dev.get(n), orc.get(n)
        assert a.shape == b.shape, n
        if exact:
            assert np.array_equal(a, b, equal_nan=True), (
                f"{label}{n}: max abs diff {np.nanmax(np.abs(a.astype(np.float64) - b.astype(np.float64)))}")
        else:
            a64, b64 = a.astype(np.float64), b.astype(np.float64)
            scale = np.maximum(1.0, np.abs(b64))
            err = np.abs(a64 - b64) / scale
            assert np.all(np.isfinite(a64) == np.isfinite(b64)), n
            assert np.nanmax(err) <= tol, f"{label}{n}: max scaled err {np.nanmax(err):.3e} > {tol}"


def bit_exact_config(config, hydraulics, dtype):
    return np.dtype(dtype) == np.float64 and config in ("heat", "richards") and hydraulics == "default"


CASES = [
    ("heat", "default", np.float64, 20, 100),
    ("heat", "default", np.float64, 30, 100),
    ("richards", "default", np.float64, 32, 100),
    ("richards", "vg", np.float64, 32, 100),
    ("land", "default", np.float64, 32, 50),
    ("land", "vg", np.float64, 32, 50),
    ("heat", "default", np.float32, 20, 100),
    ("richards", "default", np.float32, 64, 100),
    ("land", "vg", np.float32, 64, 50),
]


@pytest.mark.parametrize("kernel", ["fused", "unfused"])
@pytest.mark.parametrize("config,hydraulics,dtype,Nz,nsteps", CASES)
def test_step_parity(config, hydraulics, dtype, Nz, nsteps, kernel):
    lat, lon = small_columns(333)  # ragged: not a multiple of the 64-column tile
    w = W.make_workload(config, lat, lon, Nz, dtype=dtype, hydraulics=hydraulics)
    orc = W.setup_oracle(w)
    dev = W.setup_device(w)
    dev.set_option("step_kernel", kernel)
    names = W.compared_fields(w)
    exact = bit_exact_config(config, hydraulics, dtype)
    tol = TOL64 if np.dtype(dtype) == np.float64 else TOL32
    # initial state (process initialisers) must already agree
    assert_fields_match(dev, orc, names, exact, tol, "init ")
    # one reference timestep!(integrator, dt) with finalize, then a run!(steps)
    orc.timestep(w["dt"], finalize=True)
    dev.step(w["dt"], 1, finalize=True)
    assert_fields_match(dev, orc, names, exact, tol, "step1 ")
    orc.run(w["dt"], nsteps - 1)
    dev.step(w["dt"], nsteps - 1, finalize=True)
    assert dev.clock() == orc.clock()
    assert_fields_match(dev, orc, names, exact, tol, f"step{nsteps} ")
    assert dev.status() == orc.status() == 0


@pytest.mark.parametrize("config,hydraulics", [("heat", "default"), ("richards", "default"), ("richards", "vg"),
                                                ("land", "vg")])
@pytest.mark.parametrize("Nz", [32, 20, 64])
def test_fused_equals_unfused_bitwise(config, hydraulics, Nz):
    """All implementations share the device arithmetic, so they must agree bit for bit on every path."""
    lat, lon = small_columns(500)
    w = W.make_workload(config, lat, lon, Nz, hydraulics=hydraulics)
    a, b = W.setup_device(w), W.setup_device(w)
    b.set_option("step_kernel", "unfused")
    for nsteps, fin in ((1, True), (7, False), (12, True)):
        a.step(w["dt"], nsteps, finalize=fin)
        b.step(w["dt"], nsteps, finalize=fin)
        # without finalize the auxiliaries hold the values of the last compute_auxiliary! (pre-update state)
        for n in W.compared_fields(w):
            assert np.array_equal(a.get(n), b.get(n), equal_nan=True), (config, n, nsteps, fin)


@pytest.mark.parametrize("kernel", ["fused", "unfused"])
def test_skipping_intermediate_conductivity_stores(kernel):
    """TRM_OPT_WRITE_KF_EVERY_STEP = 0: hydraulic_conductivity is never an input of a step, so storing it only
    when finalizing must leave every field -- including the final K -- unchanged."""
    lat, lon = small_columns(200)
    w = W.make_workload("richards", lat, lon, 32)
    a, b = W.setup_device(w), W.setup_device(w)
    for d in (a, b):
        d.set_option("step_kernel", kernel)
    b.set_option("write_kf_every_step", 0)
    a.step(w["dt"], 20, True)
    b.step(w["dt"], 20, True)
    for n in W.compared_fields(w):
        assert np.array_equal(a.get(n), b.get(n)), n


def test_process_interface_parity():
    """Stand-alone compute_auxiliary! / compute_tendencies! / explicit_step! / closure! / invclosure!"""
    lat, lon = small_columns(130)
    for hydraulics, exact in (("default", True), ("vg", False)):
        w = W.make_workload("richards", lat, lon, 32, hydraulics=hydraulics)
        orc, dev = W.setup_oracle(w), W.setup_device(w)
        orc.update_state(True)
        dev.update_state(True)
        names = ["hydraulic_conductivity", "tend_internal_energy", "tend_saturation_water_ice",
                 "tend_surface_excess_water"]
        assert_fields_match(dev, orc, names, exact, TOL64, "update_state ")
        orc.explicit_step(w["dt"])
        dev.explicit_step(w["dt"])
        assert_fields_match(dev, orc, ["internal_energy", "saturation_water_ice", "surface_excess_water",
                                       "tend_internal_energy"], exact, TOL64, "explicit_step ")
        orc.closure()
        dev.closure()
        assert_fields_match(dev, orc, W.compared_fields(w), exact, TOL64, "closure ")
        orc.invclosure()
        dev.invclosure()
        # pressure -> saturation evaluates theta(psi) = (-psi_s/psi)^lambda with a non-integer lambda: generic pow
        assert_fields_match(dev, orc, ["internal_energy", "saturation_water_ice", "liquid_water_fraction",
                                       "water_table"], False, TOL64, "invclosure ")


def test_land_process_interface_parity():
    lat, lon = small_columns(130)
    w = W.make_workload("land", lat, lon, 32, hydraulics="vg")
    orc, dev = W.setup_oracle(w), W.setup_device(w)
    orc.compute_auxiliary()
    dev.compute_auxiliary()
    assert_fields_match(dev, orc, W.FIELDS_SEB + ("hydraulic_conductivity",), False, TOL64, "compute_auxiliary ")


@pytest.mark.parametrize("config,hydraulics", [("heat", "default"), ("richards", "default"), ("land", "vg")])
def test_heun_parity(config, hydraulics):
    lat, lon = small_columns(97)
    w = W.make_workload(config, lat, lon, 20, hydraulics=hydraulics)
    orc, dev = W.setup_oracle(w), W.setup_device(w)
    for _ in range(5):
        orc.timestep_heun(w["dt"], finalize=True)
    dev.step_heun(w["dt"], 5, finalize=True)
    # the library finalizes once after the last step, the loop above after every step: identical for
    # SoilModel (auxiliaries do not feed back) but not for LandModel's in-place skin temperature
    if config == "land":
        orc = W.setup_oracle(w)
        for n in range(5):
            orc.timestep_heun(w["dt"], finalize=(n == 4))
    exact = bit_exact_config(config, hydraulics, np.float64)
    assert_fields_match(dev, orc, W.compared_fields(w), exact, TOL64, "heun ")
    assert dev.clock() == orc.clock()


@pytest.mark.parametrize("config,hydraulics,Nz", [("heat", "default", 20), ("richards", "default", 32), ("richards", "vg", 64),
                                                   ("land", "default", 32), ("land", "vg", 20)])
def test_heun_fused_equals_unfused_bitwise(config, hydraulics, Nz):
    """Heun in two fused launches (predictor into the stage, corrector from the stage's tendencies) against the
    reference-order kernels on a second copy of the state: same arithmetic, so bit for bit -- including the stored
    hydraulic conductivity, surface excess water, water table and skin temperature."""
    lat, lon = small_columns(131)
    w = W.make_workload(config, lat, lon, Nz, hydraulics=hydraulics)
    if config == "richards":
        w["bcs"][("saturation_water_ice", "top")] = ("flux", np.where(np.arange(131) % 3 == 0, -2.0e-4, 0.0))
        w["bcs"][("internal_energy", "bottom")] = ("flux", np.full(131, 0.05))
    a, b = W.setup_device(w), W.setup_device(w)
    b.set_option("step_kernel", "unfused")
    for nsteps, fin in ((3, False), (4, True)):
        a.step_heun(w["dt"], nsteps, fin)
        b.step_heun(w["dt"], nsteps, fin)
        for n in W.compared_fields(w):
            assert np.array_equal(a.get(n), b.get(n), equal_nan=True), (n, nsteps)
    assert a.clock() == b.clock() and a.status() == b.status()


def test_halo_policy_mirror():
    lat, lon = small_columns(70)
    w = W.make_workload("heat", lat, lon, 20, halo_policy="mirror")
    orc, dev = W.setup_oracle(w), W.setup_device(w)
    orc.run(w["dt"], 30)
    dev.step(w["dt"], 30, True)
    assert_fields_match(dev, orc, W.compared_fields(w), True, 0.0, "mirror ")
    w0 = W.make_workload("heat", lat, lon, 20)
    d0 = W.setup_device(w0)
    d0.step(w0["dt"], 30, True)
    assert not np.array_equal(d0.temperature, dev.temperature)  # the policy matters (SURVEY C-1)


@pytest.mark.parametrize("Nh", [1, 63, 64, 65, 129])
@pytest.mark.parametrize("Nz", [2, 3, 5])
@pytest.mark.parametrize("kernel", ["fused", "unfused"])
def test_ragged_and_tiny_shapes(Nh, Nz, kernel):
    lat, lon = small_columns(max(Nh, 2))
    w = W.make_workload("richards", lat[:Nh], lon[:Nh], Nz)
    orc, dev = W.setup_oracle(w), W.setup_device(w)
    dev.set_option("step_kernel", kernel)
    orc.run(w["dt"], 10)
    dev.step(w["dt"], 10, True)
    assert_fields_match(dev, orc, W.compared_fields(w), True, 0.0, f"Nh={Nh} Nz={Nz} ")


@pytest.mark.parametrize("Nz", [17, 31, 33, 50, 63])
@pytest.mark.parametrize("config,dtype", [("richards", np.float64), ("land", np.float64), ("land", np.float32)])
def test_partly_filled_lane_groups(config, dtype, Nz):
    """Level counts that do not fill their 32- / 64-lane group (the reference's own tests use N = 50): the idle lanes
    sit next to the top level in the wave-shift stencil, in the ballots of the repair / water table, and -- for fp32 --
    in the packed two-column kernel.  Euler and Heun, against the oracle."""
    lat, lon = small_columns(45)
    w = W.make_workload(config, lat, lon, Nz, dtype=dtype)
    if config == "richards":
        w["bcs"][("saturation_water_ice", "top")] = ("flux", np.where(np.arange(45) % 2 == 0, -3.0e-4, 0.0))
    exact = bit_exact_config(config, "default", dtype)
    tol = TOL64 if dtype == np.float64 else TOL32
    orc, dev = W.setup_oracle(w), W.setup_device(w)
    orc.run(w["dt"], 12)
    dev.step(w["dt"], 12, True)
    assert_fields_match(dev, orc, W.compared_fields(w), exact, tol, f"Nz={Nz} euler ")
    orc, dev = W.setup_oracle(w), W.setup_device(w)
    for n in range(4):
        orc.timestep_heun(w["dt"], n == 3)
    dev.step_heun(w["dt"], 4, True)
    assert_fields_match(dev, orc, W.compared_fields(w), exact, tol, f"Nz={Nz} heun ")


def test_bc_kinds_heun_parity():
    """Heun with the generic boundary kinds runs in one launch per step as well (k_heun_generic): against the oracle and the
    reference-order kernels bit for bit, with a gradient condition that is a time series (the stage takes it at t + dt)."""
    lat, lon = small_columns(75)
    w = W.make_workload("richards", lat, lon, 20)
    rng = np.random.default_rng(11)
    w["bcs"] = {
        ("temperature", "top"): ("value", w["T0"] + 2.0),
        ("temperature", "bottom"): ("gradient", np.full(75, 0.02)),
        ("internal_energy", "bottom"): ("flux", np.full(75, 0.05)),
        ("saturation_water_ice", "top"): ("flux", -1.0e-8 * rng.random(75)),
        ("pressure_head", "bottom"): ("gradient", 0.0),                        # FreeDrainage()
    }
    times = np.array([0.0, 100.0, 250.0, 1000.0])
    grad = rng.uniform(-0.05, 0.05, (4, 75))
    orc, dev, ref = W.setup_oracle(w), W.setup_device(w), W.setup_device(w)
    ref.set_option("step_kernel", "unfused")
    for t in (orc, dev, ref):
        t.set_bc_series("liquid_water_fraction", "top", "gradient", times, grad)
    for n in range(9):
        orc.timestep_heun(w["dt"], n == 8)
    dev.step_heun(w["dt"], 9, True)
    ref.step_heun(w["dt"], 9, True)
    assert_fields_match(dev, orc, W.compared_fields(w), True, 0.0, "heun generic bcs ")
    for n in W.compared_fields(w) + ["tend_internal_energy", "tend_saturation_water_ice"]:
        assert np.array_equal(dev.get(n), ref.get(n), equal_nan=True), n
    assert dev.clock() == ref.clock()


@pytest.mark.parametrize("params", [dict(swrc=1, unsat_k=1, vg_alpha=1.3, vg_n=1.7), dict(swrc=1, unsat_k=1, vg_alpha=2.0, vg_n=3.0),
                                    dict(swrc=1, unsat_k=0, vg_alpha=2.0, vg_n=2.0), dict(swrc=0, unsat_k=1, vg_alpha=2.0, vg_n=2.0)])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_generic_hydraulics_parity(params, dtype):
    """The compile-time van Genuchten instance is n = 2 (every reference test and example); any other exponent and the
    mixed retention / conductivity combinations take the run-time instance (HYD_GENERIC): generic x^y, device pow vs the
    oracle's, 1e-10 / 1e-4.  Fused == reference-order kernels bit for bit there too."""
    lat, lon = small_columns(60)
    w = W.make_workload("land", lat, lon, 20, dtype=dtype)
    w["params"].update(params)
    orc, dev, ref = W.setup_oracle(w), W.setup_device(w), W.setup_device(w)
    ref.set_option("step_kernel", "unfused")
    orc.run(w["dt"], 15)
    dev.step(w["dt"], 15, True)
    ref.step(w["dt"], 15, True)
    assert_fields_match(dev, orc, W.compared_fields(w), False, TOL64 if dtype == np.float64 else TOL32, f"{params} ")
    for n in W.compared_fields(w):
        assert np.array_equal(dev.get(n), ref.get(n), equal_nan=True), n


def test_bc_kinds_parity():
    """Value / Flux / Gradient boundary conditions on every variable that carries them."""
    lat, lon = small_columns(80)
    w = W.make_workload("richards", lat, lon, 16)
    rng = np.random.default_rng(7)
    w["bcs"] = {
        ("temperature", "top"): ("value", w["T0"] + 3.0),
        ("temperature", "bottom"): ("value", w["T0"] - 1.0),
        ("internal_energy", "bottom"): ("flux", np.full(80, 0.05)),            # geothermal heat flux
        ("saturation_water_ice", "top"): ("flux", -1.0e-8 * rng.random(80)),   # infiltration (negative = downward)
        ("pressure_head", "bottom"): ("gradient", 0.0),                        # FreeDrainage()
        ("liquid_water_fraction", "top"): ("gradient", 0.1),
    }
    orc, dev = W.setup_oracle(w), W.setup_device(w)
    orc.run(w["dt"], 40)
    dev.step(w["dt"], 40, True)
    assert_fields_match(dev, orc, W.compared_fields(w), True, 0.0, "bcs ")
    # and through the unfused kernels
    for kern in ("unfused",):
        dev2 = W.setup_device(w)
        dev2.set_option("step_kernel", kern)
        dev2.step(w["dt"], 40, True)
        assert_fields_match(dev2, orc, W.compared_fields(w), True, 0.0, f"bcs {kern} ")


def test_saturation_repair_cases_on_device():
    """K8 (test/soil/soil_hydrology_tests.jl:93-123) through trm_closure, against the oracle."""
    import oracle
    thickness = trm.UniformSpacing(dz=0.1, N=100).get_spacing()
    p = trm._capi.default_params()
    p.flow, p.swrc, p.unsat_k, p.vg_alpha, p.vg_n = 1, 1, 1, 2.0, 2.0
    grid = trm.ColumnGrid(trm.PrescribedSpacing(dz=list(thickness)), 3)
    dev = trm.DeviceState(grid, p)
    orc = oracle.Oracle(3, thickness, oracle.default_params(flow=1, swrc=1, unsat_k=1, vg_alpha=2.0, vg_n=2.0))
    zc = dev.z_centers()
    cases = np.stack([np.maximum(1.1 + zc, 1.0), np.minimum(-0.1 - zc, 1.0), np.minimum(-0.1 - zc, 0.0)], axis=1)
    dev.set("saturation_water_ice", cases)
    orc.set("saturation_water_ice", cases)
    dev.closure()
    orc.closure()
    for n in ("saturation_water_ice", "surface_excess_water", "water_table"):
        assert np.array_equal(dev.get(n), orc.get(n)), n
    sat = dev.saturation_water_ice
    assert np.allclose(sat[:, 0], 1.0) and np.all(sat[:, 1] >= 0) and np.allclose(sat[:, 2], 0.0)


@pytest.mark.parametrize("Nz", [2, 5, 32, 64, 100])
def test_saturation_repair_random_profiles(Nz):
    """adjust_saturation_profile! (soil_hydrology.jl:185-219) on random profiles through trm_closure: mostly legal
    cells with sparse or dense over- (sat > 1) and undersaturated (sat < 0) runs, -0.0 cells, exact 0 and 1,
    on a non-uniform grid.  The device runs range-limited lane-serial passes (sequential kernel for Nz > 64);
    saturation and the surface overflow must match the oracle's full sequential passes bit for bit."""
    import oracle
    rng = np.random.Generator(np.random.PCG64(1234 + Nz))
    Nh = 257
    thickness = np.round(rng.uniform(0.05, 0.4, size=Nz), 3)
    p = trm._capi.default_params()
    p.flow = 1
    grid = trm.ColumnGrid(trm.PrescribedSpacing(dz=list(thickness)), Nh)
    dev = trm.DeviceState(grid, p)
    orc = oracle.Oracle(Nh, thickness, oracle.default_params(flow=1))
    sat = rng.uniform(0.05, 0.999, size=(Nz, Nh))
    density = rng.choice([0.0, 0.03, 0.2, 0.7], size=Nh)[None, :]        # per column: none / sparse / dense
    r = rng.uniform(size=(Nz, Nh))
    sat = np.where(r < 0.5 * density, rng.uniform(1.0, 1.6, size=(Nz, Nh)), sat)
    sat = np.where((r >= 0.5 * density) & (r < density), rng.uniform(-0.5, 0.0, size=(Nz, Nh)), sat)
    special = rng.uniform(size=(Nz, Nh))
    sat = np.where(special < 0.02, -0.0, sat)
    sat = np.where((special >= 0.02) & (special < 0.04), 1.0, sat)
    sat = np.where((special >= 0.04) & (special < 0.05), 0.0, sat)
    sat[:, 0] = np.linspace(1.2, 1.5, Nz)          # every level oversaturated: the carry runs the full column
    sat[:, 1] = -np.linspace(0.1, 0.3, Nz)         # every level negative
    dev.set("saturation_water_ice", sat)
    orc.set("saturation_water_ice", sat)
    dev.closure()
    orc.closure()
    for n in ("saturation_water_ice", "surface_excess_water"):
        a, b = dev.get(n), orc.get(n)
        assert np.array_equal(a, b), (n, np.argwhere(a != b)[:5])
        assert np.array_equal(np.signbit(a), np.signbit(b)), n + " (sign of zero)"


@pytest.mark.parametrize("kernel", ["fused", "unfused"])
def test_saturation_repair_inside_step(kernel):
    """Drive the serial repair from legal states: a strong infiltration flux oversaturates the top cells; the
    excess is pushed upward cell by cell and overflows into surface_excess_water.  (The deficit pass is
    covered by test_saturation_repair_cases_on_device: a cell driven to sat = 0 has psi = -Inf under
    BrooksCorey, in the reference as well.)  BrooksCorey + linear K: bit-exact against the oracle."""
    lat, lon = small_columns(96)
    w = W.make_workload("richards", lat, lon, 32)
    flux = np.where(np.arange(96) % 2 == 0, -4.0e-4, 0.0)   # m/s; negative = downward (infiltration)
    w["bcs"][("saturation_water_ice", "top")] = ("flux", flux)
    orc, dev = W.setup_oracle(w), W.setup_device(w)
    dev.set_option("step_kernel", kernel)
    orc.run(w["dt"], 6)
    dev.step(w["dt"], 6, True)
    assert orc.get("surface_excess_water").max() > 0           # the overflow path ran
    assert orc.get("saturation_water_ice").max() == 1.0
    assert_fields_match(dev, orc, ["saturation_water_ice", "surface_excess_water", "water_table", "pressure_head",
                                   "internal_energy", "temperature"], True, 0.0, "repair-in-step ")


def test_zero_steps_and_clock():
    lat, lon = small_columns(10)
    w = W.make_workload("heat", lat, lon, 20)
    dev = W.setup_device(w)
    before = dev.temperature
    dev.step(300.0, 0, False)
    assert dev.clock() == (0.0, 0)
    assert np.array_equal(before, dev.temperature)
    dev.step(300.0, 2, True)
    dev.step(900.0, 1, True)
    assert dev.clock() == (1500.0, 3)


def test_status_flags_replace_asserts():
    """SoilVolume's @assert bounds (soil_volume.jl:26-28) -> TRM_STATUS_COMPOSITION_OUT_OF_RANGE."""
    lat, lon = small_columns(10)
    w = W.make_workload("heat", lat, lon, 20)
    dev = W.setup_device(w)
    assert dev.status() == 0
    dev.set("saturation_water_ice", 2.0)
    dev.compute_auxiliary()
    assert dev.status() & trm._capi.STATUS_COMPOSITION
    dev2 = W.setup_device(w)
    bad = w["fields"]["temperature"].copy()
    dev2.set("internal_energy", np.full_like(bad, np.nan))
    dev2.step(300.0, 1, True)
    assert dev2.status() & trm._capi.STATUS_NAN


def test_reductions_match_numpy():
    lat, lon = small_columns(1000)
    w = W.make_workload("richards", lat, lon, 32)
    dev = W.setup_device(w)
    dev.step(w["dt"], 5, True)
    T = dev.temperature.astype(np.float64)
    assert np.allclose(dev.reduce("temperature", "sum"), T.sum(axis=1), rtol=1e-12)
    assert np.array_equal(dev.reduce("temperature", "min"), T.min(axis=1))
    assert np.array_equal(dev.reduce("temperature", "max"), T.max(axis=1))
    assert np.all(dev.reduce("temperature", "hasnan") == 0)
    g = dev._grid_arrays()
    sat = dev.saturation_water_ice.astype(np.float64)
    assert np.allclose(dev.reduce("saturation_water_ice", "volume_integral_z")[0], (sat * g["dzc"][:, None]).sum(),
                       rtol=1e-12)
    assert np.allclose(dev.reduce("water_table", "sum")[0], dev.water_table.sum(), rtol=1e-12)


def test_missing_gpu_path_is_loud():
    """No silent fallback: an invalid device ordinal fails with an error, never computes on the host."""
    grid = trm.ColumnGrid(trm.ExponentialSpacing(N=10), 4, device=99)
    with pytest.raises(trm.TerrariumHipError):
        trm.DeviceState(grid, trm._capi.default_params())


def test_deep_columns_take_the_unfused_kernels():
    """Nz > 64 does not fit the lane = level mapping: trm_step transparently uses the reference-order kernels."""
    lat, lon = small_columns(40)
    w = W.make_workload("richards", lat, lon, 100)
    orc, dev = W.setup_oracle(w), W.setup_device(w)
    orc.run(w["dt"], 20)
    dev.step(w["dt"], 20, True)
    assert_fields_match(dev, orc, W.compared_fields(w), True, 0.0, "Nz=100 ")


@pytest.mark.parametrize("kernel", ["fused", "unfused"])
@pytest.mark.parametrize("heun", [False, True])
def test_per_cell_vwc_forcing(kernel, heun):
    """The user `vwc_forcing` (soil_hydrology.jl:37-38; K15, soil_hydrology_tests.jl:195-232) as a per-cell field:
    a root-zone sink profile that differs per column.  BrooksCorey + linear K => bit-exact."""
    lat, lon = small_columns(70)
    w = W.make_workload("richards", lat, lon, 32)
    orc, dev = W.setup_oracle(w), W.setup_device(w)
    dev.set_option("step_kernel", kernel)
    zc = dev.z_centers()
    F = -2.0e-7 * np.exp(zc / 0.5)[:, None] * (1.0 + 0.5 * np.cos(np.arange(70)))[None, :]
    dev.set("vwc_forcing", F)
    orc.set("vwc_forcing", F)
    if heun:
        for k in range(6):
            orc.timestep_heun(w["dt"], True)
        dev.step_heun(w["dt"], 6, True)
    else:
        orc.run(w["dt"], 20)
        dev.step(w["dt"], 20, True)
    assert_fields_match(dev, orc, W.compared_fields(w), True, 0.0, "vwc forcing field ")
    # the field is really used, and the option switches back to the scalar
    ref = W.setup_device(w)
    (ref.step_heun(w["dt"], 6, True) if heun else ref.step(w["dt"], 20, True))
    assert not np.array_equal(ref.get("saturation_water_ice"), dev.get("saturation_water_ice"))
    dev2 = W.setup_device(w)
    dev2.set("vwc_forcing", F)
    dev2.set_option("vwc_forcing_field", 0)
    (dev2.step_heun(w["dt"], 6, True) if heun else dev2.step(w["dt"], 20, True))
    assert np.array_equal(ref.get("saturation_water_ice"), dev2.get("saturation_water_ice"))


def test_land_surface_inputs_follow_external_state_changes():
    """The fused LandModel step hands (T, sat, liq) of the top cell to the next surface-energy-balance launch through
    compact arrays; anything else that writes the state (upload, invclosure, closure, the process interface) must
    make the next launch read the fields again.  Sequence mirrored on the oracle."""
    lat, lon = small_columns(77)
    w = W.make_workload("land", lat, lon, 32)
    orc, dev = W.setup_oracle(w), W.setup_device(w)
    names = W.compared_fields(w)

    def both(f):
        f(orc)
        f(dev)

    orc.steps(w["dt"], 4)
    dev.step(w["dt"], 4, False)
    T2 = dev.get("temperature") + np.linspace(-3.0, 3.0, 77)[None, :]
    both(lambda o: (o.set("temperature", T2), o.invclosure()))                 # new temperature -> new internal energy
    orc.steps(w["dt"], 3)
    dev.step(w["dt"], 3, False)
    sat2 = np.clip(dev.get("saturation_water_ice") * 0.9, 0.05, 1.0)
    both(lambda o: (o.set("saturation_water_ice", sat2), o.closure()))
    orc.run(w["dt"], 3)
    dev.step(w["dt"], 3, True)
    assert_fields_match(dev, orc, names, False, TOL64, "external writes ")
    # process interface in between (unfused kernels), then fused steps again
    both(lambda o: (o.update_state(True), o.explicit_step(w["dt"]), o.closure()))
    dev.set_clock(*orc.clock())
    orc.run(w["dt"], 2)
    dev.step(w["dt"], 2, True)
    assert_fields_match(dev, orc, names, False, TOL64, "after process interface ")


def test_create_rejects_fields_beyond_the_32bit_offset_range():
    """The step kernel addresses a field with 32-bit byte offsets: one field must stay below 4 GiB per context
    (8.4 M columns x 64 levels in fp64); trm_create says so instead of wrapping around."""
    p = trm._capi.default_params()
    with pytest.raises(trm.TerrariumHipError, match="4 GiB"):
        trm.DeviceState(trm.ColumnGrid(trm.UniformSpacing(dz=0.1, N=32), 2 ** 24), p)          # 2^24 * 32 * 8 B = 4 GiB
    with pytest.raises(trm.TerrariumHipError, match="4 GiB"):
        trm.DeviceState(trm.ColumnGrid(trm.UniformSpacing(dz=0.1, N=64), 2 ** 24, dtype=np.float32), p)
    with pytest.raises(trm.TerrariumHipError):
        trm.DeviceState(trm.ColumnGrid(trm.UniformSpacing(dz=0.1, N=1), 4), p)                 # Nz >= 2


def test_save_and_restore_state_on_device():
    """trm_save_state / trm_restore_state: a device-side checkpoint of every field, the clock and the status word."""
    lat, lon = small_columns(50)
    w = W.make_workload("land", lat, lon, 20)
    dev = W.setup_device(w)
    with pytest.raises(trm.TerrariumHipError):
        dev.restore_state()
    dev.step(w["dt"], 3, False)
    dev.save_state()
    dev.step(w["dt"], 7, True)
    a = {n: dev.get(n) for n in W.compared_fields(w)}
    clock_a = dev.clock()
    dev.restore_state()
    assert dev.clock() == (3 * w["dt"], 3)
    dev.step(w["dt"], 7, True)
    assert dev.clock() == clock_a
    for n, v in a.items():
        assert np.array_equal(dev.get(n), v, equal_nan=True), n


@pytest.mark.parametrize("hydraulics", ["default", "vg"])
@pytest.mark.parametrize("config,Nz,Nh", [("heat", 20, 101), ("richards", 32, 130), ("richards", 64, 77), ("land", 64, 64), ("land", 20, 33),
                                          ("richards", 5, 1), ("land", 32, 2)])
def test_packed_fp32_equals_scalar_bitwise(config, Nz, Nh, hydraulics):
    """fp32 (reference-default or van Genuchten hydraulics) steps two columns per lane with packed instructions
    (trm_packed_f32.hpp): every operation is the scalar kernel's, so the results are the scalar kernel's bit for bit --
    odd column counts, partly filled waves, flux boundary conditions, the saturation repair and LandModel included."""
    lat, lon = small_columns(max(Nh, 2))
    lat, lon = lat[:Nh], lon[:Nh]
    if config == "heat" and hydraulics == "vg":
        pytest.skip("hydraulics do not enter the heat-only configuration")
    w = W.make_workload(config, lat, lon, Nz, dtype=np.float32, hydraulics=hydraulics)
    if config == "richards":
        w["bcs"][("saturation_water_ice", "top")] = ("flux", np.where(np.arange(Nh) % 3 == 0, -3.0e-4, 0.0))
        w["bcs"][("internal_energy", "bottom")] = ("flux", np.full(Nh, 0.05))
        w["bcs"][("temperature", "bottom")] = ("value", w["T0"] - 1.0)
    a, b = W.setup_device(w), W.setup_device(w)
    assert a.get_option("packed_f32") == 1
    b.set_option("packed_f32", 0)
    for nsteps, fin in ((1, False), (7, False), (12, True)):
        a.step(w["dt"], nsteps, fin)
        b.step(w["dt"], nsteps, fin)
        for n in W.compared_fields(w):
            x, y = a.get(n), b.get(n)
            assert np.array_equal(x, y, equal_nan=True), (n, nsteps, np.argwhere(x != y)[:4])
    assert a.status() == b.status() and a.clock() == b.clock()


@pytest.mark.parametrize("hydraulics", ["default", "vg"])
@pytest.mark.parametrize("config,Nz,Nh", [("richards", 32, 130), ("land", 64, 77), ("land", 20, 33)])
def test_packed_fp32_with_the_pressure_head_derived_equals_the_stored_one_bitwise(config, Nz, Nh, hydraulics):
    """TRM_OPT_DERIVE_CLOSURE_FIELDS = 4: the packed fp32 step re-derives the liquid fraction AND the pressure head (from the stored
    saturation and water table) instead of reading them -- the same function of the same operands the previous step stored, so the
    same bits; an upload of the pressure head / the water table / the saturation makes the next step read the fields again."""
    lat, lon = small_columns(Nh)
    w = W.make_workload(config, lat, lon, Nz, dtype=np.float32, hydraulics=hydraulics)
    a, b = W.setup_device(w), W.setup_device(w)
    a.set_option("derive_closure_fields", 4)
    b.set_option("derive_closure_fields", 0)
    for d in (a, b):
        d.step(w["dt"], 1, False)
        d.step(w["dt"], 9, False)
    psi = b.get("pressure_head")
    psi[2] += 0.125                                   # a user edit of the closure field: the step after it must READ it
    for d in (a, b):
        d.set("pressure_head", psi)
        d.step(w["dt"], 1, False)
        d.step(w["dt"], 6, True)
    for n in W.compared_fields(w):
        assert np.array_equal(a.get(n), b.get(n), equal_nan=True), n
    assert a.status() == b.status() and a.clock() == b.clock()


def test_derivation_modes_the_fp64_column_program_no_longer_has_select_what_it_offers():
    """TRM_OPT_DERIVE_CLOSURE_FIELDS = 3 / 4 / 5 on an fp64 context: the instances for "liquid fraction alone" and "pressure head as
    well" were measured slower than deriving T and liq (EXPERIMENTS.md) and removed in round 5; the values stay legal and select the
    derivation the column program offers -- same bits as reading the fields."""
    lat, lon = small_columns(140)
    w = W.make_workload("richards", lat, lon, 32)
    ref = W.setup_device(w)
    ref.set_option("derive_closure_fields", 0)
    ref.step(w["dt"], 12, True)
    for mode in (3, 4, 5):
        d = W.setup_device(w)
        d.set_option("derive_closure_fields", mode)
        d.step(w["dt"], 12, True)
        assert d.last_program()["derive"] == "T_liq"
        for n in W.compared_fields(w):
            assert np.array_equal(d.get(n), ref.get(n), equal_nan=True), (mode, n)


def test_external_stream_and_async_option():
    """trm_set_stream (a torch / HIP stream owned by the caller) + TRM_OPT_ASYNC: the launches are only enqueued, the
    caller synchronises; results equal the synchronous run on the context's own stream."""
    import torch
    lat, lon = small_columns(300)
    w = W.make_workload("land", lat, lon, 32)
    a, b = W.setup_device(w), W.setup_device(w)
    stream = torch.cuda.Stream()
    b.set_stream(stream.cuda_stream)
    b.set_option("asynchronous", 1)
    a.step(w["dt"], 20, True)
    b.step(w["dt"], 20, True)          # returns after enqueueing
    stream.synchronize()
    for n in W.compared_fields(w):
        assert np.array_equal(a.get(n), b.get(n), equal_nan=True), n
    # events recorded on that stream see the library's kernels (what bench.py relies on)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        e0.record()
        b.step(w["dt"], 20, False)
        e1.record()
    e1.synchronize()
    assert e0.elapsed_time(e1) > 0.05   # 40 launches cannot take less than this; 0 would mean the events saw nothing
    b.set_stream(None)
    b.set_option("asynchronous", 0)
    b.step(w["dt"], 1, True)


# ---- state.tendencies after a step (the fused step must not leave stale tendency fields) -----------------
@pytest.mark.parametrize("integrator", ["euler", "heun"])
@pytest.mark.parametrize("config,hydraulics,dtype", [("heat", "default", np.float64), ("richards", "default", np.float64),
                                                     ("land", "vg", np.float64), ("land", "default", np.float32)])
def test_tendency_fields_after_finalizing_step(config, hydraulics, dtype, integrator):
    """After timestep!(integrator, dt) the reference's state.tendencies hold the last step's tendencies incl. the
    compute_z_bcs! term (averaged over the stages for Heun).  Fused (finalize = 1) == unfused == oracle."""
    lat, lon = small_columns(130)
    w = W.make_workload(config, lat, lon, 64 if np.dtype(dtype) == np.float32 else 32, dtype=dtype, hydraulics=hydraulics)
    if config == "heat":   # a flux condition at the bottom as well, so that compute_z_bcs! has something to add
        w["bcs"][("internal_energy", "bottom")] = ("flux", np.full(lat.size, 0.05))
    a, b, orc = W.setup_device(w), W.setup_device(w), W.setup_oracle(w)
    b.set_option("step_kernel", "unfused")
    names = ["tend_internal_energy"] + (["tend_saturation_water_ice", "tend_surface_excess_water"] if config != "heat" else [])
    for d in (a, b):
        (d.step_heun if integrator == "heun" else d.step)(w["dt"], 7, finalize=True)
    for n in range(7):   # run!(steps = 7): compute_auxiliary! once, after the last step (it advances the skin temperature)
        (orc.timestep_heun if integrator == "heun" else orc.timestep)(w["dt"], n == 6)
    for n in names:
        assert np.array_equal(a.get(n), b.get(n)), n
        assert np.any(a.get(n) != 0) or n == "tend_surface_excess_water", n
    exact = bit_exact_config(config, hydraulics, dtype)
    assert_fields_match(a, orc, names, exact, TOL64 if np.dtype(dtype) == np.float64 else TOL32)


def test_tendency_fields_are_refused_when_not_materialised():
    lat, lon = small_columns(70)
    w = W.make_workload("richards", lat, lon, 32)
    d = W.setup_device(w)
    d.step(w["dt"], 3, finalize=False)
    for call in (lambda: d.get("tend_internal_energy"), lambda: d.reduce("tend_saturation_water_ice", "sum")):
        with pytest.raises(trm.TerrariumHipError) as e:
            call()
        assert e.value.code == trm._capi.TRM_ESTALE
    d.update_state(True)            # materialises them again
    assert np.all(np.isfinite(d.get("tend_internal_energy")))
    d.step(w["dt"], 1, finalize=True)
    assert np.all(np.isfinite(d.get("tend_saturation_water_ice")))
    d.set_option("step_kernel", "unfused")
    d.step(w["dt"], 1, finalize=False)
    assert np.all(np.isfinite(d.get("tend_internal_energy")))


def test_set_bc_replaces_a_series_and_nan_propagating_minmax():
    lat, lon = small_columns(65)
    w = W.make_workload("heat", lat, lon, 20)
    d, orc = W.setup_device(w), W.setup_oracle(w)
    times = np.array([0.0, 1.0e6])
    d.set_bc_series("temperature", "top", "value", times, np.stack([np.full(65, -30.0), np.full(65, -30.0)]))
    d.set_bc("temperature", "top", "value", w["bcs"][("temperature", "top")][1])   # the constant wins from now on
    d.step(w["dt"], 20, finalize=True)
    orc.run(w["dt"], 20)
    assert np.array_equal(d.get("temperature"), orc.get("temperature"))
    # minimum / maximum propagate NaN as Julia's do
    T = d.get("temperature")
    T[3, 7] = np.nan
    d.set("temperature", T)
    mn, mx = d.reduce("temperature", "min"), d.reduce("temperature", "max")
    assert np.isnan(mn[3]) and np.isnan(mx[3])
    keep = np.arange(20) != 3
    assert np.array_equal(mn[keep], T[keep].min(axis=1)) and np.array_equal(mx[keep], T[keep].max(axis=1))


def test_upload_download_round_trip_layouts():
    """trm_upload / trm_download transpose on the device: every field kind, ragged sizes, both precisions."""
    rng = np.random.default_rng(5)
    for dtype, Nh, Nz in ((np.float64, 1, 2), (np.float64, 333, 20), (np.float32, 65, 64), (np.float64, 31, 70), (np.float32, 1000, 33)):
        p = trm._capi.default_params()
        p.flow = 1
        d = trm.DeviceState(trm.ColumnGrid(trm.UniformSpacing(dz=0.1, N=Nz), Nh, dtype=dtype), p)
        for name in ("temperature", "hydraulic_conductivity", "surface_excess_water", "tend_saturation_water_ice"):
            a = rng.standard_normal((d.rows(name), Nh)).astype(dtype)
            d.set(name, a)
            b = d.get(name)
            assert np.array_equal(b if b.ndim == 2 else b[None, :], a), (name, dtype, Nh, Nz)
        d.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("Nh", [28476, 28474, 28475])
def test_no_spurious_status_flags_from_tail_lanes(Nh, dtype):
    """Lanes beyond the last column carry a clamped copy of it that the saturation repair skips: when that column is
    oversaturated before the repair, the copy must not raise the composition flag (round-2 regression: shards of 28 476
    columns reported TRM_STATUS_COMPOSITION_OUT_OF_RANGE while the full grid and the reference-order kernels did not)."""
    lat, lon = W.columns_from_mask("N145")
    w = W.make_workload("richards", lat[:Nh], lon[:Nh], 32, dtype=dtype)
    fused, unfused = W.setup_device(w), W.setup_device(w)
    unfused.set_option("step_kernel", "unfused")
    for d in (fused, unfused):
        d.step(w["dt"], 10, finalize=True)
    assert unfused.status() == 0 and fused.status() == 0
    fused.step_heun(w["dt"], 3, finalize=True)
    fused.set_option("steps_per_launch", 4)
    fused.step(w["dt"], 8, finalize=True)
    assert fused.status() == 0
