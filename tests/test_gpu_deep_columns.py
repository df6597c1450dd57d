"""Columns of 65 ... 128 levels on the fused path (csrc/trm_column_deep.hpp: two soil levels per lane).  The reference's own
saturation-adjustment test uses UniformSpacing(N = 100) (test/soil/soil_hydrology_tests.jl:93-123); such grids used to take the
reference-order kernels.  k_column_deep must agree BIT FOR BIT with them and with the oracle (heat, heat + Richards with the
reference-default hydraulics), to 1e-10 with van Genuchten / the LandModel."""
import numpy as np
import pytest

import workloads as W
import terrarium_jl_amd as trm

pytestmark = pytest.mark.gpu


def small_columns(n):
    lat, lon = W.columns_from_mask("N72")
    sel = np.linspace(0, lat.size - 1, n).astype(int)
    return lat[sel], lon[sel]


CASES = [("heat", "default", np.float64, 65, 37), ("richards", "default", np.float64, 100, 101), ("richards", "default", np.float64, 127, 5),
         ("richards", "vg", np.float64, 128, 64), ("land", "default", np.float64, 96, 130), ("land", "vg", np.float64, 99, 33),
         ("richards", "default", np.float32, 100, 50), ("land", "vg", np.float32, 66, 41), ("heat", "default", np.float32, 128, 9)]


@pytest.mark.parametrize("config,hydraulics,dtype,Nz,Nh", CASES)
def test_deep_columns_fused_equals_reference_order_kernels_bitwise(config, hydraulics, dtype, Nz, Nh):
    lat, lon = small_columns(Nh)
    w = W.make_workload(config, lat, lon, Nz, dtype=dtype, hydraulics=hydraulics)
    if config == "heat":
        w["bcs"][("internal_energy", "bottom")] = ("flux", np.full(lat.size, 0.05))
        w["bcs"][("temperature", "bottom")] = ("value", np.full(lat.size, 1.5))
    if config == "richards":
        w["bcs"][("saturation_water_ice", "top")] = ("flux", np.full(lat.size, -1.0e-7))
    a, b = W.setup_device(w), W.setup_device(w)
    a.set_option("derive_closure_fields", 1 if Nz % 2 else 0)     # (with and without T / liq derived in registers)
    b.set_option("step_kernel", "unfused")
    nsteps = 30 if config != "land" else 20
    for d in (a, b):
        d.step(w["dt"], 1, finalize=False)
        d.step(w["dt"], nsteps, finalize=False)
        d.step(w["dt"], 1, finalize=True)
    names = W.compared_fields(w) + ["tend_internal_energy"] + (["tend_saturation_water_ice", "tend_surface_excess_water"] if config != "heat" else [])
    for n in names:
        assert np.array_equal(a.get(n), b.get(n), equal_nan=True), n
    assert a.status() == b.status() and a.clock() == b.clock()
    if dtype == np.float64 and hydraulics == "default" and config != "land":
        o = W.setup_oracle(w)
        o.run(w["dt"], nsteps + 2)
        for n in W.compared_fields(w):
            assert np.array_equal(a.get(n), o.get(n)), n


# test/soil/soil_hydrology_tests.jl:93-123 (K8) on its own grid, UniformSpacing(N = 100), now through the fused step: the
# repair runs inside k_column_deep when a step produces the out-of-range profiles
def test_saturation_adjustment_on_the_reference_grid_inside_the_fused_step():
    import oracle
    grid = trm.ColumnGrid(trm.UniformSpacing(dz=0.1, N=100), 12)
    rng = np.random.default_rng(2)
    sat = np.clip(0.97 + 0.02 * rng.normal(size=(100, 12)), 0.0, 1.0)
    sat[-1] = 0.999
    sat[40:45, ::2] = 1.0
    integ = trm.initialize(trm.SoilModel(grid, soil=trm.SoilEnergyWaterCarbon(hydrology=trm.SoilHydrology(vertical_flow=trm.RichardsEq()))),
                           trm.ForwardEuler(dt=60.0), boundary_conditions=trm.InfiltrationFlux(-5.0e-5),
                           initializers=dict(temperature=3.0, saturation_water_ice=sat))
    st = integ.state
    st.set_option("steps_per_launch", 1)
    o = oracle.Oracle(12, grid.thickness, oracle.default_params(flow=1))
    o.set("temperature", 3.0); o.set("saturation_water_ice", sat)
    o.set_bc("saturation_water_ice", "top", "flux", -5.0e-5)
    o.initialize()
    trm.run(integ, steps=25)
    o.run(60.0, 25)
    assert np.any(st.surface_excess_water > 0)                  # the strong infiltration overflows the top cell: the repair has run
    s = st.saturation_water_ice
    assert s.min() >= 0.0 and s.max() <= 1.0
    for n in ("saturation_water_ice", "surface_excess_water", "water_table", "pressure_head", "internal_energy", "temperature", "hydraulic_conductivity"):
        assert np.array_equal(st.get(n), o.get(n)), n


# Value / Flux / Gradient boundary kinds on every variable that carries them (the reference's FreeDrainage() is a Gradient condition on
# the pressure head) and the per-cell vwc_forcing field: the GENERIC instance of k_column_deep, against the reference-order kernels
# and the oracle, bit for bit
@pytest.mark.parametrize("config,dtype,Nz,Nh", [("richards", np.float64, 100, 83), ("richards", np.float64, 127, 11), ("heat", np.float64, 80, 20),
                                                 ("richards", np.float32, 66, 40), ("land", np.float64, 96, 30)])
def test_deep_columns_with_generic_boundary_kinds_run_fused_and_equal_the_reference_order_kernels(config, dtype, Nz, Nh):
    lat, lon = small_columns(Nh)
    w = W.make_workload(config, lat, lon, Nz, dtype=dtype)
    rng = np.random.default_rng(7)
    if config == "heat":
        w["bcs"][("temperature", "bottom")] = ("gradient", 0.01)
        w["bcs"][("liquid_water_fraction", "top")] = ("gradient", 0.1)
    else:
        w["bcs"].update({("temperature", "bottom"): ("value", w["T0"] - 1.0), ("internal_energy", "bottom"): ("flux", np.full(Nh, 0.05)),
                         ("pressure_head", "bottom"): ("gradient", 0.0), ("liquid_water_fraction", "top"): ("gradient", 0.1)})
        if config == "richards":
            w["bcs"][("saturation_water_ice", "top")] = ("flux", -1.0e-8 * rng.random(Nh))
            w["bcs"][("pressure_head", "top")] = ("value", np.full(Nh, -0.3))
    a, b = W.setup_device(w), W.setup_device(w)
    b.set_option("step_kernel", "unfused")
    o = W.setup_oracle(w) if (dtype == np.float64 and config != "land") else None
    if config != "heat":        # a root-zone sink that differs per cell and column (soil_hydrology.jl:37-38)
        zc = a.z_centers()
        F = (-2.0e-7 * np.exp(zc / 0.5)[:, None] * (1.0 + 0.5 * np.cos(np.arange(Nh)))[None, :]).astype(dtype)
        for d in (a, b) + ((o,) if o is not None else ()):
            d.set("vwc_forcing", F)
    nsteps = 25
    for d in (a, b):
        d.step(w["dt"], 1, finalize=False)
        d.step(w["dt"], nsteps - 2, finalize=False)
        d.step(w["dt"], 1, finalize=True)
    names = W.compared_fields(w) + ["tend_internal_energy"] + (["tend_saturation_water_ice", "tend_surface_excess_water"] if config != "heat" else [])
    for n in names:
        assert np.array_equal(a.get(n), b.get(n), equal_nan=True), n
    assert a.status() == b.status() and a.clock() == b.clock()
    if o is not None:
        o.run(w["dt"], nsteps)
        for n in W.compared_fields(w):
            assert np.array_equal(a.get(n), o.get(n)), n
    # it really is the fused kernel: one launch per step against a dozen
    ta, tb = a.step_timed(w["dt"], 20, finalize=False), b.step_timed(w["dt"], 20, finalize=False)
    assert ta < 0.6 * tb, (ta, tb)


# heun.jl:37-71 on deep columns: both stages of k_column_deep<PROG_HEUN> in registers, one launch per step -- bit for bit the
# reference-order (staged) Heun and the oracle's, incl. boundary series evaluated at t for the state and t + dt for the stage
HEUN_CASES = [("heat", "default", np.float64, 100, 60), ("richards", "default", np.float64, 100, 75), ("richards", "vg", np.float64, 127, 9),
              ("land", "default", np.float64, 96, 40), ("richards", "default", np.float32, 128, 33), ("land", "vg", np.float32, 65, 21)]


@pytest.mark.parametrize("config,hydraulics,dtype,Nz,Nh", HEUN_CASES)
def test_deep_columns_heun_in_one_launch_equals_the_staged_heun_bitwise(config, hydraulics, dtype, Nz, Nh):
    lat, lon = small_columns(Nh)
    w = W.make_workload(config, lat, lon, Nz, dtype=dtype, hydraulics=hydraulics)
    if config == "richards":
        w["bcs"][("saturation_water_ice", "top")] = ("flux", np.full(lat.size, -1.0e-7))
    a, b = W.setup_device(w), W.setup_device(w)
    a.set_option("derive_closure_fields", 1 if Nz % 2 else 0)
    b.set_option("step_kernel", "unfused")
    o = W.setup_oracle(w) if (dtype == np.float64 and hydraulics == "default" and config != "land") else None
    if config != "land":      # a time-dependent top temperature: the stage takes its value at t + dt
        times = w["dt"] * np.arange(0, 16)
        vals = np.stack([w["T0"] + 10.0 * np.sin(2 * np.pi * t / 86400.0 - lon) + 0.01 * t / w["dt"] for t in times])
        for d in (a, b) + ((o,) if o is not None else ()):
            d.set_bc_series("temperature", "top", "value", times, vals)
    nsteps = 12
    for d in (a, b):
        d.step_heun(w["dt"], 1, finalize=False)
        d.step_heun(w["dt"], nsteps - 2, finalize=False)
        d.step_heun(w["dt"], 1, finalize=True)
    names = W.compared_fields(w) + ["tend_internal_energy"] + (["tend_saturation_water_ice", "tend_surface_excess_water"] if config != "heat" else [])
    for n in names:
        assert np.array_equal(a.get(n), b.get(n), equal_nan=True), n
    assert a.status() == b.status() and a.clock() == b.clock()
    if o is not None:
        for k in range(nsteps):
            o.timestep_heun(w["dt"], True)
        for n in W.compared_fields(w):
            assert np.array_equal(a.get(n), o.get(n)), n


# model_integrator.jl:72-88 (run!'s loop) on deep columns: the resident multi-step program of k_column_deep -- the library's default
# for nsteps > 1 without the surface energy balance and without series -- equals one launch per step bit for bit
@pytest.mark.parametrize("config,hydraulics,dtype,Nz,Nh", [("heat", "default", np.float64, 100, 70), ("richards", "default", np.float64, 100, 90),
                                                            ("richards", "vg", np.float64, 65, 17), ("richards", "default", np.float32, 128, 40)])
def test_deep_columns_multistep_program_equals_per_step_launches_bitwise(config, hydraulics, dtype, Nz, Nh):
    lat, lon = small_columns(Nh)
    w = W.make_workload(config, lat, lon, Nz, dtype=dtype, hydraulics=hydraulics)
    if config == "richards":
        w["bcs"][("saturation_water_ice", "top")] = ("flux", np.full(lat.size, -2.0e-7))
    a, b, c = W.setup_device(w, steps_per_launch=0), W.setup_device(w, steps_per_launch=1), W.setup_device(w, steps_per_launch=7)
    for d in (a, b, c):
        d.step(w["dt"], 23, finalize=False)
        d.step(w["dt"], 10, finalize=True)
    names = W.compared_fields(w) + ["tend_internal_energy"] + (["tend_saturation_water_ice", "tend_surface_excess_water", "surface_excess_water", "water_table"] if config != "heat" else [])
    for n in names:
        assert np.array_equal(a.get(n), b.get(n), equal_nan=True), n
        assert np.array_equal(c.get(n), b.get(n), equal_nan=True), n
    assert a.status() == b.status() == c.status() and a.clock() == b.clock() == c.clock()
    # the default really is the program: one launch for 40 steps is much shorter than 40 launches
    a.synchronize(); b.synchronize()
    ta, tb = a.step_timed(w["dt"], 40, finalize=False), b.step_timed(w["dt"], 40, finalize=False)
    assert ta < 0.8 * tb, (ta, tb)


# heun.jl:37-71 with EVERY boundary kind on deep columns (the reference's FreeDrainage() -- soil_model_bcs.jl:40 -- on its own
# UniformSpacing(N = 100) grid, soil_hydrology_tests.jl:93-123): k_column_deep<PROG_HEUN, GENERIC>, one launch per step, the stage's
# boundary values from the stage's view (series evaluated at t + dt) -- bit for bit the staged reference-order Heun and the oracle
@pytest.mark.parametrize("config,dtype,Nz,Nh", [("richards", np.float64, 100, 83), ("richards", np.float64, 127, 11), ("heat", np.float64, 80, 20),
                                                 ("richards", np.float32, 66, 40), ("land", np.float64, 96, 30)])
def test_deep_columns_heun_with_generic_boundary_kinds_runs_fused(config, dtype, Nz, Nh):
    lat, lon = small_columns(Nh)
    w = W.make_workload(config, lat, lon, Nz, dtype=dtype)
    rng = np.random.default_rng(9)
    if config == "heat":
        w["bcs"][("temperature", "bottom")] = ("gradient", 0.01)
        w["bcs"][("liquid_water_fraction", "top")] = ("gradient", 0.1)
    else:
        w["bcs"].update({("temperature", "bottom"): ("value", w["T0"] - 1.0), ("internal_energy", "bottom"): ("flux", np.full(Nh, 0.05)),
                         ("pressure_head", "bottom"): ("gradient", 0.0), ("liquid_water_fraction", "top"): ("gradient", 0.1)})   # FreeDrainage()
        if config == "richards":
            w["bcs"][("saturation_water_ice", "top")] = ("flux", -1.0e-8 * rng.random(Nh))
            w["bcs"][("pressure_head", "top")] = ("value", np.full(Nh, -0.3))
    a, b = W.setup_device(w), W.setup_device(w)
    b.set_option("step_kernel", "unfused")
    o = W.setup_oracle(w) if (dtype == np.float64 and config != "land") else None
    times = w["dt"] * np.arange(0, 16)
    if config != "land":      # a time-dependent bottom temperature value / gradient: the stage takes it at t + dt
        kind = "gradient" if config == "heat" else "value"
        vals = np.stack([(0.01 if config == "heat" else w["T0"] - 1.0) + 0.002 * (t / w["dt"]) * np.ones(Nh) for t in times])
        for d in (a, b) + ((o,) if o is not None else ()):
            d.set_bc_series("temperature", "bottom", kind, times, vals)
    if config != "heat":
        zc = a.z_centers()
        F = (-2.0e-7 * np.exp(zc / 0.5)[:, None] * (1.0 + 0.5 * np.cos(np.arange(Nh)))[None, :]).astype(dtype)
        for d in (a, b) + ((o,) if o is not None else ()):
            d.set("vwc_forcing", F)
    nsteps = 12
    for d in (a, b):
        d.step_heun(w["dt"], 1, finalize=False)
        d.step_heun(w["dt"], nsteps - 2, finalize=False)
        d.step_heun(w["dt"], 1, finalize=True)
    names = W.compared_fields(w) + ["tend_internal_energy"] + (["tend_saturation_water_ice", "tend_surface_excess_water"] if config != "heat" else [])
    for n in names:
        assert np.array_equal(a.get(n), b.get(n), equal_nan=True), n
    assert a.status() == b.status() and a.clock() == b.clock()
    if o is not None:
        for k in range(nsteps):
            o.timestep_heun(w["dt"], True)
        for n in W.compared_fields(w):
            assert np.array_equal(a.get(n), o.get(n)), n
    ta, tb = a.step_heun_timed(w["dt"], 10, finalize=False), b.step_heun_timed(w["dt"], 10, finalize=False)
    assert ta < 0.5 * tb, (ta, tb)          # one launch per step against the staged sequence


# LandModel(vegetation = VegetationCarbon) on deep columns (land_model.jl:79-97): the soil half of every step is k_column_deep, the
# 0-D half the per-column form of k_surface_veg; Heun stores what the 0-D processes need of the stage -- bit for bit the
# reference-order kernels, ForwardEuler and Heun, fp64 and fp32
# (129 ... 256 levels: k_surface_veg in front of k_column_wide, ForwardEuler and Heun)
@pytest.mark.parametrize("heun", [False, True])
@pytest.mark.parametrize("dtype,Nz,Nh", [(np.float64, 100, 70), (np.float64, 65, 33), (np.float32, 128, 40), (np.float64, 160, 21), (np.float32, 250, 14)])
def test_vegetation_coupled_land_model_on_deep_columns_runs_fused(dtype, Nz, Nh, heun):
    lat, lon = small_columns(Nh)
    w = W.make_workload("landveg", lat, lon, Nz, dtype=dtype, hydraulics="vg")
    a, b = W.setup_device(w), W.setup_device(w)
    b.set_option("step_kernel", "unfused")
    step = (lambda d, n, fin: d.step_heun(w["dt"], n, finalize=fin)) if heun else (lambda d, n, fin: d.step(w["dt"], n, finalize=fin))
    for d in (a, b):
        step(d, 1, False)
        step(d, 10, False)
        step(d, 1, True)
    names = W.compared_fields(w) + ["tend_internal_energy", "tend_saturation_water_ice", "tend_canopy_water", "tend_carbon_vegetation"]
    for n in names:
        assert np.array_equal(a.get(n), b.get(n), equal_nan=True), n
    assert a.status() == b.status() and a.clock() == b.clock()
    timed = (lambda d: d.step_heun_timed(w["dt"], 10, finalize=False)) if heun else (lambda d: d.step_timed(w["dt"], 10, finalize=False))
    ta, tb = timed(a), timed(b)
    assert ta < 0.7 * tb, (ta, tb)


# Columns of 129 ... 256 levels: four levels per lane (csrc/trm_column_wide.hpp), ForwardEuler and Heun in one launch per step, the
# branch-free and the generic boundary kinds -- bit for bit the reference-order kernels and the oracle; ragged level counts (the top
# lane partly filled, an odd number of levels), ragged column counts, fp64 and fp32
WIDE_CASES = [("heat", "default", np.float64, 129, 21), ("richards", "default", np.float64, 200, 37), ("richards", "default", np.float64, 256, 9),
              ("richards", "vg", np.float64, 131, 12), ("land", "default", np.float64, 160, 17), ("richards", "default", np.float32, 254, 15),
              ("land", "vg", np.float32, 130, 10)]


@pytest.mark.parametrize("heun", [False, True])
@pytest.mark.parametrize("generic", [False, True])
@pytest.mark.parametrize("config,hydraulics,dtype,Nz,Nh", WIDE_CASES)
def test_wide_columns_run_fused_and_equal_the_reference_order_kernels_bitwise(config, hydraulics, dtype, Nz, Nh, generic, heun):
    lat, lon = small_columns(Nh)
    w = W.make_workload(config, lat, lon, Nz, dtype=dtype, hydraulics=hydraulics)
    rng = np.random.default_rng(13)
    if generic:
        w["bcs"][("liquid_water_fraction", "top")] = ("gradient", 0.1)
        if config == "heat":
            w["bcs"][("temperature", "bottom")] = ("gradient", 0.01)
        else:
            w["bcs"].update({("temperature", "bottom"): ("value", w["T0"] - 1.0), ("pressure_head", "bottom"): ("gradient", 0.0)})     # FreeDrainage()
            if config == "richards":
                w["bcs"][("pressure_head", "top")] = ("value", np.full(Nh, -0.3))
    else:
        w["bcs"][("internal_energy", "bottom")] = ("flux", np.full(Nh, 0.05))
        if config == "richards":
            w["bcs"][("saturation_water_ice", "top")] = ("flux", -2.0e-7 * rng.random(Nh))
    a, b = W.setup_device(w), W.setup_device(w)
    b.set_option("step_kernel", "unfused")
    o = W.setup_oracle(w) if (dtype == np.float64 and hydraulics == "default" and config != "land") else None
    if config != "land" and not (generic and config == "heat"):      # a time-dependent boundary temperature: the Heun stage takes it at t + dt
        times = w["dt"] * np.arange(0, 20)
        side = "bottom" if generic else "top"
        vals = np.stack([w["T0"] + (-1.0 if generic else 10.0 * np.sin(2 * np.pi * t / 86400.0 - lon)) + 0.01 * t / w["dt"] for t in times])
        for d in (a, b) + ((o,) if o is not None else ()):
            d.set_bc_series("temperature", side, "value", times, vals)
    nsteps = 14
    step = (lambda d, n, fin: d.step_heun(w["dt"], n, finalize=fin)) if heun else (lambda d, n, fin: d.step(w["dt"], n, finalize=fin))
    for d in (a, b):
        step(d, 1, False)
        step(d, nsteps - 2, False)
        step(d, 1, True)
    names = W.compared_fields(w) + ["tend_internal_energy"] + (["tend_saturation_water_ice", "tend_surface_excess_water"] if config != "heat" else [])
    for n in names:
        assert np.array_equal(a.get(n), b.get(n), equal_nan=True), n
    assert a.status() == b.status() and a.clock() == b.clock()
    if o is not None:
        for k in range(nsteps):
            (o.timestep_heun if heun else o.timestep)(w["dt"], True)
        for n in W.compared_fields(w):
            assert np.array_equal(a.get(n), o.get(n)), n
    timed = (lambda d: d.step_heun_timed(w["dt"], 10, finalize=False)) if heun else (lambda d: d.step_timed(w["dt"], 10, finalize=False))
    ta, tb = timed(a), timed(b)
    assert ta < 0.6 * tb, (ta, tb)          # one launch per step against the reference-order sequence


def test_saturation_repair_inside_the_wide_fused_step():
    """The lane-serial repair passes with four levels per lane: strong infiltration into a nearly saturated 200-level column with
    saturated pockets -- oversaturation travels up through lanes and slots, the top overflows (soil_hydrology.jl:185-219)."""
    import oracle
    grid = trm.ColumnGrid(trm.UniformSpacing(dz=0.05, N=200), 11)
    rng = np.random.default_rng(4)
    sat = np.clip(0.97 + 0.02 * rng.normal(size=(200, 11)), 0.0, 1.0)
    sat[-1] = 0.999
    sat[77:83, ::2] = 1.0
    sat[150:153, 1::3] = 1.0
    integ = trm.initialize(trm.SoilModel(grid, soil=trm.SoilEnergyWaterCarbon(hydrology=trm.SoilHydrology(vertical_flow=trm.RichardsEq()))),
                           trm.ForwardEuler(dt=30.0), boundary_conditions=trm.InfiltrationFlux(-5.0e-5),
                           initializers=dict(temperature=3.0, saturation_water_ice=sat))
    st = integ.state
    st.set_option("steps_per_launch", 1)
    o = oracle.Oracle(11, grid.thickness, oracle.default_params(flow=1))
    o.set("temperature", 3.0); o.set("saturation_water_ice", sat)
    o.set_bc("saturation_water_ice", "top", "flux", -5.0e-5)
    o.initialize()
    trm.run(integ, steps=25)
    o.run(30.0, 25)
    assert np.any(st.surface_excess_water > 0)
    s = st.saturation_water_ice
    assert s.min() >= 0.0 and s.max() <= 1.0
    for n in ("saturation_water_ice", "surface_excess_water", "water_table", "pressure_head", "internal_energy", "temperature", "hydraulic_conductivity"):
        assert np.array_equal(st.get(n), o.get(n)), n
