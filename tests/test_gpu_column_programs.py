"""The register-resident column programs (csrc/trm_column.hpp: k_column) against the per-step kernels they replace
and against the oracle: ForwardEuler with temperature / liquid fraction derived in registers, Heun in one launch, and
`m` steps per launch with the column held in registers.  All three must be BIT-IDENTICAL to the reference-order
kernels (one launch per reference kernel) -- they perform the same operations in the same order."""
import numpy as np
import pytest

import workloads as W
import terrarium_jl_amd as trm

pytestmark = pytest.mark.gpu


def small_columns(n, name="N72"):
    lat, lon = W.columns_from_mask(name)
    sel = np.linspace(0, lat.size - 1, n).astype(int)
    return lat[sel], lon[sel]


def all_fields(w):
    return W.compared_fields(w)


CONFIGS = [("heat", "default", np.float64, 30), ("richards", "default", np.float64, 32), ("richards", "vg", np.float64, 20),
           ("land", "default", np.float64, 32), ("land", "vg", np.float64, 50), ("richards", "vg", np.float32, 64),
           ("land", "vg", np.float32, 40), ("heat", "default", np.float32, 20)]


@pytest.mark.parametrize("config,hydraulics,dtype,Nz", CONFIGS)
def test_euler_program_equals_reference_order_kernels_bitwise(config, hydraulics, dtype, Nz):
    """k_column<PROG_EULER>, reading T / liq on its first step and deriving them from (U, sat) afterwards, against the
    reference-order kernels and against the same program with the derivation switched off."""
    lat, lon = small_columns(203)
    w = W.make_workload(config, lat, lon, Nz, dtype=dtype, hydraulics=hydraulics)
    if config == "heat":
        w["bcs"][("internal_energy", "bottom")] = ("flux", np.full(lat.size, 0.05))
        w["bcs"][("temperature", "bottom")] = ("value", np.full(lat.size, 1.5))
    new, noderive, ref = W.setup_device(w), W.setup_device(w), W.setup_device(w)
    new.set_option("derive_closure_fields", 1)
    noderive.set_option("derive_closure_fields", 0)
    ref.set_option("step_kernel", "unfused")
    for d in (new, noderive, ref):
        d.set_option("packed_f32", 0)
        d.step(w["dt"], 1, finalize=False)      # first step: stored T / liq are the user's
        d.step(w["dt"], 23, finalize=False)
        d.step(w["dt"], 1, finalize=True)
    for n in all_fields(w) + ["tend_internal_energy"]:
        a = ref.get(n)
        assert np.array_equal(new.get(n), a, equal_nan=True), n
        assert np.array_equal(noderive.get(n), a, equal_nan=True), n
    assert new.status() == ref.status()


def test_derivation_is_dropped_when_the_state_is_touched():
    """An upload of T between steps makes the stored (T, liq) the user's again: the next step must read them."""
    lat, lon = small_columns(100)
    w = W.make_workload("richards", lat, lon, 32)
    a, b = W.setup_device(w), W.setup_device(w)
    a.set_option("derive_closure_fields", 1)
    b.set_option("step_kernel", "unfused")
    for d in (a, b):
        d.step(w["dt"], 5, finalize=False)
        T = d.get("temperature")
        d.set("temperature", T + 0.25)          # inconsistent with U on purpose
        d.step(w["dt"], 5, finalize=True)
    for n in all_fields(w):
        assert np.array_equal(a.get(n), b.get(n)), n
    # save / restore keeps the bookkeeping straight
    a.save_state(); b.save_state()
    for d in (a, b):
        d.set("internal_energy", d.get("internal_energy") * 1.01)
        d.restore_state()
        d.step(w["dt"], 3, finalize=True)
    for n in all_fields(w):
        assert np.array_equal(a.get(n), b.get(n)), n


@pytest.mark.parametrize("config,hydraulics,dtype,Nz", CONFIGS)
def test_heun_single_launch_equals_reference_order_kernels_bitwise(config, hydraulics, dtype, Nz):
    lat, lon = small_columns(131)
    w = W.make_workload(config, lat, lon, Nz, dtype=dtype, hydraulics=hydraulics)
    if config == "heat":
        w["bcs"][("internal_energy", "bottom")] = ("flux", np.full(lat.size, 0.05))
    a, b = W.setup_device(w), W.setup_device(w)
    b.set_option("step_kernel", "unfused")
    for d in (a, b):
        d.step_heun(w["dt"], 12, finalize=False)
        d.step_heun(w["dt"], 1, finalize=True)
    for n in all_fields(w) + ["tend_internal_energy"]:
        assert np.array_equal(a.get(n), b.get(n), equal_nan=True), n


@pytest.mark.parametrize("m", [2, 7, 50])
@pytest.mark.parametrize("config,hydraulics,dtype,Nz", CONFIGS)
def test_multistep_program_equals_per_step_launches_bitwise(config, hydraulics, dtype, Nz, m):
    """TRM_OPT_STEPS_PER_LAUNCH = m: the column stays in registers for m steps (LandModel: with the surface energy
    balance evaluated in the kernel) -- same state, diagnostics, tendencies, clock and status as one launch per step."""
    lat, lon = small_columns(131)
    w = W.make_workload(config, lat, lon, Nz, dtype=dtype, hydraulics=hydraulics)
    a, b = W.setup_device(w), W.setup_device(w)
    a.set_option("steps_per_launch", m)
    b.set_option("steps_per_launch", 1)
    for d in (a, b):
        d.set_option("packed_f32", 0)
        d.step(w["dt"], 23, finalize=False)     # 23 = q * m + r: full launches and a shorter last one
        d.step(w["dt"], 10, finalize=True)
    assert a.clock() == b.clock()
    for n in all_fields(w) + ["tend_internal_energy"] + (["tend_saturation_water_ice", "tend_surface_excess_water"] if config != "heat" else []):
        assert np.array_equal(a.get(n), b.get(n), equal_nan=True), n
    assert a.status() == b.status()


@pytest.mark.parametrize("config,hydraulics,dtype,Nz", CONFIGS)
def test_default_steps_per_launch_is_the_resident_program_and_bit_identical(config, hydraulics, dtype, Nz):
    """TRM_OPT_STEPS_PER_LAUNCH = 0 (the library default): a plain trm_step(ctx, dt, n, fin) -- run!'s loop,
    model_integrator.jl:72-88 -- takes the resident-column program wherever it is legal (launches of up to 50 steps); same
    bits, clock, status and tendencies as one launch per step."""
    lat, lon = small_columns(97)
    w = W.make_workload(config, lat, lon, Nz, dtype=dtype, hydraulics=hydraulics)
    a, b = W.setup_device(w, steps_per_launch=0), W.setup_device(w, steps_per_launch=1)
    assert a.get_option("steps_per_launch") == 0 and trm.DeviceState(a.grid, a.params).get_option("steps_per_launch") == 0
    for d in (a, b):
        d.step(w["dt"], 57, finalize=False)     # 50 + 7
        d.step(w["dt"], 1, finalize=False)
        d.step(w["dt"], 12, finalize=True)
    assert a.clock() == b.clock()
    for n in all_fields(w) + ["tend_internal_energy"] + (["tend_saturation_water_ice", "tend_surface_excess_water"] if config != "heat" else []):
        assert np.array_equal(a.get(n), b.get(n), equal_nan=True), n
    assert a.status() == b.status()


PIPELINE_CONFIGS = [("land", "default", np.float64, 32, 131), ("land", "vg", np.float64, 50, 257), ("land", "default", np.float32, 64, 203),
                    ("land", "vg", np.float32, 40, 129), ("land", "default", np.float64, 20, 64), ("land", "vg", np.float64, 32, 65)]


@pytest.mark.parametrize("derive", [0, 1])
@pytest.mark.parametrize("config,hydraulics,dtype,Nz,Nh", PIPELINE_CONFIGS)
def test_land_halves_interleaved_equal_the_launch_pair_bitwise(config, hydraulics, dtype, Nz, Nh, derive):
    """TRM_OPT_PIPELINE_PARTS = 1: every launch steps the soil columns of one half of the context and evaluates the surface
    processes of the other half (k_land_euler / k_land_pk; the seam on a multiple of 64 columns, ragged second half) --
    same bits as k_surface + k_column per step: fields, diagnostics, tendencies, status, clock; with and without the
    derivation of T / liq, starting from a state the user has touched (top arrays invalid) and from a stepped one."""
    lat, lon = small_columns(Nh)
    w = W.make_workload(config, lat, lon, Nz, dtype=dtype, hydraulics=hydraulics)
    a, b = W.setup_device(w), W.setup_device(w)
    a.set_option("pipeline_parts", 1)
    b.set_option("pipeline_parts", 0)
    for d in (a, b):
        d.set_option("derive_closure_fields", derive)
        d.step(w["dt"], 9, finalize=False)       # from the initial state: the first surface evaluations read the fields
        d.step(w["dt"], 1, finalize=False)       # (a single step never interleaves)
        d.step(w["dt"], 2, finalize=False)       # from a stepped state: they read the top-cell arrays
        T = d.get("temperature")
        d.set("temperature", T)                  # an upload invalidates the top-cell arrays and the closure bookkeeping
        d.step(w["dt"], 3, finalize=True)
    assert a.clock() == b.clock()
    names = all_fields(w) + ["tend_internal_energy", "tend_saturation_water_ice", "tend_surface_excess_water"]
    for n in names:
        assert np.array_equal(a.get(n), b.get(n), equal_nan=True), n
    assert a.status() == b.status()


def test_land_interleaving_falls_back_where_it_does_not_apply():
    """Series-fed inputs, the coupled vegetation and a SoilModel keep the launch pair / the single launch; results equal."""
    lat, lon = small_columns(150)
    for config, hyd in (("land", "default"), ("landveg", "vg"), ("richards", "default")):
        w = W.make_workload(config, lat, lon, 32, hydraulics=hyd)
        a, b = W.setup_device(w), W.setup_device(w)
        a.set_option("pipeline_parts", 1)
        b.set_option("pipeline_parts", 0)
        tt = 600.0 * np.arange(4) * (w["dt"] / 60.0)
        ph = 2 * np.pi * tt[:, None] / 86400.0 - w["lon"][None, :]
        for d in (a, b):
            if config == "land":
                d.set_forcing_series("air_temperature", tt, w["T0"][None, :] + 5.0 * np.sin(ph), "linear")
            d.step(w["dt"], 7, finalize=True)
        for n in all_fields(w):
            assert np.array_equal(a.get(n), b.get(n), equal_nan=True), (config, n)


def test_multistep_with_repair_and_overflow_matches_oracle():
    """Columns that overflow into surface_excess_water and cells that the repair touches, 40 steps in launches of 8."""
    lat, lon = small_columns(64)
    w = W.make_workload("richards", lat, lon, 32)
    sat = w["fields"]["saturation_water_ice"].copy()
    sat[-3:, ::3] = 0.999                        # nearly saturated top cells over a wetting profile
    sat[5:9, 1::4] = 1.0
    w["fields"]["saturation_water_ice"] = sat
    w["bcs"][("saturation_water_ice", "top")] = ("flux", np.full(lat.size, -2.0e-6))   # infiltration into a full column
    d, o = W.setup_device(w), W.setup_oracle(w)
    d.set_option("steps_per_launch", 8)
    d.step(w["dt"], 40, finalize=True)
    o.run(w["dt"], 40)
    assert np.any(d.get("surface_excess_water") > 0)
    for n in all_fields(w):
        assert np.array_equal(d.get(n), o.get(n)), n


@pytest.mark.parametrize("m", [3, 16])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_multistep_with_time_series_interpolated_in_the_kernel(dtype, m):
    """Device-resident series (boundary values, forcing inputs; FieldTimeSeries and raster rules, all indexing modes) are
    interpolated inside the multi-step program, step by step: identical to update_inputs! + one launch per step, including
    what the input fields / boundary arrays hold afterwards."""
    lat, lon = small_columns(101)
    # heat: a top temperature series (cyclical) and a bottom heat-flux series (clamp)
    w = W.make_workload("heat", lat, lon, 20, dtype=dtype)
    a, b = W.setup_device(w), W.setup_device(w)
    a.set_option("steps_per_launch", m)
    b.set_option("steps_per_launch", 1)
    times = np.array([0.0, 2000.0, 5000.0, 9000.0])
    vals = np.stack([w["T0"] + x for x in (0.0, 4.0, -3.0, 1.0)])
    flux = np.stack([np.full(lat.size, x) for x in (0.0, 0.08, 0.02, 0.05)])
    for d in (a, b):
        d.set_bc_series("temperature", "top", "value", times, vals, "cyclical")
        d.set_bc_series("internal_energy", "bottom", "flux", times + 300.0, flux, "clamp")
        d.step(w["dt"], 37, finalize=True)
    for n in all_fields(w):
        assert np.array_equal(a.get(n), b.get(n)), n
    assert a.clock() == b.clock()
    # land: air temperature (linear), shortwave (raster rule), rainfall (clamp) series + a constant wind
    w = W.make_workload("land", lat, lon, 32, dtype=dtype, hydraulics="default")
    a, b = W.setup_device(w), W.setup_device(w)
    a.set_option("steps_per_launch", m)
    b.set_option("steps_per_launch", 1)
    tt = 600.0 * np.arange(6)
    ph = 2 * np.pi * tt[:, None] / 86400.0 - w["lon"][None, :]
    for d in (a, b):
        d.set_option("packed_f32", 0)
        d.set_forcing_series("air_temperature", tt, w["T0"][None, :] + 5.0 * np.sin(ph), "linear")
        d.set_forcing_series("surface_shortwave_down", tt + 90.0, np.maximum(0.0, 600.0 * np.sin(ph)), "raster")
        d.set_forcing_series("rainfall", tt, 1.0e-7 * (1 + np.cos(ph)), "clamp")
        d.step(w["dt"], 61, finalize=True)       # 61 x 60 s: beyond the last node
    for n in all_fields(w) + ["air_temperature", "surface_shortwave_down", "rainfall"]:
        assert np.array_equal(a.get(n), b.get(n), equal_nan=True), n
    assert a.status() == b.status()


def test_multistep_falls_back_for_generic_boundary_kinds():
    lat, lon = small_columns(70)
    w = W.make_workload("heat", lat, lon, 20)
    a, b = W.setup_device(w), W.setup_device(w)
    a.set_option("steps_per_launch", 10)
    b.set_option("steps_per_launch", 1)
    for d in (a, b):
        d.set_bc("temperature", "bottom", "gradient", 0.01)
        d.step(w["dt"], 25, finalize=True)
    assert np.array_equal(a.get("temperature"), b.get("temperature"))


def test_divide_sequences():
    """div_nr (trm_device.hpp): the step's variable-divisor divides without v_div_scale / v_div_fmas.  Through the C ABI:
    the energy closure T = U / C and the linear hydraulic conductivity K_sat * water / theta_sat over sweeps of operands
    (random mantissas, edge exponents of the legal ranges, zeros) must equal numpy's IEEE division bit for bit."""
    rng = np.random.default_rng(7)
    Nh, Nz = 4096, 32
    p = trm._capi.default_params()
    p.flow = 1
    d = trm.DeviceState(trm.ColumnGrid(trm.UniformSpacing(dz=0.1, N=Nz), Nh), p)
    por = 0.49
    sat = rng.uniform(1e-6, 1.0, (Nz, Nh))
    sat[0] = np.ldexp(rng.uniform(0.5, 1.0, Nh), rng.integers(-60, 0, Nh))      # tiny saturations
    U = np.concatenate([rng.uniform(0.0, 1e9, (Nz // 2, Nh)), np.ldexp(rng.uniform(0.5, 1, (Nz // 2, Nh)), rng.integers(-200, 30, (Nz // 2, Nh)))])
    U[3, :7] = 0.0
    frozen = rng.random((Nz, Nh)) < 0.4
    L = p.rho_w * p.Lsl
    Lth = L * sat * por
    U = np.where(frozen, -Lth - U, U)            # thawed (U >= 0) or fully frozen (U < -L_theta)
    d.set("internal_energy", U)
    d.set("saturation_water_ice", sat)
    d.closure()
    liq = np.where(U >= 0, 1.0, 0.0)
    wi = sat * por
    water, ice, air = wi * liq, wi * (1.0 - liq), (1.0 - sat) * por
    C = p.c_water * water
    C = C + p.c_ice * ice
    C = C + p.c_air * air
    C = C + p.c_mineral * ((1.0 - por) * 1.0)
    C = C + p.c_organic * ((1.0 - por) * 0.0)
    T_expected = np.where(U >= 0, U, U + Lth) / C
    assert np.array_equal(d.get("temperature"), T_expected)
    d.compute_auxiliary()
    K_expected = (p.K_sat * water) / (water + ice + air)
    Kc = d.get("hydraulic_conductivity")
    assert np.array_equal(Kc[0], K_expected[0]) and np.array_equal(Kc[-1], K_expected[-1])
    assert np.array_equal(Kc[1:-2], np.minimum(K_expected[1:-1], K_expected[:-2]))


def test_divide_sequences_fp32():
    """div_nr(float): through the packed fp32 step's closures (T = U / C, K = K_sat * water / theta_sat) on swept operands,
    against numpy's float32 division, bit for bit; packed == scalar kernels on the same inputs."""
    rng = np.random.default_rng(9)
    Nh, Nz = 2048, 64
    p = trm._capi.default_params()
    p.flow = 1
    d = trm.DeviceState(trm.ColumnGrid(trm.UniformSpacing(dz=0.1, N=Nz), Nh, dtype=np.float32), p)
    f = np.float32
    por = f(f(1) - f(0)) * f(0.49) + f(0) * f(0.9)
    sat = rng.uniform(1e-4, 1.0, (Nz, Nh)).astype(f)
    U = np.concatenate([rng.uniform(0.0, 1e9, (Nz // 2, Nh)), np.ldexp(rng.uniform(0.5, 1, (Nz // 2, Nh)), rng.integers(-60, 28, (Nz // 2, Nh)))]).astype(f)
    frozen = rng.random((Nz, Nh)) < 0.4
    L = f(p.rho_w) * f(p.Lsl)
    Lth = (L * sat) * por
    U = np.where(frozen, -Lth - U, U).astype(f)
    d.set("internal_energy", U)
    d.set("saturation_water_ice", sat)
    d.closure()
    liq = np.where(U >= 0, f(1), f(0)).astype(f)
    wi = sat * por
    water, ice, air = wi * liq, wi * (f(1) - liq), (f(1) - sat) * por
    solid = f(1) - por
    C = f(p.c_water) * water
    C = C + f(p.c_ice) * ice
    C = C + f(p.c_air) * air
    C = C + f(p.c_mineral) * (solid * f(1))
    C = C + f(p.c_organic) * (solid * f(0))
    T_expected = (np.where(U >= 0, U, U + Lth).astype(f) / C).astype(f)
    assert np.array_equal(d.get("temperature"), T_expected)
    d.compute_auxiliary()
    K_expected = ((f(p.K_sat) * water) / ((water + ice) + air)).astype(f)
    Kc = d.get("hydraulic_conductivity")
    assert np.array_equal(Kc[0], K_expected[0]) and np.array_equal(Kc[-1], K_expected[-1])


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_free_drainage_keeps_the_branch_free_programs(dtype):
    """FreeDrainage() (soil_model_bcs.jl:40: GradientBoundaryCondition(0) on the pressure head at the bottom) forms the halo an unset
    condition forms, bit for bit, so the context keeps the column programs -- derivation, signature instances, the resident multi-step
    program -- instead of the generic-boundary kernels: results against the reference-order kernels (which evaluate the Gradient
    formula) and the oracle; a non-zero gradient, a zero gradient at the TOP, a buffer that has been handed out take the generic path."""
    lat, lon = small_columns(260)
    w = W.make_workload("richards", lat, lon, 32, dtype=dtype)                  # (fp32: the packed two-columns-per-lane step)
    w["bcs"][("pressure_head", "bottom")] = ("gradient", 0.0)                      # FreeDrainage()
    w["bcs"][("temperature", "bottom")] = ("gradient", np.zeros(lat.size))          # (zero heat-flux gradient alike)
    fast, slow, ref, multi = W.setup_device(w), W.setup_device(w), W.setup_device(w), W.setup_device(w, steps_per_launch=0)
    o = W.setup_oracle(w) if dtype == np.float64 else None
    assert fast.get_option("info_generic_boundary_kernels") == 0 and fast.get_option("info_bc_signature") == 2
    slow.set_option("zero_gradient_fast", 0)
    assert slow.get_option("info_generic_boundary_kernels") == 1
    ref.set_option("step_kernel", "unfused")
    for d in (fast, slow, ref, multi):
        d.set_option("derive_closure_fields", 1)
        d.step(w["dt"], 1, finalize=False)
        d.step(w["dt"], 28, finalize=False)
        d.step(w["dt"], 1, finalize=True)
    for k in range(30 if o is not None else 0):
        o.timestep(w["dt"], True)
    for n in all_fields(w):
        a = ref.get(n)
        for d in (fast, slow, multi):
            assert np.array_equal(d.get(n), a, equal_nan=True), n
        assert o is None or np.array_equal(o.get(n), a), n
    # Heun alike
    for d in (fast, ref):
        d.step_heun(w["dt"], 7, finalize=True)
    for n in all_fields(w):
        assert np.array_equal(fast.get(n), ref.get(n), equal_nan=True), n
    # what ends it
    fast.set_bc("pressure_head", "bottom", "gradient", np.full(lat.size, 1.0e-3))
    assert fast.get_option("info_generic_boundary_kernels") == 1
    fast.set_bc("pressure_head", "bottom", "gradient", np.zeros(lat.size))
    assert fast.get_option("info_generic_boundary_kernels") == 0
    fast.bc_device_array("pressure_head", "bottom")                                  # handed out: the caller may write it
    assert fast.get_option("info_generic_boundary_kernels") == 1
    top = W.setup_device(W.make_workload("richards", lat, lon, 32, dtype=dtype))
    top.set_bc("pressure_head", "top", "gradient", np.zeros(lat.size))
    assert top.get_option("info_generic_boundary_kernels") == 1
    neg = W.setup_device(W.make_workload("richards", lat, lon, 32, dtype=dtype))
    neg.set_bc("pressure_head", "bottom", "gradient", np.full(lat.size, -0.0))          # -0.0 is not +0.0
    assert neg.get_option("info_generic_boundary_kernels") == 1
    # the host mirror's alias arrives as the scalar 0.0
    grid = trm.ColumnGrid(trm.ExponentialSpacing(N=20), 40)
    bcs = trm.merge_boundary_conditions(trm.PrescribedSurfaceTemperature("Ts", 1.0), trm.FreeDrainage())
    integ = trm.initialize(trm.SoilModel(grid, soil=trm.SoilEnergyWaterCarbon(hydrology=trm.SoilHydrology(vertical_flow=trm.RichardsEq()))), trm.ForwardEuler(dt=60.0),
                           boundary_conditions=bcs, initializers=dict(temperature=2.0, saturation_water_ice=0.7))
    assert integ.state.get_option("info_generic_boundary_kernels") == 0 and integ.state.get_option("info_bc_signature") == 2
    trm.run(integ, steps=5)


SIGNATURES = [("heat", {}, 2), ("heat", {("internal_energy", "bottom"): ("flux", 0.05)}, 6), ("richards", {}, 2),
              ("richards", {("internal_energy", "bottom"): ("flux", 0.05)}, 6), ("richards", "closed", 0), ("land", {}, 64),
              ("land", {("internal_energy", "bottom"): ("flux", 0.05)}, 68), ("richards", {("saturation_water_ice", "top"): ("flux", -2.0e-7)}, 34),
              ("richards", {("saturation_water_ice", "top"): ("flux", -2.0e-7), ("internal_energy", "bottom"): ("flux", 0.05)}, 38)]


@pytest.mark.parametrize("heun", [False, True])
@pytest.mark.parametrize("hydraulics,Nz", [("default", 32), ("vg", 50)])
@pytest.mark.parametrize("config,extra,signature", SIGNATURES)
def test_programs_with_the_boundary_signature_compiled_in_equal_the_runtime_program_bitwise(config, extra, signature, hydraulics, Nz, heun):
    """TRM_OPT_BC_SIGNATURE (BCSIG of k_column): the deriving ForwardEuler program with the boundary kinds as compile-time constants --
    no conditions, a prescribed surface temperature, that + a bottom heat flux, the LandModel wiring -- against the same program reading
    the kinds at run time and against the reference-order kernels; signatures without an instance (68, 38) take the run-time program."""
    lat, lon = small_columns(333)
    w = W.make_workload(config, lat, lon, Nz, hydraulics=hydraulics)
    if extra == "closed":
        w["bcs"].clear()
    else:
        for key, (kind, value) in extra.items():
            w["bcs"][key] = (kind, np.full(lat.size, value))
    sig, run, ref = W.setup_device(w), W.setup_device(w), W.setup_device(w)
    assert sig.get_option("bc_signature") == 1 and sig.get_option("info_bc_signature") == signature
    run.set_option("bc_signature", 0)
    ref.set_option("step_kernel", "unfused")
    for d in (sig, run, ref):
        d.set_option("derive_closure_fields", 1)
        step = d.step_heun if heun else d.step          # (Heun: the one-launch program, the stage's boundary values compiled in alike)
        step(w["dt"], 1, finalize=False)                # the first step reads T / liq as stored: the instances without the derivation
        step(w["dt"], 17, finalize=False)
        step(w["dt"], 1, finalize=True)
    for n in all_fields(w) + ["tend_internal_energy"]:
        a = ref.get(n)
        assert np.array_equal(sig.get(n), a, equal_nan=True), n
        assert np.array_equal(run.get(n), a, equal_nan=True), n
    assert sig.status() == run.status() == ref.status()


@pytest.mark.parametrize("hydraulics,Nz", [("default", 64), ("vg", 30)])
@pytest.mark.parametrize("config,signature", [("land", 64), ("richards", 2)])
def test_packed_fp32_step_with_the_boundary_signature_compiled_in(config, signature, hydraulics, Nz):
    """k_step_pk<..., DERIVE_LIQ, BCSIG>: the packed fp32 step of an HBM-resident state (the liquid fraction derived) with the LandModel
    wiring / the prescribed surface temperature compiled in, against the run-time kinds and the reference-order kernels."""
    lat, lon = small_columns(251)
    w = W.make_workload(config, lat, lon, Nz, dtype=np.float32, hydraulics=hydraulics)
    sig, run, ref = W.setup_device(w), W.setup_device(w), W.setup_device(w)
    assert sig.get_option("packed_f32") == 1 and sig.get_option("info_bc_signature") == signature
    run.set_option("bc_signature", 0)
    ref.set_option("step_kernel", "unfused")
    for d in (sig, run, ref):
        d.set_option("derive_closure_fields", 3)
        d.step(w["dt"], 1, finalize=False)
        d.step(w["dt"], 11, finalize=False)
        d.step(w["dt"], 1, finalize=True)
    for n in all_fields(w) + ["tend_internal_energy"]:
        a = ref.get(n)
        assert np.array_equal(sig.get(n), a, equal_nan=True), n
        assert np.array_equal(run.get(n), a, equal_nan=True), n
    assert sig.status() == run.status() == ref.status()


def test_staged_per_column_outputs_forced_on_small_grids():
    """The column programs store their per-column outputs either directly or through the workgroup's staging table
    (template parameter STAGED of the deriving instances; the library stages HBM-resident fp64 states and large LandModels).  TRM_STAGED_SMALL=1 forces the
    staged path for every context: the parity and program tests of this file and the LandModel parity cases must pass unchanged
    (the variable is read once per process, hence the child process)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # (and the per-column inputs by vector loads, ColumnArgs::scalar_in = 0: what HBM-resident states take -- the small grids of the
    # suite otherwise run with direct stores and the scalar memory path)
    # Both are compiled into the instances that derive T / liq (every large fp64 state): TRM_DERIVE_DEFAULT=1 makes the small
    # grids of the suite take those.
    env = dict(os.environ, TRM_STAGED_SMALL="1", TRM_SCALAR_INPUTS="0", TRM_DERIVE_DEFAULT="1")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_column_programs.py"), os.path.join(root, "tests", "test_gpu_parity.py"),
                          "-m", "gpu", "-q", "-x", "-W", "ignore::DeprecationWarning", "-k", "not staged_per_column_outputs and not hypothesis",
                          "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-1000:]
    assert " passed" in out.stdout and "failed" not in out.stdout, out.stdout[-500:]


def test_fast_path_bookkeeping_is_what_the_steps_leave_behind():
    """What the library tracks about its own buffers decides which kernel instance the NEXT step takes: the LandModel's surface
    evaluation reads the compact top-cell arrays only while they describe the state, T / liq are re-derived in registers only
    while the stored ones are the closure of the stored state.  A wrong flag costs speed, never correctness -- round 4 shipped a
    build for an hour in which the first one was stuck at 0 (C4 +7 %, C5 +17 %) with every parity test green -- so it is pinned."""
    lat, lon = small_columns(300)
    w = W.make_workload("land", lat, lon, 32)
    for steps_per_launch in (1, 0):
        d = W.setup_device(w, steps_per_launch=steps_per_launch)
        assert d.get_option("info_top_arrays_current") == 0 and d.get_option("info_closure_consistent") == 0     # the user's initial state
        d.step(w["dt"], 1, finalize=False)
        assert d.get_option("info_top_arrays_current") == 1 and d.get_option("info_closure_consistent") == 1
        d.step(w["dt"], 7, finalize=True)
        assert d.get_option("info_top_arrays_current") == 1 and d.get_option("info_closure_consistent") == 1
        d.step_heun(w["dt"], 2, finalize=False)
        assert d.get_option("info_top_arrays_current") == 1
        d.set("temperature", d.get("temperature"))                 # an upload: the library no longer knows
        assert d.get_option("info_top_arrays_current") == 0 and d.get_option("info_closure_consistent") == 0
        d.step(w["dt"], 1, finalize=False)
        assert d.get_option("info_top_arrays_current") == 1
        d.device_array("temperature")                              # a device pointer to T has escaped: never trust the copies again
        d.step(w["dt"], 1, finalize=False)
        assert d.get_option("info_top_arrays_current") == 0 and d.get_option("info_closure_consistent") == 0
    s = W.setup_device(W.make_workload("richards", lat, lon, 32))
    s.step(w["dt"], 2, finalize=False)
    assert s.get_option("info_top_arrays_current") == 0 and s.get_option("info_closure_consistent") == 1       # (a SoilModel has no top arrays)
