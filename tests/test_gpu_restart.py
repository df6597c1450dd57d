"""Restart through the C ABI ("download/upload of all prognostic buffers + clock is sufficient for restart";
docs/src/running/time_stepping.md:97-139): `trm.checkpoint(integ)` downloads the restart set (trm_download) with the clock,
`trm.restore(fresh, ckpt)` uploads it into a FRESH context (trm_upload, trm_set_clock) -- 20 steps must equal 12 steps +
restart + 8 steps bit for bit, ForwardEuler and Heun, on the vegetation-coupled LandModel (skin temperature, canopy water,
carbon pools, the carried net assimilation) and on Richards with a time series streamed through a device window."""
import pickle

import numpy as np
import pytest

import terrarium_jl_amd as trm
import workloads as W
from test_gpu_coupled_vegetation import land_with_vegetation, PROG, SURFACE, VEG_AUX, CANOPY_AUX

pytestmark = pytest.mark.gpu


def coupled_integrator(n, stepper, dt=0.5, N=20, seed=5):
    rng = np.random.default_rng(seed)
    grid = trm.ColumnGrid(trm.ExponentialSpacing(dz_max=1.0, N=N), n)
    zc = grid.z_centers()
    T0 = (5.0 - 0.02 * zc)[:, None] + rng.uniform(-1, 1, n)[None, :]
    sat0 = np.clip(np.minimum(1.0, 0.8 - 0.05 * zc)[:, None] * (1 + 0.05 * rng.uniform(-1, 1, n))[None, :], 0.05, 1.0)
    inits = dict(temperature=T0, saturation_water_ice=sat0, carbon_vegetation=rng.uniform(1.0, 2.0, n),
                 vegetation_area_fraction=rng.uniform(0.05, 0.9, n), canopy_water=rng.uniform(0.0, 1.0e-4, n))
    inputs = dict(SAI=rng.uniform(0.0, 1.0, n), rainfall=rng.uniform(0.0, 2.0e-7, n), air_temperature=rng.uniform(2.0, 20.0, n),
                  specific_humidity=rng.uniform(1.0e-3, 5.0e-3, n), windspeed=rng.uniform(0.0, 4.0, n), CO2=rng.uniform(300.0, 500.0, n),
                  daily_leaf_respiration=rng.uniform(0.0, 1e-6, n))
    return trm.initialize(land_with_vegetation(grid), stepper(dt=dt), initializers=inits, inputs=inputs)


@pytest.mark.parametrize("finalized", [True, False])
@pytest.mark.parametrize("stepper", [trm.ForwardEuler, trm.Heun])
def test_restart_of_the_vegetation_coupled_land_model_into_a_fresh_context(stepper, finalized):
    """`finalized`: the first leg ends as run! does, with compute_auxiliary! (model_integrator.jl:85-86) -- one more evaluation of
    the surface energy balance and of the net assimilation the next step's stomatal conductance reads, in the reference as
    here -- so the uninterrupted run is run(12); run(8) in ONE context.  Not finalized (the driver's inner loop): 12 + 8 steps
    across the restart equal 20 steps in one call."""
    n = 150
    whole = coupled_integrator(n, stepper)
    first = coupled_integrator(n, stepper)
    if finalized:
        trm.run(whole, steps=12)
        trm.run(whole, steps=8)
        trm.run(first, steps=12)
    else:
        trm.run(whole, steps=20)
        first._step(first.timestepper.dt, 12, finalize=False)
    ckpt = pickle.loads(pickle.dumps(trm.checkpoint(first)))          # (what a restart file holds: plain arrays and numbers)
    assert set(ckpt["fields"]) == {"internal_energy", "temperature", "liquid_water_fraction", "saturation_water_ice", "pressure_head",
                                   "surface_excess_water", "water_table", "skin_temperature", "carbon_vegetation", "vegetation_area_fraction",
                                   "net_assimilation", "canopy_water"}
    first.state.close()
    fresh = coupled_integrator(n, stepper, seed=5)                     # a cold start of the same set-up; its state is replaced
    trm.restore(fresh, ckpt)
    assert fresh.state.clock() == (ckpt["time"], 12)
    trm.run(fresh, steps=8)
    assert fresh.state.clock() == whole.state.clock()
    for name in PROG + SURFACE + VEG_AUX + CANOPY_AUX + ("liquid_water_fraction", "pressure_head", "water_table", "hydraulic_conductivity"):
        assert np.array_equal(fresh.state.get(name), whole.state.get(name), equal_nan=True), name
    assert fresh.state.status() == whole.state.status() == 0


@pytest.mark.parametrize("stepper", [trm.ForwardEuler, trm.Heun])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_restart_with_a_windowed_series(stepper, dtype):
    """Richards with the top temperature from a 120-level record streamed through a 10-level window: the fresh integrator's
    window starts at the head of the record and is moved forward to the restored clock (trim + append)."""
    lat, lon = W.synthetic_columns(257)
    w = W.make_workload("richards", lat, lon, 32, dtype=dtype)
    nt = 120
    t = 150.0 * np.arange(nt) + 40.0 * np.sin(np.arange(nt))
    vals = w["T0"][None, :] + 10.0 * np.sin(2 * np.pi * t[:, None] / 86400.0 - w["lon"][None, :])

    def make():
        grid = trm.ColumnGrid(trm.PrescribedSpacing(dz=list(w["thickness"])), w["Nh"], dtype=dtype)
        model = trm.SoilModel(grid, soil=trm.SoilEnergyWaterCarbon(hydrology=trm.SoilHydrology(vertical_flow=trm.RichardsEq())))
        bc = trm.PrescribedSurfaceTemperature("Ts", trm.FieldTimeSeries(t, vals).windowed(10))
        return trm.initialize(model, stepper(dt=w["dt"]), boundary_conditions=bc,
                              initializers=dict(temperature=w["fields"]["temperature"], saturation_water_ice=w["fields"]["saturation_water_ice"]))

    whole = make()
    trm.run(whole, steps=200)
    first = make()
    trm.run(first, steps=120)
    ckpt = trm.checkpoint(first)
    first.state.close()
    fresh = make()
    trm.restore(fresh, ckpt)
    info = fresh.state.series_info(("temperature", "top"))
    assert info["t_first"] <= ckpt["time"] < info["t_last"] and info["capacity"] == 10
    trm.run(fresh, steps=80)
    assert fresh.state.clock() == whole.state.clock() == (200 * w["dt"], 200)
    for name in W.compared_fields(w):
        assert np.array_equal(fresh.state.get(name), whole.state.get(name), equal_nan=True), name


def test_restore_back_in_time_in_a_used_integrator_and_the_status_word():
    """A restore that goes BACK in time in the integrator that wrote the checkpoint: its window has been trimmed past the
    checkpoint's clock and is started over from the head of the record; the status word the checkpoint carried is put back (a NaN
    raised before the checkpoint is still reported after the restart, a clean checkpoint clears a later flag)."""
    lat, lon = W.synthetic_columns(130)
    w = W.make_workload("richards", lat, lon, 32)
    nt = 120
    t = 150.0 * np.arange(nt) + 40.0 * np.sin(np.arange(nt))
    vals = w["T0"][None, :] + 10.0 * np.sin(2 * np.pi * t[:, None] / 86400.0 - w["lon"][None, :])

    def make():
        grid = trm.ColumnGrid(trm.PrescribedSpacing(dz=list(w["thickness"])), w["Nh"])
        model = trm.SoilModel(grid, soil=trm.SoilEnergyWaterCarbon(hydrology=trm.SoilHydrology(vertical_flow=trm.RichardsEq())))
        bc = trm.PrescribedSurfaceTemperature("Ts", trm.FieldTimeSeries(t, vals).windowed(10))
        return trm.initialize(model, trm.ForwardEuler(dt=w["dt"]), boundary_conditions=bc,
                              initializers=dict(temperature=w["fields"]["temperature"], saturation_water_ice=w["fields"]["saturation_water_ice"]))

    whole = make()
    trm.run(whole, steps=160)
    used = make()
    trm.run(used, steps=40)
    ckpt = trm.checkpoint(used)                       # clean, at step 40
    trm.run(used, steps=100)                          # the window moves on: its head is far beyond the checkpoint's clock now
    assert used.state.series_info(("temperature", "top"))["t_first"] > ckpt["time"]
    T = used.state.get("temperature"); T[0, 0] = np.nan
    used.state.set("temperature", T)
    trm.run(used, steps=1)
    assert used.state.status() & 1
    trm.restore(used, ckpt)
    assert used.state.status() == 0 and used.state.clock() == (40 * w["dt"], 40)
    info = used.state.series_info(("temperature", "top"))
    assert info["t_first"] <= ckpt["time"] < info["t_last"]
    trm.run(used, steps=120)
    for name in W.compared_fields(w):
        assert np.array_equal(used.state.get(name), whole.state.get(name), equal_nan=True), name
    flagged = dict(ckpt, status=1)
    trm.restore(used, flagged)
    assert used.state.status() == 1


def test_restore_refuses_a_checkpoint_of_another_grid():
    a, b = coupled_integrator(20, trm.ForwardEuler), coupled_integrator(21, trm.ForwardEuler)
    with pytest.raises(ValueError, match="another grid"):
        trm.restore(b, trm.checkpoint(a))
    ck = trm.checkpoint(a)
    del ck["fields"]["net_assimilation"]
    with pytest.raises(ValueError, match="lacks"):
        trm.restore(coupled_integrator(20, trm.ForwardEuler), ck)
