"""Property-based parity (hypothesis): random grids, column counts, initial states, boundary kinds and parameters -- the fused
kernels, the reference-order kernels and the oracle must agree bit for bit on the pure-arithmetic configurations (fp64 heat,
heat + Richards with the reference-default hydraulics), to 1e-10 on van Genuchten / LandModel.  Seeds are derandomised so that
the round-end run sees the same cases."""
import numpy as np
import pytest
hypothesis = pytest.importorskip("hypothesis")
from hypothesis import event, given, settings, strategies as st, HealthCheck

import oracle
import terrarium_jl_amd as trm

pytestmark = pytest.mark.gpu

BC_CHOICES = [
    {},
    {("temperature", "top"): "value"},
    {("temperature", "top"): "value", ("temperature", "bottom"): "value"},
    {("internal_energy", "bottom"): "flux", ("temperature", "top"): "value"},
    {("saturation_water_ice", "top"): "flux", ("internal_energy", "top"): "flux"},
    {("temperature", "top"): "gradient"},                                   # generic kinds: k_step_wave
    {("pressure_head", "bottom"): "gradient", ("temperature", "top"): "value"},
    {("liquid_water_fraction", "top"): "gradient"},
]


@st.composite
def cases(draw):
    Nz = draw(st.integers(2, 64))
    Nh = draw(st.integers(1, 150))
    seed = draw(st.integers(0, 2**31 - 1))
    config = draw(st.sampled_from(["heat", "richards", "richards", "land"]))
    hyd = draw(st.sampled_from(["default", "default", "vg"]))
    bcs = draw(st.sampled_from(BC_CHOICES))
    heun = draw(st.booleans())
    spl = draw(st.sampled_from([1, 1, 3]))
    return Nz, Nh, seed, config, hyd, bcs, heun, spl


@settings(max_examples=120, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(cases())
def test_random_configurations_agree(case):
    Nz, Nh, seed, config, hyd, bcs, heun, spl = case
    rng = np.random.default_rng(seed)
    thickness = np.sort(rng.uniform(0.05, 0.15, Nz) * np.exp(np.linspace(0.0, rng.uniform(0.0, 4.0), Nz)))   # thin at the surface
    params = dict(flow=0 if config == "heat" else 1, seb=1 if config == "land" else 0)
    if hyd == "vg" and config != "heat":
        params.update(swrc=1, unsat_k=1, vg_alpha=2.0, vg_n=2.0)
    if config == "land":
        bcs = {k: v for k, v in bcs.items() if k[1] != "top" or k[0] not in ("internal_energy", "saturation_water_ice")}   # wired by the LandModel
    grid = trm.ColumnGrid(trm.PrescribedSpacing(dz=list(thickness)), Nh)
    zc = grid.z_centers()
    T0 = rng.uniform(-6.0, 12.0, Nh)
    T = T0[None, :] - rng.uniform(0.0, 0.08) * zc[:, None] + rng.normal(0.0, 0.3, (Nz, Nh))
    sat = np.clip(rng.uniform(0.3, 0.95) - rng.uniform(0.0, 0.06) * zc[:, None] + rng.normal(0.0, 0.03, (Nz, Nh)), 0.08, 1.0) if config != "heat" else np.ones((Nz, Nh))
    dt = 1.0 if config != "heat" else 100.0      # (explicit schemes on layers of a few cm)
    values = {"value": lambda k: (T0 + rng.uniform(-3, 3, Nh)) if k[0] == "temperature" else rng.uniform(0.1, 0.9, Nh),
              "flux": lambda k: rng.uniform(-1.0, 1.0, Nh) * (0.05 if k[0] == "internal_energy" else 2.0e-8),
              "gradient": lambda k: rng.uniform(-0.05, 0.05, Nh)}
    bc_values = {k: (kind, values[kind](k)) for k, kind in bcs.items()}
    inputs = dict(air_temperature=T0 + rng.uniform(-4, 4, Nh), windspeed=rng.uniform(0.0, 5.0, Nh), rainfall=rng.uniform(0.0, 3.0e-8, Nh),
                  surface_shortwave_down=rng.uniform(0.0, 500.0, Nh), specific_humidity=rng.uniform(5e-4, 6e-3, Nh)) if config == "land" else {}

    p = oracle.default_params(**params)
    o = oracle.Oracle(Nh, thickness, p)
    devs = []
    for kernel in ("fused", "unfused"):
        tp = trm._capi.default_params()
        for k, v in params.items():
            setattr(tp, k, v)
        d = trm.DeviceState(grid, tp)
        d.set_option("step_kernel", kernel)
        if kernel == "fused":
            d.set_option("steps_per_launch", spl)
        devs.append(d)
    for target in [o] + devs:
        target.set("temperature", T)
        target.set("saturation_water_ice", sat)
        if config == "land":
            target.set("skin_temperature", T[-1])
        for (var, side), (kind, val) in bc_values.items():
            target.set_bc(var, side, kind, val)
        for name, val in inputs.items():
            (target.set if target is o else target.set_forcing)(name, val)
        target.initialize()
    nsteps = 6
    for n in range(nsteps):
        fin = n == nsteps - 1
        (o.timestep_heun if heun else o.timestep)(dt, fin)
    for d in devs:
        (d.step_heun if heun else d.step)(dt, nsteps, True)
    exact = config in ("heat", "richards") and hyd == "default"
    names = ["temperature", "internal_energy", "liquid_water_fraction"] + (["saturation_water_ice", "pressure_head", "surface_excess_water", "water_table"] if config != "heat" else []) \
        + (["skin_temperature", "ground_heat_flux", "latent_heat_flux", "infiltration"] if config == "land" else [])
    assert devs[0].status() == devs[1].status(), case
    assert (devs[0].status() & 2) == (o.status() & 2), case          # (the oracle does not scan for NaN: bit 0 is the device's)
    if o.status() != 0 or not all(np.all(np.isfinite(o.get(n))) for n in ("temperature", "saturation_water_ice", "pressure_head")):
        # an unstable draw (a dried-out top cell: psi = -Inf): the device must have flagged it; values of an invalid state
        # are outside the parity contract (min / max drop NaN operands)
        assert devs[0].status() != 0, case
        event("unstable draw")
        return
    event(f"compared: {config}/{hyd}{' heun' if heun else ''}{' multi' if spl > 1 else ''}")
    for name in names:
        a, b, c = devs[0].get(name), devs[1].get(name), o.get(name)
        assert np.array_equal(a, b, equal_nan=True), (name, "fused != unfused", case)
        if exact:
            assert np.array_equal(a, c, equal_nan=True), (name, case)
        else:
            ok = np.isfinite(c)
            assert np.array_equal(np.isfinite(a), ok), (name, case)
            assert np.max(np.abs(a[ok] - c[ok]) / np.maximum(1.0, np.abs(c[ok])), initial=0.0) <= 1e-10, (name, case)
