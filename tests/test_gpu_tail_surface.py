"""TRM_OPT_TAIL_SURFACE: the bare-ground LandModel stepped one launch per step evaluates the NEXT step's surface processes
(land_model.jl:79-88) at the tail of its column launch (k_column_tail: per 64 columns, by the last workgroup to finish them)
and the next step accepts them instead of launching k_surface.  Same operations per column as the k_surface + k_column pair, so
every field, diagnostic, tendency, the status word and the clock must agree BIT FOR BIT with the pair (option 0) -- and with
the CPU oracle to the LandModel tolerance -- whatever comes between two steps."""
import numpy as np
import pytest

import terrarium_jl_amd as trm
import workloads as W

pytestmark = pytest.mark.gpu


def small_columns(n):
    lat, lon = W.columns_from_mask("N72")
    sel = np.linspace(0, lat.size - 1, n).astype(int)
    return lat[sel], lon[sel]


def all_fields(w):
    return W.compared_fields(w)


TENDENCIES = ["tend_internal_energy", "tend_saturation_water_ice", "tend_surface_excess_water"]


def pair(w, derive=None):
    a, b = W.setup_device(w), W.setup_device(w)
    a.set_option("tail_surface", 1)
    b.set_option("tail_surface", 0)
    if derive is not None:
        for d in (a, b):
            d.set_option("derive_closure_fields", derive)
    return a, b


def assert_same(a, b, w, tendencies=True):
    assert a.clock() == b.clock()
    for n in all_fields(w) + (TENDENCIES if tendencies else []):
        assert np.array_equal(a.get(n), b.get(n), equal_nan=True), n
    assert a.status() == b.status()


# ragged sizes around the 64-column clusters and the 8-column workgroups; both level pitches; both compiled hydraulics
TAIL_CONFIGS = [("default", 32, 131), ("vg", 50, 257), ("default", 20, 64), ("vg", 32, 65), ("default", 64, 203), ("default", 32, 1), ("default", 32, 7),
                ("vg", 32, 513), ("default", 32, 4099)]


@pytest.mark.parametrize("derive", [0, 1])
@pytest.mark.parametrize("hydraulics,Nz,Nh", TAIL_CONFIGS)
def test_tail_surface_equals_the_launch_pair_bitwise(hydraulics, Nz, Nh, derive):
    lat, lon = small_columns(Nh)
    w = W.make_workload("land", lat, lon, Nz, hydraulics=hydraulics)
    a, b = pair(w, derive)
    for d in (a, b):
        d.step(w["dt"], 9, finalize=False)     # the first step launches k_surface, the eight that follow accept the pending set
    assert a.get_option("info_tail_pending") == 1 and b.get_option("info_tail_pending") == 0
    assert_same(a, b, w, tendencies=False)     # (downloads keep the pending set)
    assert a.get_option("info_tail_pending") == 1
    for d in (a, b):
        d.step(w["dt"], 1, finalize=False)
        d.step(w["dt"], 3, finalize=True)      # the finalizing launch's tail IS compute_auxiliary!(new state)
    assert a.get_option("info_tail_pending") == 0
    assert_same(a, b, w)
    for d in (a, b):
        d.step(w["dt"], 2, finalize=False)     # ... and the step after a finalize starts from those values
    assert_same(a, b, w, tendencies=False)


def test_tail_surface_matches_the_oracle():
    lat, lon = small_columns(300)
    for hyd in ("default", "vg"):
        w = W.make_workload("land", lat, lon, 32, hydraulics=hyd)
        d, o = W.setup_device(w), W.setup_oracle(w)
        d.set_option("tail_surface", 1)
        for _ in range(12):
            d.step(w["dt"], 1, finalize=False)
        d.step(w["dt"], 1, finalize=True)
        o.run(w["dt"], 13)
        for n in all_fields(w):
            x, y = d.get(n), o.get(n)
            assert np.max(np.abs(x - y) / np.maximum(1.0, np.abs(y))) < 1e-10, (hyd, n)
        assert d.status() == 0


def test_an_input_that_changes_between_two_steps_discards_the_pending_set():
    """Everything a caller can do between two steps that changes what the surface processes read: a forcing, the state, a boundary
    value, a restore.  The pending set was evaluated with the OLD values; the next step must not use it."""
    lat, lon = small_columns(333)
    w = W.make_workload("land", lat, lon, 32)
    a, b = pair(w)
    actions = [
        lambda d: d.set_forcing("air_temperature", w["inputs"]["air_temperature"] + 3.0),
        lambda d: d.set_forcing("surface_shortwave_down", w["inputs"]["surface_shortwave_down"] * 0.5 + 20.0),
        lambda d: d.set("skin_temperature", d.get("skin_temperature") + 1.0),
        lambda d: d.set("surface_excess_water", np.full(lat.size, 1.0e-3)),
        lambda d: d.set("temperature", d.get("temperature") - 0.5),
        lambda d: d.set_forcing("rainfall", np.full(lat.size, 2.0e-7)),
        lambda d: d.set_bc("internal_energy", "bottom", "flux", np.full(lat.size, 0.03)),
    ]
    for d in (a, b):
        d.step(w["dt"], 3, finalize=False)
    for act in actions:
        assert a.get_option("info_tail_pending") == 1
        for d in (a, b):
            act(d)
        assert a.get_option("info_tail_pending") == 0
        for d in (a, b):
            d.step(w["dt"], 2, finalize=False)
        assert_same(a, b, w, tendencies=False)
    # a bottom flux condition takes the context off the LandModel signature: the pair runs, nothing is left pending
    assert a.get_option("info_bc_signature") != 64 and a.get_option("info_tail_pending") == 0
    for d in (a, b):
        d.set_bc("internal_energy", "bottom", "noflux", 0.0)
        d.step(w["dt"], 2, finalize=False)
    assert a.get_option("info_tail_pending") == 1
    # save / restore: the restored state's surface processes are evaluated afresh
    for d in (a, b):
        d.save_state()
        d.step(w["dt"], 4, finalize=False)
        d.restore_state()
    assert a.get_option("info_tail_pending") == 0
    for d in (a, b):
        d.step(w["dt"], 3, finalize=True)
    assert_same(a, b, w)


def test_other_step_paths_between_tail_steps():
    """Heun steps, the resident multi-step program, the reference-order kernels and the stand-alone entry points between
    per-step launches: each leaves nothing pending, and the sequence equals the one without the tail evaluation."""
    lat, lon = small_columns(200)
    w = W.make_workload("land", lat, lon, 32, hydraulics="vg")
    a, b = pair(w)
    for d in (a, b):
        d.step(w["dt"], 2, finalize=False)
        d.step_heun(w["dt"], 1, finalize=False)
        d.step(w["dt"], 2, finalize=False)
        d.set_option("steps_per_launch", 4)
        d.step(w["dt"], 4, finalize=False)
        d.set_option("steps_per_launch", 1)
        d.step(w["dt"], 2, finalize=False)
        d.compute_auxiliary()                   # updates the skin temperature in place (surface_energy_balance.jl:107)
        d.step(w["dt"], 2, finalize=False)
        d.set_option("step_kernel", "unfused")
        d.step(w["dt"], 1, finalize=False)
        d.set_option("step_kernel", "fused")
        d.step(w["dt"], 2, finalize=True)
    assert_same(a, b, w)


def test_a_device_pointer_ends_the_tail_evaluation():
    """trm_field_device_ptr promises the pointer stays the field's and lets the caller write behind the library's back: no more
    swaps, nothing evaluated ahead."""
    lat, lon = small_columns(100)
    w = W.make_workload("land", lat, lon, 32)
    a, b = pair(w)
    for d in (a, b):
        d.step(w["dt"], 2, finalize=False)
    assert a.get_option("info_tail_pending") == 1
    ptr = lambda d: d.device_array("ground_heat_flux").__cuda_array_interface__["data"][0]
    p0 = ptr(a)
    assert a.get_option("info_tail_pending") == 0
    for d in (a, b):
        d.step(w["dt"], 3, finalize=False)
    assert a.get_option("info_tail_pending") == 0
    assert ptr(a) == p0
    assert_same(a, b, w, tendencies=False)


def test_where_the_tail_evaluation_does_not_apply():
    """A series-fed input, the coupled vegetation, fp32, a SoilModel: the launch pair / the single launch as before."""
    lat, lon = small_columns(150)
    for config, hyd, dtype in (("land", "default", np.float64), ("landveg", "vg", np.float64), ("richards", "default", np.float64), ("land", "default", np.float32)):
        w = W.make_workload(config, lat, lon, 32, hydraulics=hyd, dtype=dtype)
        a, b = pair(w)
        tt = 600.0 * np.arange(4) * (w["dt"] / 60.0)
        ph = 2 * np.pi * tt[:, None] / 86400.0 - w["lon"][None, :]
        for d in (a, b):
            if config == "land" and dtype == np.float64:
                d.set_forcing_series("air_temperature", tt, w["T0"][None, :] + 5.0 * np.sin(ph), "linear")
            d.step(w["dt"], 5, finalize=False)
        assert a.get_option("info_tail_pending") == 0
        for d in (a, b):
            d.step(w["dt"], 2, finalize=True)
        for n in all_fields(w):
            assert np.array_equal(a.get(n), b.get(n), equal_nan=True), (config, n)


def test_checkpoint_restart_with_the_tail_evaluation():
    """trm.checkpoint / trm.restore of a LandModel that steps with the tail evaluation: the restarted run equals the uninterrupted one."""
    lat, lon = small_columns(120)
    w = W.make_workload("land", lat, lon, 32)
    a, b = pair(w)
    a.step(w["dt"], 10, finalize=False)
    b.step(w["dt"], 10, finalize=False)
    snap = {n: a.get(n) for n in all_fields(w)}
    c = W.setup_device(w)
    c.set_option("tail_surface", 1)
    for n, v in snap.items():
        if n not in ("hydraulic_conductivity",):
            c.set(n, v)
    c.set_clock(*a.clock())
    for d in (a, b, c):
        d.step(w["dt"], 5, finalize=True)
    assert_same(a, b, w)
    for n in all_fields(w):
        assert np.array_equal(a.get(n), c.get(n), equal_nan=True), n
