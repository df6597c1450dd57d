"""TRM_OPT_SURFACE_IN_LAUNCH: the bare-ground LandModel stepped one launch per step runs its 0-D surface processes
(land_model.jl:79-88) in the first workgroups of the step launch (k_column_land) and hands ground heat flux, infiltration and the
skin temperature to the column workgroups of the same launch.  Same operations per column as the k_surface + k_column pair, so
every field, diagnostic, tendency, the status word and the clock must agree BIT FOR BIT with the pair (option 0) -- and with the
CPU oracle to the LandModel tolerance -- whatever comes between two steps."""
import numpy as np
import pytest

import terrarium_jl_amd as trm
import workloads as W

pytestmark = pytest.mark.gpu

PROGRAM_LAND = 7       # TRM_PROGRAM_COLUMN_LAND: family id in the low byte of TRM_INFO_LAST_PROGRAM
PROGRAM_PACKED_LAND = 13


def small_columns(n):
    lat, lon = W.columns_from_mask("N72")
    sel = np.linspace(0, lat.size - 1, n).astype(int)
    return lat[sel], lon[sel]


def all_fields(w):
    return W.compared_fields(w)


TENDENCIES = ["tend_internal_energy", "tend_saturation_water_ice", "tend_surface_excess_water"]


def pair(w, derive=None):
    a, b = W.setup_device(w), W.setup_device(w)
    a.set_option("surface_in_launch", 1)
    b.set_option("surface_in_launch", 0)
    if derive is not None:
        for d in (a, b):
            d.set_option("derive_closure_fields", derive)
    return a, b


def family(d):
    return d.get_option("info_last_program") & 0xff


def assert_same(a, b, w, tendencies=True):
    assert a.clock() == b.clock()
    for n in all_fields(w) + (TENDENCIES if tendencies else []):
        assert np.array_equal(a.get(n), b.get(n), equal_nan=True), n
    assert a.status() == b.status()


# ragged sizes around the 256-column surface workgroups and the 8-column workgroups; both level pitches; both compiled hydraulics
CONFIGS = [("default", 32, 131), ("vg", 50, 257), ("default", 20, 64), ("vg", 32, 65), ("default", 64, 203), ("default", 32, 1), ("default", 32, 7),
           ("vg", 32, 513), ("default", 32, 4099)]


@pytest.mark.parametrize("derive", [0, 1])
@pytest.mark.parametrize("hydraulics,Nz,Nh", CONFIGS)
def test_surface_in_launch_equals_the_launch_pair_bitwise(hydraulics, Nz, Nh, derive):
    lat, lon = small_columns(Nh)
    w = W.make_workload("land", lat, lon, Nz, hydraulics=hydraulics)
    a, b = pair(w, derive)
    for d in (a, b):
        d.step(w["dt"], 9, finalize=False)     # the first step reads the fields (k_surface), the eight that follow the top-cell arrays
    assert family(a) == PROGRAM_LAND and family(b) != PROGRAM_LAND
    assert_same(a, b, w, tendencies=False)
    for d in (a, b):
        d.step(w["dt"], 1, finalize=False)
        d.step(w["dt"], 3, finalize=True)
    assert_same(a, b, w)
    for d in (a, b):
        d.step(w["dt"], 2, finalize=False)
    assert family(a) == PROGRAM_LAND
    assert_same(a, b, w, tendencies=False)


@pytest.mark.parametrize("derive", [0, 2])
@pytest.mark.parametrize("hydraulics,Nz,Nh", [("default", 64, 131), ("vg", 50, 257), ("default", 20, 64), ("vg", 32, 65), ("default", 32, 1), ("default", 64, 7),
                                              ("default", 32, 2), ("vg", 64, 4099), ("default", 32, 4098)])
def test_packed_fp32_surface_in_launch_equals_the_launch_pair_bitwise(hydraulics, Nz, Nh, derive):
    """k_step_pk_land: the packed fp32 step (two columns per lane, two or four columns per wave) with the surface processes in the
    first workgroups of the launch; one granule per value."""
    lat, lon = small_columns(Nh)
    w = W.make_workload("land", lat, lon, Nz, hydraulics=hydraulics, dtype=np.float32)
    a, b = pair(w, derive)      # (2: the library's rule -- at these sizes T / liq are read; 0: never derived)
    if derive == 2:
        for d in (a, b):
            d.set_option("derive_closure_fields", 3)      # the liquid fraction derived, as on the HBM-resident C5 shard
    for d in (a, b):
        d.step(w["dt"], 9, finalize=False)
    assert family(a) == PROGRAM_PACKED_LAND and family(b) != PROGRAM_PACKED_LAND
    assert_same(a, b, w, tendencies=False)
    for d in (a, b):
        d.step(w["dt"], 1, finalize=False)
        d.step(w["dt"], 3, finalize=True)
    assert_same(a, b, w)
    for d in (a, b):
        d.set_forcing("air_temperature", w["inputs"]["air_temperature"].astype(np.float32) + 2.0)
        d.step(w["dt"], 2, finalize=False)
    assert family(a) == PROGRAM_PACKED_LAND
    assert_same(a, b, w, tendencies=False)


@pytest.mark.parametrize("hydraulics,Nz,Nh", [("default", 32, 131), ("vg", 50, 257), ("default", 20, 64), ("default", 64, 7), ("vg", 32, 4099)])
def test_heun_with_the_surface_processes_in_the_launch_equals_the_launch_pair_bitwise(hydraulics, Nz, Nh):
    """k_column_land<..., PROG_HEUN>: the one-launch Heun step (heun.jl:37-71; both stages in registers) with the STATE's surface
    processes in the first workgroups of the launch -- the stage's are not evaluated (its fluxes would only enter through
    compute_z_bcs!, which the reference runs for the state alone)."""
    lat, lon = small_columns(Nh)
    w = W.make_workload("land", lat, lon, Nz, hydraulics=hydraulics)
    a, b = pair(w)
    for d in (a, b):
        d.step_heun(w["dt"], 7, finalize=False)
    assert family(a) == PROGRAM_LAND and a.last_program()["program"] == "heun" and family(b) != PROGRAM_LAND
    assert_same(a, b, w, tendencies=False)
    for d in (a, b):
        d.set_forcing("air_temperature", w["inputs"]["air_temperature"] - 1.5)
        d.step_heun(w["dt"], 1, finalize=False)
        d.step(w["dt"], 2, finalize=False)
        d.step_heun(w["dt"], 3, finalize=True)
    assert_same(a, b, w)


def test_surface_in_launch_matches_the_oracle():
    lat, lon = small_columns(300)
    for hyd in ("default", "vg"):
        w = W.make_workload("land", lat, lon, 32, hydraulics=hyd)
        d, o = W.setup_device(w), W.setup_oracle(w)
        d.set_option("surface_in_launch", 1)
        for _ in range(12):
            d.step(w["dt"], 1, finalize=False)
        assert family(d) == PROGRAM_LAND
        d.step(w["dt"], 1, finalize=True)
        o.run(w["dt"], 13)
        for n in all_fields(w):
            x, y = d.get(n), o.get(n)
            assert np.max(np.abs(x - y) / np.maximum(1.0, np.abs(y))) < 1e-10, (hyd, n)
        assert d.status() == 0


def test_inputs_that_change_between_two_steps():
    """Nothing is evaluated ahead: whatever a caller changes between two steps -- a forcing, the state, a boundary value, a restore,
    a time series -- the next launch's surface workgroups read."""
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
        for d in (a, b):
            act(d)
            d.step(w["dt"], 2, finalize=False)
        assert_same(a, b, w, tendencies=False)
    # a bottom flux condition takes the context off the LandModel signature: the pair runs
    assert a.get_option("info_bc_signature") != 64 and family(a) != PROGRAM_LAND
    for d in (a, b):
        d.set_bc("internal_energy", "bottom", "noflux", 0.0)
        d.step(w["dt"], 2, finalize=False)
    assert family(a) == PROGRAM_LAND
    for d in (a, b):
        d.save_state()
        d.step(w["dt"], 4, finalize=False)
        d.restore_state()
        d.step(w["dt"], 3, finalize=True)
    assert_same(a, b, w)
    # a forcing series: update_inputs! runs in front of the launch as before
    tt = 600.0 * np.arange(6)
    ph = 2 * np.pi * tt[:, None] / 86400.0 - w["lon"][None, :]
    for d in (a, b):
        d.set_forcing_series("air_temperature", tt + d.clock()[0], w["T0"][None, :] + 5.0 * np.sin(ph), "linear")
        d.step(w["dt"], 1, finalize=False)
        d.step(w["dt"], 1, finalize=False)
        d.step(w["dt"], 1, finalize=False)
    assert family(a) == PROGRAM_LAND
    assert_same(a, b, w, tendencies=False)


def test_other_step_paths_between_in_launch_steps():
    """Heun steps, the resident multi-step program, the reference-order kernels and the stand-alone entry points between
    per-step launches; the sequence equals the one with the launch pair."""
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


def test_a_device_pointer_into_the_state_keeps_the_launch_pair():
    """With a pointer to temperature / saturation / liquid fraction handed out the top-cell arrays are never trusted again: the
    surface workgroups, which read them, do not run."""
    lat, lon = small_columns(100)
    w = W.make_workload("land", lat, lon, 32)
    a, b = pair(w)
    for d in (a, b):
        d.step(w["dt"], 2, finalize=False)
    assert family(a) == PROGRAM_LAND
    a.device_array("temperature")
    for d in (a, b):
        d.step(w["dt"], 3, finalize=False)
    assert family(a) != PROGRAM_LAND
    assert_same(a, b, w, tendencies=False)


def test_where_the_in_launch_surface_does_not_apply():
    """The coupled vegetation, a SoilModel, fp32 off the packed kernel: the launch pair / the single launch as before."""
    lat, lon = small_columns(150)
    for config, hyd, dtype in (("landveg", "vg", np.float64), ("richards", "default", np.float64), ("land", "default", np.float32)):
        w = W.make_workload(config, lat, lon, 32, hydraulics=hyd, dtype=dtype)
        a, b = pair(w)
        for d in (a, b):
            if dtype == np.float32:
                d.set_option("packed_f32", 0)
            d.step(w["dt"], 5, finalize=False)
        assert family(a) not in (PROGRAM_LAND, PROGRAM_PACKED_LAND)
        for d in (a, b):
            d.step(w["dt"], 2, finalize=True)
        for n in all_fields(w):
            assert np.array_equal(a.get(n), b.get(n), equal_nan=True), (config, n)


def test_full_size_under_load_every_word():
    """The N145 grid (56 951 columns, 7 119 column workgroups behind 223 surface workgroups: three and a half generations of
    resident workgroups, the first of which reaches the explicit step before the surface workgroups have published): 40 steps,
    every word of every field against the launch pair."""
    lat, lon = W.columns_from_mask("N145")
    for hyd in ("default", "vg"):
        w = W.make_workload("land", lat, lon, 32, hydraulics=hyd)
        a, b = pair(w)
        for d in (a, b):
            d.step(w["dt"], 40, finalize=False)
            d.step(w["dt"], 1, finalize=True)
        assert family(b) != PROGRAM_LAND
        assert_same(a, b, w)
        assert a.status() & 4 == 0


def test_fp32_shard_under_load_every_word():
    """The C5 shape at a tenth of its size (81 250 columns x 64 levels, fp32: 318 surface workgroups in front of 10 157 column
    workgroups): 30 steps, every word against the launch pair."""
    lat, lon = W.synthetic_columns(81250)
    w = W.make_workload("land", lat, lon, 64, dtype=np.float32)
    a, b = pair(w)
    for d in (a, b):
        d.step(w["dt"], 30, finalize=False)
        d.step(w["dt"], 1, finalize=True)
    assert_same(a, b, w)
    assert a.status() & 4 == 0


def test_hand_off_under_uneven_load_every_word():
    """The hand-off of the in-launch surface processes must not depend on what else the device is doing: an N145 LandModel stepped
    one launch per step on its own stream WHILE a second context streams a 0.93 GB heat + Richards state through the same device on
    another (the column waves of the LandModel then share CUs, L2 slices and the fabric with a foreign kernel, and the generations of
    its workgroups interleave with the other launch's) -- every word of every field against the launch pair stepped alone; no wave
    ever gives up waiting (status bit 4)."""
    lat, lon = W.columns_from_mask("N145")
    w = W.make_workload("land", lat, lon, 32)
    big = W.make_workload("richards", np.tile(lat, 8), np.tile(lon, 8), 32)
    a, b, hog = W.setup_device(w), W.setup_device(w), W.setup_device(big)
    a.set_option("surface_in_launch", 1)
    b.set_option("surface_in_launch", 0)
    for d in (a, hog):
        d.set_option("asynchronous", 1)         # (each context has its own stream: the launches of the two overlap on the device)
    b.step(w["dt"], 36, finalize=False)
    b.step(w["dt"], 1, finalize=True)
    hog.step(big["dt"], 2, finalize=False)
    for n in range(36):
        if n % 3 == 0:
            hog.step(big["dt"], 1, finalize=False)      # ~200 us of a foreign streaming kernel under ~7 LandModel launches
        a.step(w["dt"], 1, finalize=False)
    a.step(w["dt"], 1, finalize=True)
    a.synchronize(); hog.synchronize()
    assert family(a) == PROGRAM_LAND
    assert_same(a, b, w)
    assert a.status() & 4 == 0 and hog.status() == 0


def test_a_hand_off_that_never_comes_ends_in_a_status_flag_not_in_a_hang():
    """The column waves' wait for their surface fluxes is bounded: with the surface workgroups publishing under another tag
    (TRM_DEBUG_HANDOFF_TAG_BIAS=1, a switch of the library for this test) no word ever matches -- every column wave must give up after
    its spin limit, flag TRM_STATUS_HANDOFF_TIMEOUT (4), leave NaN where the fluxes enter (the top cell's internal energy, the skin temperature), and the
    launch must RETURN.  In a fresh process: the
    switch is read once per process."""
    import subprocess, sys, os, json
    code = (
        "import sys, json, time\n"
        "sys.path[:0] = [%r, %r]\n"
        "import numpy as np\nimport workloads as W\n"
        "from test_gpu_surface_in_launch import small_columns, family\n"
        "out = {}\n"
        "for dtype in ('float64', 'float32'):\n"
        "    lat, lon = small_columns(700)\n"
        "    w = W.make_workload('land', lat, lon, 32, dtype=getattr(np, dtype))\n"
        "    d = W.setup_device(w)\n"
        "    d.set_option('surface_in_launch', 1)\n"
        "    d.step(w['dt'], 1, finalize=False)\n"            # (the launch pair: the fields are read)
        "    assert d.status() == 0\n"
        "    t0 = time.time()\n"
        "    d.step(w['dt'], 1, finalize=False)\n"
        "    d.synchronize()\n"
        "    out[dtype] = dict(seconds=time.time() - t0, family=family(d), status=d.status(), nan=bool(np.isnan(d.get('internal_energy')[-1]).all() and np.isnan(d.get('skin_temperature')).all()))\n"
        "print(json.dumps(out))\n" % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    env = dict(os.environ, TRM_DEBUG_HANDOFF_TAG_BIAS="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    for dtype, fam in (("float64", PROGRAM_LAND), ("float32", PROGRAM_PACKED_LAND)):
        o = out[dtype]
        assert o["family"] == fam, o
        assert o["status"] & 4, o
        assert o["nan"], o
        assert o["seconds"] < 5.0, o
