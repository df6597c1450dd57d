"""Which kernel instance each BASELINE configuration takes (TRM_INFO_LAST_PROGRAM).  The library chooses among ~250 instances of its
column programs from sizes, boundary kinds and options -- pure host logic whose mistakes cost speed, never correctness: a rule that
folds to the wrong branch leaves every parity test green (it happened: profiles/r04/exp1_..._VOID_...).  This pins, for every
configuration DESIGN.md's table names, the instance that table says it runs, at the configuration's full size."""
import numpy as np
import pytest

import terrarium_jl_amd as trm
import workloads as W

pytestmark = pytest.mark.gpu

T_TOP, LAND = 2, 64      # BCSIG bits (include/terrarium_hip.h: TRM_INFO_BC_SIGNATURE)


def steady_program(w, steps_per_launch=1, heun=False, nsteps=3):
    d = W.setup_device(w, steps_per_launch=steps_per_launch)
    first = None
    for n in range(nsteps):      # the first step after initialize reads T / liq as stored; the steady state follows
        (d.step_heun if heun else d.step)(w["dt"], 1 if steps_per_launch == 1 else 4, finalize=False)
        first = first or d.last_program()
    p = d.last_program()
    assert d.status() == 0
    d.close()
    return first, p


def columns(mask, copies=1):
    lat, lon = W.columns_from_mask(mask)
    return np.tile(lat, copies), np.tile(lon, copies)


def expect(p, **kw):
    for k, v in kw.items():
        assert p[k] == v, (k, p)


def test_c2_n72_heat_only():
    w = W.make_workload("heat", *columns("N72"), 30)
    first, p = steady_program(w)
    expect(p, family="column_euler", hydraulics="default", lanes_per_column=32, derive="none", staged=False, scalar_inputs=True, bc_signature=T_TOP)
    assert first == p
    _, m = steady_program(w, steps_per_launch=0)           # the library's default for run!: the resident program
    expect(m, family="column_multi", lanes_per_column=32, surface_inline=False, series=False)


def test_c3_n145_heat_and_richards():
    w = W.make_workload("richards", *columns("N145"), 32)
    first, p = steady_program(w)
    expect(first, family="column_euler", derive="none", bc_signature=T_TOP)       # (the uploaded T / liq are read once)
    expect(p, family="column_euler", hydraulics="default", lanes_per_column=32, derive="T_liq", staged=False, scalar_inputs=True, bc_signature=T_TOP)
    _, h = steady_program(w, heun=True)
    expect(h, family="column_heun", derive="none", bc_signature=T_TOP)
    wv = W.make_workload("richards", *columns("N145"), 32, hydraulics="vg")
    expect(steady_program(wv)[1], family="column_euler", hydraulics="vg_n2", derive="T_liq", bc_signature=T_TOP)


def test_c3_on_eight_copies_is_hbm_resident():
    w = W.make_workload("richards", *columns("N145", 8), 32)
    _, p = steady_program(w)
    expect(p, family="column_euler", derive="T_liq", staged=True, scalar_inputs=False, bc_signature=T_TOP)


@pytest.mark.parametrize("hydraulics", ["default", "vg"])
def test_c4_n145_land_model(hydraulics):
    w = W.make_workload("land", *columns("N145"), 32, hydraulics=hydraulics)
    first, p = steady_program(w)
    expect(first, family="column_euler", derive="none", bc_signature=LAND)       # k_surface + k_column: the top-cell arrays are not current yet
    expect(p, family="column_land", program="euler", hydraulics="default" if hydraulics == "default" else "vg_n2", lanes_per_column=32, derive="T_liq",
           staged=True, scalar_inputs=True, bc_signature=LAND)
    _, h = steady_program(w, heun=True)
    expect(h, family="column_land", program="heun", derive="none", bc_signature=LAND)      # (Heun in one launch, the state's surface processes inside it)


def test_c4_shard_of_one_of_eight_gpus():
    lat, lon = columns("N145")
    w = W.shard_workload(W.make_workload("land", lat, lon, 32), 0, 7119)
    _, p = steady_program(w)
    expect(p, family="column_land", derive="none", staged=False, scalar_inputs=True, bc_signature=LAND)
    _, m = steady_program(w, steps_per_launch=0)
    expect(m, family="column_multi", surface_inline=True)


def test_c5_fp32_shard():
    lat, lon = W.synthetic_columns(812500)
    w = W.make_workload("land", lat, lon, 64, dtype=np.float32)
    _, p = steady_program(w)
    expect(p, family="packed_f32", hydraulics="default", lanes_per_column=64, derive="liq", staged=False, bc_signature=LAND)      # (beyond the size where the single launch pays)
    _, s = steady_program(W.make_workload("land", lat[:12696], lon[:12696], 64, dtype=np.float32))
    expect(s, family="packed_land", derive="none", bc_signature=LAND)


def test_deep_columns():
    lat, lon = columns("N72")
    for Nz, family in ((100, "deep"), (160, "wide")):
        w = W.make_workload("richards", lat, lon, Nz)
        _, p = steady_program(w)
        expect(p, family=family, program="euler", generic_boundaries=False)
        _, h = steady_program(w, heun=True)
        expect(h, family=family, program="heun")
    _, m = steady_program(W.make_workload("richards", lat, lon, 100), steps_per_launch=0)
    expect(m, family="deep", program="multi")


def test_vegetation_coupled_and_generic_boundaries():
    lat, lon = columns("N72")
    wv = W.make_workload("landveg", lat, lon, 32, hydraulics="vg")
    expect(steady_program(wv)[1], family="column_euler", derive="none", bc_signature=LAND)      # k_surface_veg + k_column
    w = W.make_workload("richards", lat, lon, 32)
    w["bcs"][("pressure_head", "bottom")] = ("gradient", np.full(lat.size, 0.5))
    expect(steady_program(w)[1], family="generic_euler")
    w["bcs"][("pressure_head", "bottom")] = ("gradient", np.zeros(lat.size))                       # FreeDrainage(): the branch-free program
    expect(steady_program(w)[1], family="column_euler", bc_signature=T_TOP)
