"""CPU-only checks: the C ABI library loads and exports every symbol include/terrarium_hip.h declares
(no compute without a GPU), and the host-side mirror of the reference interface behaves."""
import ctypes
import os
import re

import numpy as np
import pytest

import terrarium_jl_amd as trm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "terrarium_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(trm_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = declared_symbols()
    assert len(names) >= 30
    lib = ctypes.CDLL(trm._capi.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"libterrarium_hip.so does not export {n}"
    assert sorted(trm._capi.EXPORTS) == names  # the Python binding covers the whole ABI, nothing more
    header = open(os.path.join(ROOT, "include", "terrarium_hip.h")).read()
    assert lib.trm_abi_version() == int(re.search(r"#define\s+TRM_ABI_VERSION\s+(\d+)", header).group(1))


def test_default_params_match_reference_defaults():
    p = trm._capi.default_params()
    # SURVEY Appendix A-0
    assert (p.rho_w, p.Lsl, p.Tref, p.sigma) == (1000.0, 3.34e5, 273.15, 5.6704e-8)
    assert (p.k_water, p.k_ice, p.k_air, p.k_mineral, p.k_organic) == (0.57, 2.2, 0.025, 3.8, 0.25)
    assert (p.c_water, p.c_ice, p.c_air, p.c_mineral, p.c_organic) == (4.2e6, 1.9e6, 1.25e3, 2.0e6, 2.5e6)
    assert (p.por_mineral, p.por_organic, p.K_sat, p.bc_psi_s, p.bc_lambda) == (0.49, 0.9, 1.0e-5, 0.01, 0.2)
    assert (p.albedo, p.emissivity, p.kappa_s, p.C_h, p.tau_r) == (0.3, 0.97, 2.0, 1.2e-3, 3600.0)
    assert (p.flow, p.swrc, p.unsat_k, p.seb, p.halo_policy) == (0, 0, 0, 0, 0)
    import oracle
    q = oracle.default_params()
    for name, _ in trm._capi.TrmParams._fields_:
        assert getattr(p, name) == getattr(q, name), name  # product and oracle agree on every default


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(trm.TerrariumHipError, match="no CPU fallback"):
        trm.initialize(trm.SoilModel(trm.ColumnGrid(trm.ExponentialSpacing(N=10), 4)))


def test_flatten_models():
    grid = trm.ColumnGrid(trm.ExponentialSpacing(N=10), 3)
    p = trm.flatten(trm.SoilModel(grid))
    assert (p.flow, p.seb, p.swrc, p.unsat_k) == (0, 0, 0, 0)
    hp = trm.ConstantSoilHydraulics(swrc=trm.VanGenuchten(alpha=2.0, n=2.0), unsat_hydraulic_cond=trm.UnsatKVanGenuchten())
    soil = trm.SoilEnergyWaterCarbon(hydrology=trm.SoilHydrology(vertical_flow=trm.RichardsEq(), hydraulic_properties=hp,
                                                                 vwc_forcing=-1e-5))
    p = trm.flatten(trm.LandModel(grid, soil=soil))
    assert (p.flow, p.seb, p.swrc, p.unsat_k, p.vg_alpha, p.vg_n, p.vwc_forcing) == (1, 1, 1, 1, 2.0, 2.0, -1e-5)
    with pytest.raises(ValueError):
        bad = trm.ConstantSoilHydraulics(swrc=trm.BrooksCorey(), unsat_hydraulic_cond=trm.UnsatKVanGenuchten())
        trm.flatten(trm.SoilModel(grid, soil=trm.SoilEnergyWaterCarbon(hydrology=trm.SoilHydrology(hydraulic_properties=bad))))


def test_column_ring_grid_scatter_gather():
    # test/grids.jl:44-139: scatter/gather identity between the ring grid and the column vector
    mask = trm.masks.load_land_mask("N72")
    grid = trm.ColumnRingGrid(trm.ExponentialSpacing(N=30), mask)
    assert grid.Nh == 14017 and grid.Nz == 30
    cols = np.arange(grid.Nh, dtype=np.float64)
    full = grid.scatter(cols)
    assert full.shape == mask.shape and np.isnan(full[~mask]).all()
    assert np.array_equal(grid.gather(full), cols)
    lat, lon = trm.masks.masked_latlon(mask)
    assert lat.shape == (14017,) and np.all(np.abs(lat) < np.pi / 2) and np.all((0 <= lon) & (lon < 2 * np.pi))
    assert lat[0] > lat[-1]  # ring order runs north -> south


def test_bc_aliases():
    bcs = trm.merge_boundary_conditions(trm.PrescribedSurfaceTemperature("T_ub", 1.0), trm.GeothermalHeatFlux(0.05),
                                        trm.FreeDrainage(), trm.ImpermeableBoundary())
    assert bcs[("temperature", "top")] == ("value", 1.0)
    assert bcs[("internal_energy", "bottom")] == ("flux", 0.05)
    assert bcs[("pressure_head", "bottom")] == ("gradient", 0.0)
    assert bcs[("saturation_water_ice", "bottom")][0] == "noflux"


def test_ring_grid_scatter_gather_2d_and_3d():
    """column_ring_grid.jl:102-149: RingGrids.Field(field, grid; fill_value) and Field(ring_field, grid)."""
    mask = trm.masks.load_land_mask("N72")
    grid = trm.ColumnRingGrid(trm.ExponentialSpacing(N=5), mask)
    assert grid.num_columns == 14017
    cols = np.arange(grid.num_columns, dtype=np.float64)
    full = grid.scatter(cols)
    assert full.shape == mask.shape and np.isnan(full[~mask]).all() and np.array_equal(full[mask], cols)
    assert np.array_equal(grid.gather(full), cols)
    f3 = np.arange(5 * grid.num_columns, dtype=np.float32).reshape(5, -1)
    full3 = grid.scatter(f3, fill=-1.0)
    assert full3.shape == (5,) + mask.shape and full3.dtype == np.float32 and (full3[:, ~mask] == -1.0).all()
    assert np.array_equal(grid.gather(full3), f3)
    with pytest.raises(ValueError):
        grid.scatter(np.zeros(7))
    with pytest.raises(ValueError):
        grid.gather(np.zeros((3, 4)))


def test_graft_entry_build_runs():
    """The driver's build() entry point: compiles (no-op when up to date), imports, checks the exported ABI."""
    import __graft_entry__ as g
    g.build()


# src/processes/soil/stratigraphy/soil_texture.jl:6-20, soil_hydraulic_properties.jl:112-156 (parameter structs of a12)
def test_soil_texture_and_surfex_hydraulics():
    t = trm.SoilTexture()
    assert (t.sand, t.clay, t.silt) == (1.0, 0.0, 0.0)
    t = trm.SoilTexture(sand=0.4, clay=0.25)
    assert t.silt == pytest.approx(0.35)
    with pytest.raises(AssertionError):
        trm.SoilTexture(sand=0.8, clay=0.5)
    with pytest.raises(AssertionError):
        trm.SoilTexture(sand=0.5, clay=0.2, silt=0.1)
    sx = trm.SoilHydraulicsSURFEX()
    assert sx.wilting_point(t) == pytest.approx(37.13e-3 * (25.0) ** 0.5)
    assert sx.field_capacity(t) == pytest.approx(89.0e-3 * 25.0 ** 0.35)
    assert sx.saturated_hydraulic_conductivity() == 1.0e-5
    ch = trm.ConstantSoilHydraulics()
    assert (ch.field_capacity(), ch.wilting_point()) == (0.25, 0.05)
    # both parameterisations flatten to the same step parameters
    grid = trm.ColumnGrid(trm.ExponentialSpacing(N=10))
    a = trm.flatten(trm.SoilModel(grid, soil=trm.SoilEnergyWaterCarbon(hydrology=trm.SoilHydrology(vertical_flow=trm.RichardsEq(), hydraulic_properties=sx))))
    b = trm.flatten(trm.SoilModel(grid, soil=trm.SoilEnergyWaterCarbon(hydrology=trm.SoilHydrology(vertical_flow=trm.RichardsEq(), hydraulic_properties=ch))))
    for name, _ in trm._capi.TrmParams._fields_:
        assert getattr(a, name) == getattr(b, name), name
    land = trm.LandModel(grid, surface_energy_balance=trm.SurfaceEnergyBalance(albedo=trm.PrescribedAlbedo()))
    assert trm.flatten(land).prescribed_albedo == 1 and trm.flatten(trm.LandModel(grid)).prescribed_albedo == 0


def test_simulation_schedules():
    """IterationInterval / TimeInterval of the Simulation driver (Oceananigans.Utils schedules)."""
    s = trm.IterationInterval(3)
    assert [s.actuates(0.0, i) for i in range(7)] == [True, False, False, True, False, False, True]
    assert s.steps_until_next(0.0, 4, 60.0) == 2 and s.steps_until_next(0.0, 6, 60.0) == 3
    t = trm.TimeInterval(1800.0)
    assert t.next_time() == 1800.0 and t.steps_until_next(0.0, 0, 600.0) == 3 and t.steps_until_next(1500.0, 0, 600.0) == 1
    assert not t.actuates(1200.0, 2)
    assert t.actuates(1800.0, 3) and t.next_time() == 3600.0
    assert not t.actuates(3000.0, 5) and t.actuates(3700.0, 6) and t.next_time() == 5400.0


def test_python_constants_match_the_header():
    """Field ids, option ids, error codes and the vegetation parameter layout of terrarium.jl_amd/_capi.py and oracle/oracle.py
    against include/terrarium_hip.h (the header is the contract; the Python tables are copies)."""
    import re
    import oracle
    from terrarium_jl_amd import _capi
    header = open(os.path.join(ROOT, "include", "terrarium_hip.h")).read()
    enum = {m.group(1): int(m.group(2)) for m in re.finditer(r"\b(TRM_[A-Z0-9_]+)\s*=\s*(\d+)", header)}
    alias = dict(SAI="STEM_AREA_INDEX", tend_internal_energy="TEND_INTERNAL_ENERGY")
    for name, fid in _capi.FIELD.items():
        key = "TRM_FIELD_" + alias.get(name, name).upper()
        assert enum[key] == fid, (name, key)
    assert enum["TRM_FIELD_COUNT"] == max(_capi.FIELD.values()) + 1 == len(_capi.FIELD)
    for name, fid in oracle.FIELDS.items():          # the oracle's table is a subset with the same ids
        assert _capi.FIELD[name] == fid, name
    for name, oid in _capi.OPTION.items():
        key = {"asynchronous": "TRM_OPT_ASYNC", "step_kernel": "TRM_OPT_STEP_KERNEL", "write_kf_every_step": "TRM_OPT_WRITE_KF_EVERY_STEP",
               "vwc_forcing_field": "TRM_OPT_VWC_FORCING_FIELD", "packed_f32": "TRM_OPT_PACKED_F32",
               "derive_closure_fields": "TRM_OPT_DERIVE_CLOSURE_FIELDS", "steps_per_launch": "TRM_OPT_STEPS_PER_LAUNCH",
               "pipeline_parts": "TRM_OPT_PIPELINE_PARTS", "single_step_program": "TRM_OPT_SINGLE_STEP_PROGRAM",
               "info_top_arrays_current": "TRM_INFO_TOP_ARRAYS_CURRENT", "info_closure_consistent": "TRM_INFO_CLOSURE_CONSISTENT",
               "bc_signature": "TRM_OPT_BC_SIGNATURE", "info_bc_signature": "TRM_INFO_BC_SIGNATURE",
               "zero_gradient_fast": "TRM_OPT_ZERO_GRADIENT_FAST", "info_generic_boundary_kernels": "TRM_INFO_GENERIC_BOUNDARY_KERNELS",
               "surface_in_launch": "TRM_OPT_SURFACE_IN_LAUNCH", "info_last_program": "TRM_INFO_LAST_PROGRAM"}[name]
        assert enum[key] == oid, name
    for code, key in enumerate(("TRM_OK", "TRM_EINVAL", "TRM_EHIP", "TRM_ENOMEM", "TRM_EUNSUPPORTED", "TRM_ESTALE", "TRM_ECOMM")):
        assert enum[key] == code
    assert _capi.VEGETATION == dict(off=enum["TRM_VEGETATION_OFF"], standalone=enum["TRM_VEGETATION_STANDALONE"], coupled=enum["TRM_VEGETATION_COUPLED"])
    # vegetation parameter struct: the doubles of the header in order
    body = header[header.index("typedef struct trm_vegetation_params {"):header.index("} trm_vegetation_params;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = [n.strip() for decl in re.findall(r"double([^;]+);", body) for n in decl.split(",")]
    assert names == list(_capi.VEG_PARAM_NAMES) == [f[0] for f in oracle.VegParamsD._fields_]
    assert int(re.search(r"#define TRM_ABI_VERSION (\d+)", header).group(1)) == _capi.lib().trm_abi_version()


# test/utils.jl:50-59 (piecewise_linear) and soil_model_init.jl:86-114
def test_piecewise_linear_initializer():
    f = trm.piecewise_linear((1.0, 1.0), (0.0, -1.0), (-2.0, -2.0))
    assert f(2.0) == pytest.approx(1.0) and f(1.0) == pytest.approx(1.0) and f(0.5) == pytest.approx(0.0)
    assert f(-1.0) == pytest.approx(-1.5) and f(-3.0) == pytest.approx(-2.0)
    with pytest.raises(ValueError):
        trm.piecewise_linear((0.0, 1.0), (1.0, 2.0))
    init = trm.PiecewiseLinearInitialSoilTemperature((0.0, 5.0), (-0.5, 2.0), (-1.0, 1.0), (-10.0, 1.5))
    assert init.knots[1] == (-0.5, 2.0)


# test/soil/soil_stratigrapy_tests.jl:4-19
def test_soil_porosity_surfex_reference_test():
    porosity = trm.SoilPorositySURFEX()
    por0 = porosity.mineral(trm.SoilTexture(sand=0.0, silt=0.7, clay=0.3))
    assert por0 == pytest.approx(porosity.porosity_default)
    for sand in np.arange(0.1, 1.01, 0.1):
        por = porosity.mineral(trm.SoilTexture(sand=sand, silt=(1 - sand) * 0.7, clay=(1 - sand) * 0.3))
        assert 0 < por < por0


# soil_porosity.jl:30-50
def test_soil_porosity_surfex():
    grid = trm.ColumnGrid(trm.ExponentialSpacing(N=10))
    strat = trm.HomogeneousStratigraphy(porosity=trm.SoilPorositySURFEX(), texture=trm.SoilTexture(sand=0.4, clay=0.2))
    p = trm.flatten(trm.SoilModel(grid, soil=trm.SoilEnergyWaterCarbon(strat=strat)))
    assert p.por_mineral == 0.49 + (-0.11) * 0.4 and p.por_organic == 0.9


def test_exported_interface_helpers():
    """default_dt / is_adaptive (forward_euler.jl:13-15, heun.jl:16-18), znodes / zspacings of the column grid."""
    g = trm.ColumnGrid(trm.ExponentialSpacing(N=5))
    assert trm.default_dt(trm.ForwardEuler()) == 300.0 == trm.default_dt(trm.Heun()) and not trm.is_adaptive(trm.Heun())
    zc, zf, dz = trm.znodes(g), trm.znodes(g, "face"), trm.zspacings(g)
    assert zf[-1] == 0.0 and np.all(np.diff(zf) > 0) and np.allclose(np.diff(zf), dz) and np.allclose(zc, (zf[1:] + zf[:-1]) / 2)
    assert trm.get_grid(trm.SoilModel(g)) is g


def _build_abi_host(tmp_path):
    import subprocess
    exe = str(tmp_path / "abi_host")
    subprocess.check_call(["gcc", "-O1", "-Wall", "-Werror", "-std=c99", "-o", exe, os.path.join(ROOT, "tests", "abi_host.c"), "-ldl"])
    return exe


def _julia_struct_fields(name):
    """(field, type) pairs of a Julia struct printed in INTEGRATION.md's shim, in declaration order."""
    import re
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"(?:mutable )?struct " + name + r"\n(.*?)\n\s*(?:" + name + r"\(\) = new\(\)\n)?end", text, re.S)
    assert m, name
    return re.findall(r"(\w+)::(\w+)", m.group(1))


def test_c_host_sees_the_struct_layout_the_julia_shim_declares(tmp_path):
    """tests/abi_host.c -- a plain C program built with gcc against include/terrarium_hip.h that dlopens the library -- prints
    sizeof / offsetof of trm_grid, trm_params, trm_vegetation_params.  They must equal (a) the layout a Julia `ccall` derives
    from the structs INTEGRATION.md declares (C layout rules applied to the declared field order and types) and (b) the ctypes
    mirror the Python tests drive the library through."""
    import subprocess
    from terrarium_jl_amd import _capi
    exe = _build_abi_host(tmp_path)
    out = subprocess.check_output([exe, "layout", _capi.LIB_PATH], text=True).split("\n")
    off, size, sizeof, misc = {}, {}, {}, {}
    for line in out:
        w = line.split()
        if not w:
            continue
        if w[0] == "sizeof":
            sizeof[w[1]] = int(w[2])
        elif w[0] in ("abi_version", "default"):
            misc[tuple(w[:-1]) if w[0] == "default" else w[0]] = w[1:] if w[0] == "abi_version" else w[-1]
        else:
            off.setdefault(w[0], []).append((w[1], int(w[2])))
            size.setdefault(w[0], {})[w[1]] = int(w[3])
    assert misc["abi_version"][0] == misc["abi_version"][1]          # the library and the header agree
    ctype_size = dict(Cint=4, Int64=8, Cdouble=8, Ptr=8)
    for cstruct, jstruct, ct in (("trm_grid", "TrmGrid", _capi.TrmGrid), ("trm_params", "TrmParams", _capi.TrmParams),
                                 ("trm_vegetation_params", "TrmVegetationParams", _capi.TrmVegetationParams)):
        # (a) the Julia declaration: natural alignment, declaration order
        pos, expect = 0, []
        for fname, ftype in _julia_struct_fields(jstruct):
            n = ctype_size[ftype]
            pos = (pos + n - 1) // n * n
            expect.append((fname, pos))
            pos += n
        assert expect == off[cstruct], (cstruct, [a for a, b in zip(expect, off[cstruct]) if a != b][:3])
        assert (pos + 7) // 8 * 8 == sizeof[cstruct]
        # (b) ctypes
        assert [(f[0], getattr(ct, f[0]).offset) for f in ct._fields_] == off[cstruct]
        assert C_sizeof(ct) == sizeof[cstruct]
    # defaults written through the C struct land where the C host reads them
    assert float(misc[("default", "rho_w")]) == 1000.0 and float(misc[("default", "K_sat")]) == 1.0e-5
    assert float(misc[("default", "field_capacity")]) == 0.25 and misc[("default", "halo_policy")] == "0" and misc[("default", "reserved")] == "0"
    assert float(misc[("default", "tau25")]) == 2600.0 and float(misc[("default", "C_can")]) == 0.006


def C_sizeof(ct):
    import ctypes
    return ctypes.sizeof(ct)


def test_loading_the_library_first_leaves_one_hip_runtime_for_torch():
    """PyTorch-ROCm bundles its own libamdhip64.so / libhsa-runtime64.so; the loader makes the library bind to that copy even
    when nothing imported torch yet, so a later `import torch` does not bring a second runtime into the process (which then
    finds no GPU).  No GPU needed: only the loaded objects are inspected."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import terrarium_jl_amd as trm\n"
            "assert 'torch' not in sys.modules\n"
            "trm._capi.lib()\n"
            "import torch\n"
            "maps = open('/proc/self/maps').read().split('\\n')\n"
            "print(len({l.split()[-1] for l in maps if 'libamdhip64' in l}), len({l.split()[-1] for l in maps if 'libhsa-runtime64' in l}))\n") % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split()[-2:] == ["1", "1"], out.stdout
