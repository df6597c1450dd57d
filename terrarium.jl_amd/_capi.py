"""ctypes binding of libterrarium_hip.so -- exactly the entry points declared in
include/terrarium_hip.h (the stub a Julia maintainer would write with `ccall`
is in INTEGRATION.md).  There is no CPU fallback: if the library is missing the
import of this module's `lib()` raises, and `trm_create` fails without a GPU."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TRM_LIBRARY", os.path.join(_HERE, "libterrarium_hip.so"))  # override: kernel-tuning experiments

TRM_F64, TRM_F32 = 0, 1
FLOW = dict(noflow=0, richards=1)
SWRC = dict(brooks_corey=0, van_genuchten=1)
UNSATK = dict(linear=0, van_genuchten=1)
HALO = dict(reference_zero=0, mirror=1)
BC_KIND = dict(noflux=0, value=1, flux=2, gradient=3)
SIDE = dict(bottom=0, top=1)
BC_VAR = dict(internal_energy=0, saturation_water_ice=1, temperature=2, liquid_water_fraction=3, pressure_head=4)
FIELD = dict(
    internal_energy=0, saturation_water_ice=1, temperature=2, liquid_water_fraction=3, pressure_head=4,
    hydraulic_conductivity=5, tend_internal_energy=6, tend_saturation_water_ice=7, surface_excess_water=8,
    tend_surface_excess_water=9, water_table=10, skin_temperature=11, ground_heat_flux=12, surface_shortwave_up=13,
    surface_longwave_up=14, surface_net_radiation=15, sensible_heat_flux=16, latent_heat_flux=17,
    evaporation_ground=18, infiltration=19, surface_runoff=20, air_temperature=21, air_pressure=22, windspeed=23,
    specific_humidity=24, rainfall=25, surface_shortwave_down=26, surface_longwave_down=27, vwc_forcing=28,
    albedo=29, emissivity=30,
    carbon_vegetation=31, vegetation_area_fraction=32, tend_carbon_vegetation=33, tend_vegetation_area_fraction=34,
    balanced_leaf_area_index=35, phenology_factor=36, leaf_area_index=37, canopy_water_conductance=38, leaf_to_air_co2_ratio=39,
    net_assimilation=40, leaf_respiration=41, gross_primary_production=42, autotrophic_respiration=43, net_primary_production=44,
    CO2=45, soil_moisture_limiting_factor=46, daily_leaf_respiration=47, vegetation_ground_temperature=48,
    plant_available_water=49, root_fraction=50,
    canopy_water=51, tend_canopy_water=52, canopy_water_interception=53, canopy_water_removal=54, saturation_canopy_water=55,
    rainfall_ground=56, evaporation_canopy=57, transpiration=58, SAI=59,
)
INPUT_FIELDS = ("air_temperature", "air_pressure", "windspeed", "specific_humidity", "rainfall",
                "surface_shortwave_down", "surface_longwave_down", "albedo", "emissivity", "CO2",
                "soil_moisture_limiting_factor", "daily_leaf_respiration", "vegetation_ground_temperature", "SAI")
VEGETATION = dict(off=0, standalone=1, coupled=2)
VEG_PARAM_NAMES = ("tau25 Kc25 Ko25 q10_tau q10_Kc q10_Ko alpha_leaf alpha_a alpha_C3 cq k_ext T_CO2_high T_CO2_low T_photos_high "
                   "T_photos_low theta_r g1 g_min cn_sapwood cn_root aws SLA awl LAI_min LAI_max gamma_L gamma_R gamma_S nu_seed "
                   "gamma_v_min root_a root_b wilting_point field_capacity C_mass alpha_int canopy_k_ext w_can_max tau_w C_can").split()
REDUCE = dict(sum=0, min=1, max=2, hasnan=3, volume_integral_z=4)
OPTION = dict(asynchronous=0, step_kernel=1, write_kf_every_step=2, vwc_forcing_field=3, packed_f32=4,
              derive_closure_fields=5, steps_per_launch=6, pipeline_parts=7, single_step_program=8, bc_signature=9, zero_gradient_fast=10, surface_in_launch=11,
              info_top_arrays_current=100, info_closure_consistent=101, info_bc_signature=102, info_generic_boundary_kernels=103,
              info_last_program=105)
KERNEL = dict(fused=0, unfused=1)
# TRM_INFO_LAST_PROGRAM (include/terrarium_hip.h: TRM_PROGRAM_*; trm_host.hpp: program_id)
PROGRAM = ("none", "column_euler", "column_heun", "column_multi", "packed_f32", "generic_euler", "generic_heun", "column_land", "deep", "wide",
           "land_interleaved", "unfused", "vegetation", "packed_land")
DERIVE = ("none", "T_liq", "liq", "liq_psi", "all")


def decode_program(pid: int) -> dict:
    d = dict(family=PROGRAM[pid & 0xff], hydraulics=("default", "vg_n2", "generic")[(pid >> 8) & 3], lanes_per_column=32 * ((pid >> 10) & 3),
             derive=DERIVE[(pid >> 12) & 7], staged=bool((pid >> 15) & 1), scalar_inputs=bool((pid >> 16) & 1), bc_signature=((pid >> 17) & 0xff) - 1)
    extra = pid >> 25
    if d["family"] == "column_multi":
        d.update(surface_inline=bool(extra & 1), series=bool(extra & 2))
    if d["family"] == "column_land":
        d.update(program=("euler", "heun", "multi")[extra & 3])
    if d["family"] in ("deep", "wide"):
        d.update(program=("euler", "heun", "multi")[extra & 3], generic_boundaries=bool(extra & 4))
    return d
STATUS_NAN, STATUS_COMPOSITION, STATUS_HANDOFF_TIMEOUT = 1, 2, 4
TRM_OK, TRM_EINVAL, TRM_EHIP, TRM_ENOMEM, TRM_EUNSUPPORTED, TRM_ESTALE, TRM_ECOMM = range(7)

EXPORTS = (
    "trm_abi_version trm_default_params trm_create trm_destroy trm_last_error trm_field_rows trm_get_grid "
    "trm_upload trm_download trm_field_device_ptr trm_bc_device_ptr trm_set_bc trm_set_forcing trm_initialize trm_update_state "
    "trm_compute_auxiliary trm_compute_tendencies trm_reset_tendencies trm_explicit_step trm_closure trm_invclosure "
    "trm_step trm_step_heun trm_step_timed trm_step_heun_timed trm_clock trm_set_clock trm_reduce trm_status trm_set_status trm_set_option "
    "trm_get_option trm_set_stream trm_synchronize "
    "trm_set_forcing_series trm_set_bc_series trm_clear_series trm_update_inputs trm_save_state trm_restore_state "
    "trm_comm_unique_id trm_comm_init trm_comm_destroy trm_comm_info trm_reduce_global trm_status_global "
    "trm_default_vegetation_params trm_set_vegetation trm_compute_plant_available_water "
    "trm_series_append trm_series_trim_before trm_series_info trm_reset trm_download_rows trm_set_ring_grid trm_download_ring "
    "trm_scatter_ring_device trm_upload_ring trm_gather_ring_device "
    "trm_heun_predict trm_heun_stage_auxiliary trm_heun_correct trm_stage_field_device_ptr trm_stage_bc_device_ptr trm_set_forcing_device "
    "trm_series_window trm_comm_init_all trm_step_all trm_step_heun_all trm_synchronize_all trm_reduce_global_all trm_status_global_all").split()
TIME_INDEXING = dict(linear=0, clamp=1, cyclical=2, raster=3)


class TrmGrid(C.Structure):
    _fields_ = [("precision", C.c_int32), ("num_layers", C.c_int32), ("num_columns", C.c_int64),
                ("thickness", C.POINTER(C.c_double)), ("dx", C.c_double), ("device", C.c_int32),
                ("reserved", C.c_int32)]


class TrmParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "rho_w rho_i rho_a c_a Lsl Llg Lsg g Tref sigma kappa_vk eps_mw R_a "
        "k_water k_ice k_air k_mineral k_organic c_water c_ice c_air c_mineral c_organic "
        "por_mineral por_organic rho_soc rho_org "
        "K_sat theta_res bc_psi_s bc_lambda vg_alpha vg_n impedance vwc_forcing "
        "albedo emissivity kappa_s C_h min_windspeed tau_r beta_evap field_capacity").split()] + [
        (n, C.c_int32) for n in "flow swrc unsat_k seb halo_policy prescribed_albedo evap_resistance reserved".split()]


class TrmVegetationParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in VEG_PARAM_NAMES]


class TerrariumHipError(RuntimeError):
    code = None


_lib = None


def _share_hip_runtime_with_torch():
    """One HIP / HSA runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7) and
    libhsa-runtime64.so; the library is linked against /opt/rocm's libamdhip64.so.7.  Loaded after torch, the library binds
    to torch's copy (same SONAME) and both see the device; loaded BEFORE torch, the process ends up with two HSA runtimes
    and the second one finds no GPU ("No HIP GPUs are available").  So, when PyTorch is installed and no HIP runtime is in
    the process yet, torch's copy is loaded first -- without importing torch."""
    try:
        with open("/proc/self/maps") as f:
            if "libamdhip64" in f.read():
                return
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(path):
            C.CDLL(path, mode=C.RTLD_GLOBAL)
    except OSError:
        pass    # (the library's own runpath still finds /opt/rocm's runtime)


def lib():
    """Load libterrarium_hip.so; raise loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TerrariumHipError(
            f"{LIB_PATH} is missing: build it with `make -C terrarium.jl_amd/csrc` "
            "(or __graft_entry__.build()).  There is no CPU fallback.")
    _share_hip_runtime_with_torch()
    L = C.CDLL(LIB_PATH)
    if "TRM_LIBRARY" in os.environ:
        # an older build of the library in an A/B experiment (profiles/tools): entry points it lacks bind to a stub that fails when
        # CALLED; the shipped library must export every one of EXPORTS (__graft_entry__.build() checks)
        def _missing(name):
            def stub(*args):
                raise TerrariumHipError(f"{LIB_PATH} (TRM_LIBRARY) does not export {name}")
            return stub
        for name in EXPORTS:
            try:
                getattr(L, name)
            except AttributeError:
                setattr(L, name, _missing(name))
    vp, i32, i64, dbl = C.c_void_p, C.c_int, C.c_int64, C.c_double
    L.trm_abi_version.restype = i32
    L.trm_default_params.argtypes = [C.POINTER(TrmParams)]
    L.trm_create.argtypes = [C.POINTER(TrmGrid), C.POINTER(TrmParams), C.POINTER(vp)]
    L.trm_destroy.argtypes = [vp]
    L.trm_last_error.restype = C.c_char_p
    L.trm_last_error.argtypes = [vp]
    L.trm_field_rows.argtypes = [vp, i32, C.POINTER(i64)]
    L.trm_get_grid.argtypes = [vp, vp, vp, vp, vp]
    L.trm_upload.argtypes = [vp, i32, vp]
    L.trm_download.argtypes = [vp, i32, vp]
    L.trm_field_device_ptr.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(i64)]
    L.trm_bc_device_ptr.argtypes = [vp, i32, i32, C.POINTER(vp)]
    L.trm_set_bc.argtypes = [vp, i32, i32, i32, vp, dbl]
    L.trm_set_forcing.argtypes = [vp, i32, vp]
    L.trm_set_forcing_series.argtypes = [vp, i32, i32, vp, vp, i32]
    L.trm_set_bc_series.argtypes = [vp, i32, i32, i32, i32, vp, vp, i32]
    L.trm_clear_series.argtypes = [vp]
    L.trm_update_inputs.argtypes = [vp]
    L.trm_save_state.argtypes = [vp]
    L.trm_restore_state.argtypes = [vp]
    for name in ("trm_initialize", "trm_compute_auxiliary", "trm_compute_tendencies", "trm_reset_tendencies",
                 "trm_closure", "trm_invclosure", "trm_synchronize"):
        getattr(L, name).argtypes = [vp]
    L.trm_update_state.argtypes = [vp, i32]
    L.trm_explicit_step.argtypes = [vp, dbl]
    L.trm_step.argtypes = [vp, dbl, i32, i32]
    L.trm_step_heun.argtypes = [vp, dbl, i32, i32]
    L.trm_step_timed.argtypes = [vp, dbl, i32, i32, C.POINTER(C.c_float)]
    L.trm_step_heun_timed.argtypes = [vp, dbl, i32, i32, C.POINTER(C.c_float)]
    L.trm_clock.argtypes = [vp, C.POINTER(dbl), C.POINTER(i64)]
    L.trm_set_clock.argtypes = [vp, dbl, i64]
    L.trm_reduce.argtypes = [vp, i32, i32, vp]
    L.trm_status.argtypes = [vp, C.POINTER(C.c_uint32)]
    L.trm_set_option.argtypes = [vp, i32, i32]
    L.trm_get_option.argtypes = [vp, i32, C.POINTER(i32)]
    L.trm_set_stream.argtypes = [vp, vp]
    L.trm_default_vegetation_params.argtypes = [C.POINTER(TrmVegetationParams)]
    L.trm_set_vegetation.argtypes = [vp, C.POINTER(TrmVegetationParams), i32]
    L.trm_compute_plant_available_water.argtypes = [vp]
    L.trm_comm_unique_id.argtypes = [vp]
    L.trm_comm_init.argtypes = [vp, i32, i32, vp]
    L.trm_comm_destroy.argtypes = [vp]
    L.trm_comm_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.trm_reduce_global.argtypes = [vp, i32, i32, vp]
    L.trm_status_global.argtypes = [vp, C.POINTER(C.c_uint32)]
    L.trm_series_append.argtypes = [vp, i32, i32, i32, i32, vp, vp]
    L.trm_series_trim_before.argtypes = [vp, dbl]
    L.trm_series_window.argtypes = [vp, i32, i32, i32, i32]
    L.trm_series_info.argtypes = [vp, i32, i32, i32, C.POINTER(i64), C.POINTER(i64), C.POINTER(dbl), C.POINTER(dbl)]
    L.trm_reset.argtypes = [vp]
    L.trm_download_rows.argtypes = [vp, i32, i32, i32, vp]
    L.trm_set_ring_grid.argtypes = [vp, i64, vp]
    L.trm_download_ring.argtypes = [vp, i32, i32, i32, dbl, vp]
    L.trm_scatter_ring_device.argtypes = [vp, i32, i32, i32, dbl, vp]
    L.trm_upload_ring.argtypes = [vp, i32, vp]
    L.trm_gather_ring_device.argtypes = [vp, i32, vp]
    L.trm_heun_predict.argtypes = [vp, dbl]
    L.trm_heun_stage_auxiliary.argtypes = [vp]
    L.trm_heun_correct.argtypes = [vp, dbl, i32]
    L.trm_stage_field_device_ptr.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(i64)]
    L.trm_stage_bc_device_ptr.argtypes = [vp, i32, i32, C.POINTER(vp)]
    L.trm_set_forcing_device.argtypes = [vp, i32, vp]
    L.trm_comm_init_all.argtypes = [C.POINTER(vp), i32]
    L.trm_step_all.argtypes = [C.POINTER(vp), i32, dbl, i32, i32]
    L.trm_step_heun_all.argtypes = [C.POINTER(vp), i32, dbl, i32, i32]
    L.trm_synchronize_all.argtypes = [C.POINTER(vp), i32]
    L.trm_reduce_global_all.argtypes = [C.POINTER(vp), i32, i32, i32, vp]
    L.trm_status_global_all.argtypes = [C.POINTER(vp), i32, C.POINTER(C.c_uint32)]
    for name in EXPORTS:
        if name not in ("trm_last_error",):
            getattr(L, name).restype = i32
    _lib = L
    return L


def check(ctx, rc, what):
    if rc != 0:
        msg = lib().trm_last_error(ctx)
        err = TerrariumHipError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")
        err.code = rc
        raise err


def default_vegetation_params() -> TrmVegetationParams:
    p = TrmVegetationParams()
    check(None, lib().trm_default_vegetation_params(C.byref(p)), "trm_default_vegetation_params")
    return p


def default_params() -> TrmParams:
    p = TrmParams()
    check(None, lib().trm_default_params(C.byref(p)), "trm_default_params")
    return p
