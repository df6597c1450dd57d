"""ctypes harness for the CPU restatement in oracle/terrarium_oracle.hpp.

TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
and the cpu_baseline leg of bench.py may import this module; nothing under
terrarium.jl_amd/ does.  It never reads /root/reference at run time.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

# field ids (terrarium_oracle.hpp: enum FieldId)
FIELDS = dict(
    internal_energy=0, saturation_water_ice=1, temperature=2, liquid_water_fraction=3, pressure_head=4,
    hydraulic_conductivity=5, tend_internal_energy=6, tend_saturation_water_ice=7, surface_excess_water=8,
    tend_surface_excess_water=9, water_table=10, skin_temperature=11, ground_heat_flux=12, surface_shortwave_up=13,
    surface_longwave_up=14, surface_net_radiation=15, sensible_heat_flux=16, latent_heat_flux=17,
    evaporation_ground=18, infiltration=19, surface_runoff=20, air_temperature=21, air_pressure=22, windspeed=23,
    specific_humidity=24, rainfall=25, surface_shortwave_down=26, surface_longwave_down=27, vwc_forcing=28,
    albedo=29, emissivity=30,
    # LandModel with vegetation (ids of include/terrarium_hip.h)
    carbon_vegetation=31, vegetation_area_fraction=32, tend_carbon_vegetation=33, tend_vegetation_area_fraction=34,
    balanced_leaf_area_index=35, phenology_factor=36, leaf_area_index=37, canopy_water_conductance=38,
    leaf_to_air_co2_ratio=39, net_assimilation=40, leaf_respiration=41, gross_primary_production=42,
    autotrophic_respiration=43, net_primary_production=44, CO2=45, soil_moisture_limiting_factor=46,
    daily_leaf_respiration=47, plant_available_water=49, root_fraction=50, canopy_water=51, tend_canopy_water=52,
    canopy_water_interception=53, canopy_water_removal=54, saturation_canopy_water=55, rainfall_ground=56,
    evaporation_canopy=57, transpiration=58, SAI=59,
)
BC_VARS = dict(internal_energy=0, saturation_water_ice=1, temperature=2, liquid_water_fraction=3, pressure_head=4)
BC_KINDS = dict(noflux=0, value=1, flux=2, gradient=3)
TIME_INDEXING = dict(linear=0, clamp=1, cyclical=2, raster=3)


class ParamsD(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "rho_w rho_i rho_a c_a Lsl Llg Lsg g Tref sigma kappa_vk eps_mw R_a "
        "k_water k_ice k_air k_mineral k_organic c_water c_ice c_air c_mineral c_organic "
        "por_mineral por_organic rho_soc rho_org "
        "K_sat theta_res bc_psi_s bc_lambda vg_alpha vg_n impedance vwc_forcing "
        "albedo emissivity kappa_s C_h min_windspeed tau_r beta_evap field_capacity").split()] + [
        (n, C.c_int32) for n in "flow swrc unsat_k seb halo_policy prescribed_albedo evap_resistance reserved".split()]


def default_params(**overrides):
    """Reference defaults."""
    d = dict(
        rho_w=1000.0, rho_i=916.2, rho_a=1.293, c_a=1005.7, Lsl=3.34e5, Llg=2.257e6, Lsg=2.834e6, g=9.80665,
        Tref=273.15, sigma=5.6704e-8, kappa_vk=0.4, eps_mw=0.622, R_a=287.058,
        k_water=0.57, k_ice=2.2, k_air=0.025, k_mineral=3.8, k_organic=0.25,
        c_water=4.2e6, c_ice=1.9e6, c_air=0.00125e6, c_mineral=2.0e6, c_organic=2.5e6,
        por_mineral=0.49, por_organic=0.9, rho_soc=0.0, rho_org=1300.0,
        K_sat=1.0e-5, theta_res=0.0, bc_psi_s=0.01, bc_lambda=0.2, vg_alpha=1.0, vg_n=2.0, impedance=7.0,
        vwc_forcing=0.0,
        albedo=0.3, emissivity=0.97, kappa_s=2.0, C_h=1.2e-3, min_windspeed=0.01, tau_r=3600.0, beta_evap=1.0, field_capacity=0.25,
        flow=0, swrc=0, unsat_k=0, seb=0, halo_policy=0, prescribed_albedo=0, evap_resistance=0, reserved=0,
    )
    for k, v in overrides.items():
        if k not in d:
            raise KeyError(k)
        d[k] = v
    return ParamsD(**d)


def build(force=False):
    """Compile the restatement (gcc only; no reference sources are involved)."""
    so = os.path.join(_HERE, "build", "libterrarium_oracle.so")
    src = [os.path.join(_HERE, f) for f in ("oracle_capi.cpp", "terrarium_oracle.hpp", "Makefile")]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return so


_libs = {}


def _lib(omp=False):
    if omp not in _libs:
        build()
        name = "libterrarium_oracle_omp.so" if omp else "libterrarium_oracle.so"
        lib = C.CDLL(os.path.join(_HERE, "build", name))
        lib.trm_oracle_set_threads.restype = C.c_int
        lib.trm_oracle_set_threads.argtypes = [C.c_int]
        lib.trm_oracle_create.restype = C.c_void_p
        lib.trm_oracle_create.argtypes = [C.c_int, C.c_long, C.c_int, C.c_void_p, C.c_double, C.POINTER(ParamsD)]
        lib.trm_oracle_destroy.argtypes = [C.c_void_p]
        lib.trm_oracle_field_rows.restype = C.c_long
        lib.trm_oracle_field_rows.argtypes = [C.c_void_p, C.c_int]
        lib.trm_oracle_set_field.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.trm_oracle_get_field.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.trm_oracle_get_halo.restype = C.c_double
        lib.trm_oracle_get_halo.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_long]
        lib.trm_oracle_set_bc.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_double]
        lib.trm_oracle_set_series.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_long, C.c_void_p, C.c_void_p, C.c_int]
        lib.trm_oracle_clear_series.argtypes = [C.c_void_p]
        lib.trm_oracle_update_inputs.argtypes = [C.c_void_p]
        lib.trm_oracle_time_indices.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_long), C.POINTER(C.c_long)]
        lib.trm_oracle_set_land_model.argtypes = [C.c_void_p, C.c_int]
        lib.trm_oracle_set_et_coupled.argtypes = [C.c_void_p, C.c_int]
        lib.trm_oracle_grid.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        for f in ("fill_halo_regions", "initialize", "reset_tendencies", "compute_auxiliary", "compute_tendencies",
                  "closure", "invclosure", "adjust_saturation_profile", "compute_water_table", "compute_evaporation",
                  "compute_runoff", "compute_hydraulics", "compute_surface_energy_fluxes", "update_skin_temperature",
                  "seb_fluxes_only"):
            getattr(lib, "trm_oracle_" + f).argtypes = [C.c_void_p]
        lib.trm_oracle_update_state.argtypes = [C.c_void_p, C.c_int]
        lib.trm_oracle_explicit_step.argtypes = [C.c_void_p, C.c_double]
        lib.trm_oracle_timestep.argtypes = [C.c_void_p, C.c_double, C.c_int]
        lib.trm_oracle_timestep_heun.argtypes = [C.c_void_p, C.c_double, C.c_int]
        lib.trm_oracle_clone.restype = C.c_void_p
        lib.trm_oracle_clone.argtypes = [C.c_void_p]
        lib.trm_oracle_tick.argtypes = [C.c_void_p, C.c_double]
        lib.trm_oracle_average_tendencies.argtypes = [C.c_void_p, C.c_void_p]
        lib.trm_oracle_run.argtypes = [C.c_void_p, C.c_double, C.c_long]
        lib.trm_oracle_steps.argtypes = [C.c_void_p, C.c_double, C.c_long]
        lib.trm_oracle_steps_blocked.argtypes = [C.c_void_p, C.c_double, C.c_long, C.c_long]
        lib.trm_oracle_clock.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]
        lib.trm_oracle_set_clock.argtypes = [C.c_void_p, C.c_double, C.c_longlong]
        lib.trm_oracle_status.restype = C.c_uint
        lib.trm_oracle_status.argtypes = [C.c_void_p]
        P = C.POINTER(ParamsD)
        D = C.c_double
        for name, args in dict(
            porosity=[P], thermal_conductivity=[P, D, D, D, D], heat_capacity=[P, D, D, D, D],
            hydraulic_conductivity=[P, D, D, D, D], swrc_theta=[P, D, D], swrc_psi=[P, D, D],
            energy_to_temperature=[D, D, D], liquid_water_fraction=[D, D], stefan_boltzmann=[P, D, D],
            net_radiation=[D, D, D, D], longwave_up=[P, D, D, D], saturation_vapor_pressure=[D], pow=[D, D],
            safediv=[D, D], expmodel=[C.c_int, D, D, D, C.c_int],
            surface_drainage=[P, D], infiltration=[D, D, D], surface_runoff=[D, D, D],
        ).items():
            fn = getattr(lib, "trm_oracle_" + name)
            fn.restype = D
            fn.argtypes = args
        lib.trm_oracle_volumetric_fractions.argtypes = [D, D, D, D, C.c_void_p]
        _libs[omp] = lib
    return _libs[omp]


def set_threads(n, omp=True):
    """OpenMP thread count of the timing leg; returns the count in effect."""
    return _lib(omp).trm_oracle_set_threads(int(n))


def time_indices(times, t, time_indexing="linear"):
    """(f, n1, n2) of Oceananigans' FieldTimeSeries time interpolation as restated in the oracle (0-based nodes)."""
    tt = np.ascontiguousarray(times, dtype=np.float64)
    f, n1, n2 = C.c_double(), C.c_long(), C.c_long()
    _lib().trm_oracle_time_indices(tt.ctypes.data, tt.size, TIME_INDEXING[time_indexing], float(t), C.byref(f), C.byref(n1), C.byref(n2))
    return f.value, n1.value, n2.value


def scalar(name, *args):
    """Call one of the scalar physics entry points (unit known-answer tests)."""
    lib = _lib()
    args = [C.byref(a) if isinstance(a, ParamsD) else a for a in args]
    return getattr(lib, "trm_oracle_" + name)(*args)


def volumetric_fractions(por, sat, liq, org=0.0):
    out = np.zeros(5)
    _lib().trm_oracle_volumetric_fractions(por, sat, liq, org, out.ctypes.data)
    return dict(zip(("water", "ice", "air", "mineral", "organic"), out))


class Oracle:
    """Reference-order CPU driver.  Arrays cross as [rows][Nh], k = 0 bottom."""

    def __init__(self, num_columns, thickness, params=None, dtype=np.float64, dx=0.0, omp=False):
        self.lib = _lib(omp)
        self.dtype = np.dtype(dtype)
        self.Nh = int(num_columns)
        self.thickness = np.ascontiguousarray(thickness, dtype=np.float64)
        self.Nz = int(self.thickness.size)
        self.params = params if params is not None else default_params()
        prec = 0 if self.dtype == np.float64 else 1
        self.h = self.lib.trm_oracle_create(prec, self.Nh, self.Nz, self.thickness.ctypes.data, float(dx),
                                            C.byref(self.params))

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.trm_oracle_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # -- fields ---------------------------------------------------------------
    def rows(self, name):
        return self.lib.trm_oracle_field_rows(self.h, FIELDS[name])

    def set(self, name, value):
        rows = self.rows(name)
        a = np.empty((rows, self.Nh), dtype=self.dtype)
        a[...] = np.asarray(value, dtype=self.dtype).reshape((-1, 1)) if np.ndim(value) == 1 and np.size(value) == rows and rows != self.Nh else value
        a = np.ascontiguousarray(a)
        rc = self.lib.trm_oracle_set_field(self.h, FIELDS[name], a.ctypes.data)
        assert rc == 0, name

    def get(self, name):
        rows = self.rows(name)
        a = np.empty((rows, self.Nh), dtype=self.dtype)
        rc = self.lib.trm_oracle_get_field(self.h, FIELDS[name], a.ctypes.data)
        assert rc == 0, name
        return a[0] if rows == 1 else a

    def halo(self, name, top, i=0):
        return self.lib.trm_oracle_get_halo(self.h, FIELDS[name], int(top), i)

    def set_bc(self, var, side, kind, value=0.0):
        top = {"top": 1, "bottom": 0}[side]
        if np.ndim(value) == 0:
            ptr, scalar_v = None, float(value)
        else:
            arr = np.ascontiguousarray(value, dtype=self.dtype)
            assert arr.shape == (self.Nh,)
            ptr, scalar_v = arr.ctypes.data, 0.0
        rc = self.lib.trm_oracle_set_bc(self.h, BC_VARS[var], top, BC_KINDS[kind], ptr, scalar_v)
        assert rc == 0

    # -- time series input sources (input_sources.jl:142-171) -----------------
    def _series(self, times, values):
        t = np.ascontiguousarray(times, dtype=np.float64)
        v = np.ascontiguousarray(values, dtype=self.dtype)
        if v.ndim == 1:
            v = np.ascontiguousarray(np.broadcast_to(v[:, None], (v.size, self.Nh)))
        assert v.shape == (t.size, self.Nh)
        return t, v

    def set_forcing_series(self, name, times, values, time_indexing="linear"):
        t, v = self._series(times, values)
        rc = self.lib.trm_oracle_set_series(self.h, 0, FIELDS[name], 0, 0, 0, t.size, t.ctypes.data, v.ctypes.data, TIME_INDEXING[time_indexing])
        assert rc == 0

    def set_bc_series(self, var, side, kind, times, values, time_indexing="linear"):
        t, v = self._series(times, values)
        top = {"top": 1, "bottom": 0}[side]
        rc = self.lib.trm_oracle_set_series(self.h, 1, 0, BC_VARS[var], top, BC_KINDS[kind], t.size, t.ctypes.data, v.ctypes.data, TIME_INDEXING[time_indexing])
        assert rc == 0

    def clear_series(self): self.lib.trm_oracle_clear_series(self.h)
    def update_inputs(self): self.lib.trm_oracle_update_inputs(self.h)

    def enable_vegetation(self, veg_params=None):
        """LandModel(grid; soil, vegetation = VegetationCarbon): couple the vegetation and canopy processes."""
        self.veg_params = veg_params if veg_params is not None else default_vegetation_params()
        self.lib.trm_oracle_enable_vegetation.argtypes = [C.c_void_p, C.POINTER(VegParamsD)]
        self.lib.trm_oracle_enable_vegetation(self.h, C.byref(self.veg_params))

    def set_land_model(self, on=True):
        self.lib.trm_oracle_set_land_model(self.h, int(on))

    def grid(self):
        zF, dzf = np.zeros(self.Nz + 1), np.zeros(self.Nz + 1)
        zC, dzc = np.zeros(self.Nz), np.zeros(self.Nz)
        self.lib.trm_oracle_grid(self.h, zF.ctypes.data, zC.ctypes.data, dzc.ctypes.data, dzf.ctypes.data)
        return dict(zF=zF, zC=zC, dzc=dzc, dzf=dzf)

    # -- reference interface --------------------------------------------------
    def fill_halo_regions(self): self.lib.trm_oracle_fill_halo_regions(self.h)
    def initialize(self): self.lib.trm_oracle_initialize(self.h)
    def update_state(self, compute_tendencies=True): self.lib.trm_oracle_update_state(self.h, int(compute_tendencies))
    def reset_tendencies(self): self.lib.trm_oracle_reset_tendencies(self.h)
    def compute_auxiliary(self): self.lib.trm_oracle_compute_auxiliary(self.h)
    def compute_tendencies(self): self.lib.trm_oracle_compute_tendencies(self.h)
    def explicit_step(self, dt): self.lib.trm_oracle_explicit_step(self.h, float(dt))
    def closure(self): self.lib.trm_oracle_closure(self.h)
    def invclosure(self): self.lib.trm_oracle_invclosure(self.h)
    def adjust_saturation_profile(self): self.lib.trm_oracle_adjust_saturation_profile(self.h)
    def compute_water_table(self): self.lib.trm_oracle_compute_water_table(self.h)
    # surface-process passes of compute_auxiliary!, one at a time
    def set_et_coupled(self, on=True): self.lib.trm_oracle_set_et_coupled(self.h, int(on))
    def compute_hydraulics(self): self.lib.trm_oracle_compute_hydraulics(self.h)
    def compute_evaporation(self): self.lib.trm_oracle_compute_evaporation(self.h)
    def compute_runoff(self): self.lib.trm_oracle_compute_runoff(self.h)
    def compute_surface_energy_fluxes(self): self.lib.trm_oracle_compute_surface_energy_fluxes(self.h)
    def update_skin_temperature(self): self.lib.trm_oracle_update_skin_temperature(self.h)
    def seb_fluxes_only(self): self.lib.trm_oracle_seb_fluxes_only(self.h)
    def timestep(self, dt, finalize=True): self.lib.trm_oracle_timestep(self.h, float(dt), int(finalize))
    def timestep_heun(self, dt, finalize=True): self.lib.trm_oracle_timestep_heun(self.h, float(dt), int(finalize))

    # Heun stepped by hand (heun.jl:37-71), for tests that evaluate a state-dependent function at the stage between the halves
    def clone(self):
        """`deepcopy(state)` (heun.jl:24): an independent oracle with the same fields, clock, boundary values and parameters"""
        c = object.__new__(Oracle)
        c.__dict__.update(lib=self.lib, dtype=self.dtype, Nh=self.Nh, thickness=self.thickness, Nz=self.Nz, params=self.params)
        c.h = self.lib.trm_oracle_clone(self.h)
        return c

    def tick(self, dt): self.lib.trm_oracle_tick(self.h, float(dt))
    def average_tendencies(self, stage): self.lib.trm_oracle_average_tendencies(self.h, stage.h)

    def timestep_heun_by_hand(self, dt, finalize=True, at_state=None, at_stage=None, at_stage_tendencies=None):
        """timestep!(integrator, ::Heun) (heun.jl:37-71) with callbacks: `at_state(oracle)` before update_state!(state);
        `at_stage(stage)` before update_state!(stage) (its clock has ticked) -- where fill_halo_regions! evaluates a boundary-value
        function, the stage's auxiliary fields still the state's copies (copyto!, heun.jl:45); `at_stage_tendencies(stage)` between
        compute_auxiliary!(stage) and compute_tendencies!(stage) (state_variables.jl:72-80) -- where the tendency kernel evaluates a
        forcing function, the stage's auxiliary fields its own."""
        if at_state:
            at_state(self)
        self.update_state(True)
        stage = self.clone()
        stage.explicit_step(dt)
        stage.closure()
        stage.tick(dt)
        if at_stage:
            at_stage(stage)
        if at_stage_tendencies:
            stage.update_state(False)           # reset tendencies, compute_auxiliary!
            at_stage_tendencies(stage)
            stage.compute_tendencies()
        else:
            stage.update_state(True)
        self.average_tendencies(stage)
        self.explicit_step(dt)
        self.closure()
        self.tick(dt)
        if finalize:
            self.compute_auxiliary()
    def run(self, dt, steps): self.lib.trm_oracle_run(self.h, float(dt), int(steps))
    def steps(self, dt, steps): self.lib.trm_oracle_steps(self.h, float(dt), int(steps))
    def steps_blocked(self, dt, steps, block=64): self.lib.trm_oracle_steps_blocked(self.h, float(dt), int(steps), int(block))

    def clock(self):
        t, it = C.c_double(), C.c_longlong()
        self.lib.trm_oracle_clock(self.h, C.byref(t), C.byref(it))
        return t.value, it.value

    def set_clock(self, time, iteration=0):
        self.lib.trm_oracle_set_clock(self.h, float(time), int(iteration))

    def status(self):
        return self.lib.trm_oracle_status(self.h)


# ---- vegetation (oracle/vegetation_oracle.hpp) -------------------------------------------------------------------------------
VEG_PARAM_NAMES = ("tau25 Kc25 Ko25 q10_tau q10_Kc q10_Ko alpha_leaf alpha_a alpha_C3 cq k_ext T_CO2_high T_CO2_low T_photos_high "
                   "T_photos_low theta_r g1 g_min cn_sapwood cn_root aws SLA awl LAI_min LAI_max gamma_L gamma_R gamma_S nu_seed "
                   "gamma_v_min root_a root_b wilting_point field_capacity C_mass alpha_int canopy_k_ext w_can_max tau_w C_can").split()


class VegParamsD(C.Structure):
    _fields_ = [(n, C.c_double) for n in VEG_PARAM_NAMES]


def default_vegetation_params(**overrides):
    """Reference defaults (needleleaf-tree PFT): photosynthesis.jl:17-68, stomatal_conductance.jl:16-24,
    autotrophic_respiration.jl:14-23, carbon_dynamics.jl:18-43, vegetation_dynamics.jl:15-22, root_distribution.jl:23-29,
    soil_hydraulic_properties.jl:74-80, physical_constants.jl:50."""
    d = dict(tau25=2600.0, Kc25=30.0, Ko25=3.0e4, q10_tau=0.57, q10_Kc=2.1, q10_Ko=1.2, alpha_leaf=0.17, alpha_a=0.5, alpha_C3=0.08,
             cq=4.6e-6, k_ext=0.5, T_CO2_high=42.0, T_CO2_low=-4.0, T_photos_high=30.0, T_photos_low=15.0, theta_r=0.7,
             g1=2.3, g_min=0.5, cn_sapwood=330.0, cn_root=29.0, aws=10.0,
             SLA=10.0, awl=2.0, LAI_min=1.0, LAI_max=6.0, gamma_L=0.3, gamma_R=0.3, gamma_S=0.05,
             nu_seed=0.001, gamma_v_min=0.002, root_a=7.0, root_b=2.0, wilting_point=0.05, field_capacity=0.25, C_mass=12.0,
             alpha_int=0.2, canopy_k_ext=0.5, w_can_max=2.0e-4, tau_w=86400.0, C_can=0.006)
    for k, v in overrides.items():
        if k not in d:
            raise KeyError(k)
        d[k] = v
    return VegParamsD(**d)


VEG_FIELDS = dict(carbon_vegetation=0, vegetation_area_fraction=1, tend_carbon_vegetation=2, tend_vegetation_area_fraction=3,
                  balanced_leaf_area_index=4, phenology_factor=5, leaf_area_index=6, canopy_water_conductance=7,
                  leaf_to_air_co2_ratio=8, net_assimilation=9, leaf_respiration=10, gross_primary_production=11,
                  autotrophic_respiration=12, net_primary_production=13, air_temperature=14, air_pressure=15,
                  specific_humidity=16, surface_shortwave_down=17, CO2=18, soil_moisture_limiting_factor=19,
                  daily_leaf_respiration=20, ground_temperature=21)
VEG_SCALARS = dict(lambda_NPP=0, LAI_b=1, Lambda_loc=2, C_veg_tend=3, f_deciduous=4, phenology_factor=5, LAI=6, gamma_v=7, nu_star=8,
                   nu_tendency=9, gw_can=10, lambda_c=11, tau=12, Kc=13, Ko=14, Gamma_star=15, PAR=16, APAR=17, pres_i=18,
                   temperature_stress=19, c_1=20, c_2=21, Vc_max=22, JE=23, JC=24, Rd=25, Ag=26, resp_Rd=27, resp_An=28,
                   f_temp_air=29, f_temp_soil=30, resp10=31, Rm=32, Rg=33, Ra=34, NPP=35, root_density=36, plant_available_water=37,
                   canopy_interception=38, canopy_saturation_fraction=39, canopy_water_removal=40, w_can_tendency=41,
                   precip_ground=42, transpiration=43, evaporation_ground=44, evaporation_canopy=45, canopy_ground_resistance=46)


def _veg_lib():
    lib = _lib()
    if not getattr(lib, "_veg_ready", False):
        lib.trm_oracle_veg_create.restype = C.c_void_p
        lib.trm_oracle_veg_create.argtypes = [C.c_int, C.c_long, C.POINTER(VegParamsD), C.POINTER(ParamsD)]
        lib.trm_oracle_veg_destroy.argtypes = [C.c_void_p]
        lib.trm_oracle_veg_set.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.trm_oracle_veg_get.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.trm_oracle_veg_compute_auxiliary.argtypes = [C.c_void_p]
        lib.trm_oracle_veg_compute_tendencies.argtypes = [C.c_void_p]
        lib.trm_oracle_veg_timestep.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_int]
        lib.trm_oracle_veg_time.restype = C.c_double
        lib.trm_oracle_veg_time.argtypes = [C.c_void_p]
        lib.trm_oracle_veg_scalar.restype = C.c_double
        lib.trm_oracle_veg_scalar.argtypes = [C.POINTER(VegParamsD), C.c_int, C.POINTER(C.c_double)]
        lib._veg_ready = True
    return lib


def veg_scalar(name, *args, params=None):
    """One of the scalar vegetation formulas (unit known-answer tests of test/vegetation/*.jl)."""
    p = params if params is not None else default_vegetation_params()
    x = (C.c_double * 8)(*[float(a) for a in args])
    return _veg_lib().trm_oracle_veg_scalar(C.byref(p), VEG_SCALARS[name], x)


class VegetationOracle:
    """The standalone VegetationModel (src/models/vegetation/vegetation_model.jl) on `num_columns` points."""

    def __init__(self, num_columns, veg_params=None, params=None, dtype=np.float64):
        self.lib = _veg_lib()
        self.dtype = np.dtype(dtype)
        self.Nh = int(num_columns)
        self.veg_params = veg_params if veg_params is not None else default_vegetation_params()
        self.params = params if params is not None else default_params()
        self.h = self.lib.trm_oracle_veg_create(0 if self.dtype == np.float64 else 1, self.Nh, C.byref(self.veg_params), C.byref(self.params))

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.trm_oracle_veg_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def set(self, name, value):
        a = np.empty(self.Nh, dtype=self.dtype)
        a[...] = value
        assert self.lib.trm_oracle_veg_set(self.h, VEG_FIELDS[name], a.ctypes.data) == 0, name

    def get(self, name):
        a = np.empty(self.Nh, dtype=self.dtype)
        assert self.lib.trm_oracle_veg_get(self.h, VEG_FIELDS[name], a.ctypes.data) == 0, name
        return a

    def compute_auxiliary(self): self.lib.trm_oracle_veg_compute_auxiliary(self.h)
    def compute_tendencies(self): self.lib.trm_oracle_veg_compute_tendencies(self.h)
    def timestep(self, dt, finalize=True, heun=False): self.lib.trm_oracle_veg_timestep(self.h, float(dt), int(finalize), int(heun))
    def time(self): return self.lib.trm_oracle_veg_time(self.h)
