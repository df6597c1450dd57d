// oracle/oracle_capi.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
// extern "C" surface of the CPU restatement (terrarium_oracle.hpp) for the
// ctypes harness in oracle/oracle.py.  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load the resulting library.
#include "terrarium_oracle.hpp"
#ifdef _OPENMP
#include <omp.h>
#endif

using namespace trm_oracle;

struct OracleHandle {
    int precision;  // 0 = f64, 1 = f32
    Oracle<double>* d;
    Oracle<float>* f;
};

#define DISPATCH(h, expr)            \
    do {                             \
        if ((h)->precision == 0) {   \
            auto* o = (h)->d;        \
            expr;                    \
        } else {                     \
            auto* o = (h)->f;        \
            expr;                    \
        }                            \
    } while (0)

extern "C" {

// number of OpenMP threads of the timing leg (no-op in the serial build); returns the count in effect
int trm_oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

OracleHandle* trm_oracle_create(int precision, long nh, int nz, const double* thickness, double dx, const ParamsD* params) {
    OracleHandle* h = new OracleHandle{precision, nullptr, nullptr};
    if (precision == 0) h->d = new Oracle<double>(nh, nz, thickness, dx, *params);
    else h->f = new Oracle<float>(nh, nz, thickness, dx, *params);
    return h;
}
void trm_oracle_destroy(OracleHandle* h) {
    if (!h) return;
    delete h->d;
    delete h->f;
    delete h;
}
// LandModel(grid; soil, vegetation = VegetationCarbon): couple the vegetation and canopy processes (land_model.jl:24-34)
void trm_oracle_enable_vegetation(OracleHandle* h, const VegParamsD* vp) { DISPATCH(h, o->enable_vegetation(*vp)); }
long trm_oracle_field_rows(OracleHandle* h, int id) {
    long r = 0;
    DISPATCH(h, r = o->field_rows(id));
    return r;
}
int trm_oracle_set_field(OracleHandle* h, int id, const void* src) {
    if (h->precision == 0) return h->d->set_field(id, (const double*)src);
    return h->f->set_field(id, (const float*)src);
}
int trm_oracle_get_field(OracleHandle* h, int id, void* dst) {
    if (h->precision == 0) return h->d->get_field(id, (double*)dst);
    return h->f->get_field(id, (float*)dst);
}
double trm_oracle_get_halo(OracleHandle* h, int id, int top, long i) {
    double r = 0;
    DISPATCH(h, r = (double)o->get_halo(id, top, i));
    return r;
}
int trm_oracle_set_bc(OracleHandle* h, int var, int top, int kind, const void* values, double scalar) {
    if (h->precision == 0) return h->d->set_bc(var, top, kind, (const double*)values, scalar);
    return h->f->set_bc(var, top, kind, (const float*)values, (float)scalar);
}
int trm_oracle_set_series(OracleHandle* h, int is_bc, int field, int var, int top, int kind, long nt, const double* times, const void* values, int indexing) {
    if (h->precision == 0) return h->d->set_series(is_bc != 0, field, var, top, kind, nt, times, (const double*)values, indexing);
    return h->f->set_series(is_bc != 0, field, var, top, kind, nt, times, (const float*)values, indexing);
}
void trm_oracle_clear_series(OracleHandle* h) { DISPATCH(h, o->series.clear()); }
void trm_oracle_update_inputs(OracleHandle* h) { DISPATCH(h, o->update_inputs()); }
void trm_oracle_time_indices(const double* times, long nt, int indexing, double t, double* f, long* n1, long* n2) {
    std::vector<double> tv(times, times + nt);
    trm_oracle::Oracle<double>::interpolating_time_indices(tv, indexing, t, *f, *n1, *n2);
}
void trm_oracle_set_land_model(OracleHandle* h, int on) { DISPATCH(h, o->land_model = on != 0); }
void trm_oracle_grid(OracleHandle* h, double* zF /*Nz+1*/, double* zC /*Nz*/, double* dzc /*Nz*/, double* dzf /*Nz+1*/) {
    DISPATCH(h, {
        for (int k = 1; k <= o->Nz + 1; ++k) { zF[k - 1] = o->g.zF[k]; dzf[k - 1] = o->g.dzf[k]; }
        for (int k = 1; k <= o->Nz; ++k) { zC[k - 1] = o->g.zC[k]; dzc[k - 1] = o->g.dzc[k]; }
    });
}
void trm_oracle_fill_halo_regions(OracleHandle* h) { DISPATCH(h, o->fill_halo_regions()); }
void trm_oracle_initialize(OracleHandle* h) { DISPATCH(h, o->initialize_processes()); }
void trm_oracle_update_state(OracleHandle* h, int tendencies) { DISPATCH(h, o->update_state(tendencies != 0)); }
void trm_oracle_reset_tendencies(OracleHandle* h) { DISPATCH(h, o->reset_tendencies()); }
void trm_oracle_compute_auxiliary(OracleHandle* h) { DISPATCH(h, o->compute_auxiliary()); }
void trm_oracle_compute_tendencies(OracleHandle* h) { DISPATCH(h, o->compute_tendencies()); }
void trm_oracle_explicit_step(OracleHandle* h, double dt) {
    if (h->precision == 0) h->d->explicit_step(dt);
    else h->f->explicit_step((float)dt);
}
void trm_oracle_closure(OracleHandle* h) { DISPATCH(h, o->closure()); }
void trm_oracle_invclosure(OracleHandle* h) { DISPATCH(h, o->invclosure()); }
void trm_oracle_adjust_saturation_profile(OracleHandle* h) { DISPATCH(h, o->adjust_saturation_profile()); }
void trm_oracle_compute_water_table(OracleHandle* h) { DISPATCH(h, o->compute_water_table()); }
void trm_oracle_timestep(OracleHandle* h, double dt, int finalize) { DISPATCH(h, o->timestep_euler(dt, finalize != 0)); }
void trm_oracle_timestep_heun(OracleHandle* h, double dt, int finalize) { DISPATCH(h, o->timestep_heun(dt, finalize != 0)); }
// Heun stepped BY HAND from the test harness (heun.jl:37-71), so that a test can evaluate a state-dependent forcing / boundary
// value at the stage between the two halves: `stage = deepcopy(state)` (heun.jl:24, 45), tick!(clock), average_tendencies!
OracleHandle* trm_oracle_clone(OracleHandle* h) {
    OracleHandle* c = new OracleHandle{h->precision, nullptr, nullptr};
    if (h->precision == 0) c->d = new Oracle<double>(*h->d);
    else c->f = new Oracle<float>(*h->f);
    return c;
}
void trm_oracle_tick(OracleHandle* h, double dt) { DISPATCH(h, o->tick(dt)); }
void trm_oracle_average_tendencies(OracleHandle* h, OracleHandle* stage) {
    if (h->precision == 0) h->d->average_tendencies(*stage->d);
    else h->f->average_tendencies(*stage->f);
}
void trm_oracle_run(OracleHandle* h, double dt, long steps) { DISPATCH(h, o->run(dt, steps)); }
// `steps` Euler steps without the trailing compute_auxiliary (timing leg)
void trm_oracle_steps(OracleHandle* h, double dt, long steps) {
    DISPATCH(h, { for (long s = 0; s < steps; ++s) o->timestep_euler(dt, false); });
}
// the same, cache-blocked over `block` columns at a time (BASELINE.md 4.2 "fused driver")
void trm_oracle_steps_blocked(OracleHandle* h, double dt, long steps, long block) { DISPATCH(h, o->steps_blocked(dt, steps, block)); }
void trm_oracle_clock(OracleHandle* h, double* time, long long* iteration) {
    DISPATCH(h, { *time = o->time; *iteration = o->iteration; });
}
void trm_oracle_set_clock(OracleHandle* h, double time, long long iteration) {
    DISPATCH(h, { o->time = time; o->iteration = iteration; });
}
unsigned trm_oracle_status(OracleHandle* h) {
    unsigned s = 0;
    DISPATCH(h, s = o->status);
    return s;
}

// ---- scalar entry points for the unit known-answer tests --------------------
double trm_oracle_porosity(const ParamsD* pd) { return porosity(Params<double>(*pd)); }
double trm_oracle_thermal_conductivity(const ParamsD* pd, double por, double sat, double liq, double org) {
    Params<double> p(*pd);
    return thermal_conductivity(p, volumetric_fractions(por, sat, liq, org));
}
double trm_oracle_heat_capacity(const ParamsD* pd, double por, double sat, double liq, double org) {
    Params<double> p(*pd);
    return heat_capacity(p, volumetric_fractions(por, sat, liq, org));
}
void trm_oracle_volumetric_fractions(double por, double sat, double liq, double org, double* out5) {
    Fractions<double> f = volumetric_fractions(por, sat, liq, org);
    out5[0] = f.water; out5[1] = f.ice; out5[2] = f.air; out5[3] = f.mineral; out5[4] = f.organic;
}
double trm_oracle_hydraulic_conductivity(const ParamsD* pd, double por, double sat, double liq, double org) {
    Params<double> p(*pd);
    return hydraulic_conductivity_cell(p, por, liq, volumetric_fractions(por, sat, liq, org));
}
double trm_oracle_swrc_theta(const ParamsD* pd, double psi, double theta_sat) { return swrc_theta(Params<double>(*pd), psi, theta_sat); }
double trm_oracle_swrc_psi(const ParamsD* pd, double theta, double theta_sat) { return swrc_psi(Params<double>(*pd), theta, theta_sat); }
double trm_oracle_energy_to_temperature(double U, double Ltheta, double C) { return energy_to_temperature(U, Ltheta, C); }
double trm_oracle_liquid_water_fraction(double U, double Ltheta) { return liquid_water_fraction(U, Ltheta); }
double trm_oracle_stefan_boltzmann(const ParamsD* pd, double T, double emis) { return stefan_boltzmann(Params<double>(*pd), T, emis); }
double trm_oracle_net_radiation(double sw_up, double sw_down, double lw_up, double lw_down) { return sw_up - sw_down + lw_up - lw_down; }
double trm_oracle_longwave_up(const ParamsD* pd, double lw_down, double Ts, double emis) {
    Params<double> p(*pd);
    return stefan_boltzmann(p, Ts + p.Tref, emis) + (1.0 - emis) * lw_down;
}
double trm_oracle_saturation_vapor_pressure(double T) { return saturation_vapor_pressure(T); }
double trm_oracle_pow(double x, double y) { return jl_pow(x, y); }
double trm_oracle_safediv(double x, double y) { return safediv(x, y); }

// Generic explicit integrators on the 0-D model du/dt = u + v used by the
// reference's time-stepper test (test/timestepping/heun.jl:6-49): restates
// explicit_step! (abstract_timestepper.jl:113-141) and heun.jl:27-71.
double trm_oracle_expmodel(int heun, double u, double v, double dt, int steps) {
    for (int s = 0; s < steps; ++s) {
        double G = u + v;  // compute_tendencies!
        if (!heun) {
            u = u + G * dt;
        } else {
            double us = u + G * dt;   // stage Euler step
            double Gs = us + v;       // tendencies at the stage
            double Ga = (G + Gs) / 2; // average_tendencies!
            u = u + Ga * dt;
        }
    }
    return u;
}

// direct_surface_runoff.jl scalar functions (test/surface_hydrology/surface_runoff_tests.jl)
double trm_oracle_surface_drainage(const ParamsD* pd, double S) { return compute_surface_drainage(Params<double>(*pd), S); }
double trm_oracle_infiltration(double influx, double sat_top, double max_infil) { return compute_infiltration(influx, sat_top, max_infil); }
double trm_oracle_surface_runoff(double rain, double drainage, double infil) { return compute_surface_runoff(rain, drainage, infil); }
// surface-process passes of compute_auxiliary!, one at a time (unit known-answer tests)
void trm_oracle_set_et_coupled(OracleHandle* h, int on) { DISPATCH(h, o->et_coupled = on != 0); }
void trm_oracle_compute_evaporation(OracleHandle* h) { DISPATCH(h, o->compute_evaporation()); }
void trm_oracle_compute_runoff(OracleHandle* h) { DISPATCH(h, o->compute_runoff()); }
void trm_oracle_compute_hydraulics(OracleHandle* h) { DISPATCH(h, o->compute_hydraulics()); }
void trm_oracle_compute_surface_energy_fluxes(OracleHandle* h) { DISPATCH(h, o->compute_surface_energy_fluxes()); }
// compute_surface_energy_fluxes!(out, i, j, ...) once per column, skin temperature as it is (surface_energy_balance.jl:119-144)
void trm_oracle_seb_fluxes_only(OracleHandle* h) { DISPATCH(h, { for (long i = 0; i < o->Nh; ++i) o->seb_fluxes(i); }); }
void trm_oracle_update_skin_temperature(OracleHandle* h) { DISPATCH(h, o->update_skin_temperature()); }

// ---- vegetation (oracle/vegetation_oracle.hpp) ---------------------------------------------------------------------
struct VegHandle {
    int precision;
    VegetationOracle<double>* d;
    VegetationOracle<float>* f;
};
VegHandle* trm_oracle_veg_create(int precision, long nh, const VegParamsD* vp, const ParamsD* cp) {
    VegHandle* h = new VegHandle{precision, nullptr, nullptr};
    if (precision == 0) h->d = new VegetationOracle<double>(nh, *vp, *cp);
    else h->f = new VegetationOracle<float>(nh, *vp, *cp);
    return h;
}
void trm_oracle_veg_destroy(VegHandle* h) { if (h) { delete h->d; delete h->f; delete h; } }
int trm_oracle_veg_set(VegHandle* h, int id, const void* src) {
    if (h->precision == 0) { auto* v = h->d->field(id); if (!v) return 1; std::memcpy(v->data(), src, sizeof(double) * h->d->Nh); }
    else { auto* v = h->f->field(id); if (!v) return 1; std::memcpy(v->data(), src, sizeof(float) * h->f->Nh); }
    return 0;
}
int trm_oracle_veg_get(VegHandle* h, int id, void* dst) {
    if (h->precision == 0) { auto* v = h->d->field(id); if (!v) return 1; std::memcpy(dst, v->data(), sizeof(double) * h->d->Nh); }
    else { auto* v = h->f->field(id); if (!v) return 1; std::memcpy(dst, v->data(), sizeof(float) * h->f->Nh); }
    return 0;
}
void trm_oracle_veg_compute_auxiliary(VegHandle* h) { DISPATCH(h, o->compute_auxiliary()); }
void trm_oracle_veg_compute_tendencies(VegHandle* h) { DISPATCH(h, o->compute_tendencies()); }
void trm_oracle_veg_timestep(VegHandle* h, double dt, int finalize, int heun) {
    DISPATCH(h, { if (heun) o->timestep_heun(dt, finalize != 0); else o->timestep_euler(dt, finalize != 0); });
}
double trm_oracle_veg_time(VegHandle* h) { double t = 0; DISPATCH(h, t = o->time); return t; }
// scalar formulas for the unit known-answer tests (test/vegetation/*.jl); `what` selects the function, x[] its arguments
double trm_oracle_veg_scalar(const VegParamsD* vp, int what, const double* x) {
    VegParams<double> p(*vp);
    double a = 0, b = 0, c3 = 0;
    switch (what) {
        case 0: return veg_lambda_NPP(p, x[0]);
        case 1: return veg_LAI_b(p, x[0]);
        case 2: return veg_Lambda_loc(p, x[0]);
        case 3: return veg_C_veg_tend(p, x[0], x[1]);
        case 4: return veg_f_deciduous<double>();
        case 5: return veg_phenology_factor<double>();
        case 6: return veg_LAI(x[0]);
        case 7: return veg_gamma_v(p);
        case 8: return veg_nu_star(p, x[0]);
        case 9: return veg_nu_tendency(p, x[0], x[1], x[2], x[3]);
        case 10: return veg_gw_can(p, x[0], x[1], x[2], x[3], x[4]);
        case 11: return veg_lambda_c(p, x[0]);
        case 12: veg_kinetic(p, x[0], a, b, c3); return a;
        case 13: veg_kinetic(p, x[0], a, b, c3); return b;
        case 14: veg_kinetic(p, x[0], a, b, c3); return c3;
        case 15: return veg_Gamma_star(x[0], x[1]);
        case 16: return veg_PAR(p, x[0]);
        case 17: return veg_APAR(p, x[0], x[1]);
        case 18: return veg_pres_i(x[0], x[1]);
        case 19: return veg_temperature_stress(p, x[0]);
        case 20: veg_assimilation_factors(p, x[0], x[1], x[2], x[3], x[4], x[5], a, b); return a;
        case 21: veg_assimilation_factors(p, x[0], x[1], x[2], x[3], x[4], x[5], a, b); return b;
        case 22: return veg_Vc_max(x[0], x[1], x[2], x[3], x[4], x[5], x[6]);
        case 23: veg_JE_JC(x[0], x[1], x[2], x[3], a, b); return a;
        case 24: veg_JE_JC(x[0], x[1], x[2], x[3], a, b); return b;
        case 25: return veg_Rd(p, x[0], x[1]);
        case 26: return veg_Ag(p, x[0], x[1], x[2], x[3], x[4]);
        case 27: veg_respiration_assimilation(p, x[0], x[1], x[2], x[3], x[4], x[5], x[6], a, b); return a;   // Rd
        case 28: veg_respiration_assimilation(p, x[0], x[1], x[2], x[3], x[4], x[5], x[6], a, b); return b;   // An
        case 29: veg_f_temp(x[0], x[1], a, b); return a;
        case 30: veg_f_temp(x[0], x[1], a, b); return b;
        case 31: return veg_resp10<double>();
        case 32: return veg_Rm(p, x[0], x[1], x[2], x[3], x[4]);
        case 33: return veg_Rg(x[0], x[1]);
        case 34: return veg_Ra(p, x[0], x[1], x[2], x[3], x[4], x[5]);
        case 35: return veg_NPP(x[0], x[1]);
        case 36: return veg_root_density(p, x[0]);
        case 37: return veg_plant_available_water(p, x[0]);
        case 38: return canopy_interception(p, x[0], x[1], x[2]);
        case 39: return canopy_saturation_fraction(p, x[0], x[1], x[2]);
        case 40: return canopy_water_removal(p, x[0]);
        case 41: return canopy_w_can_tendency(x[0], x[1], x[2]);
        case 42: return canopy_precip_ground(x[0], x[1], x[2]);
        case 43: return canopy_transpiration(x[0], x[1], x[2]);
        case 44: return canopy_evaporation_ground(x[0], x[1], x[2], x[3]);
        case 45: return canopy_evaporation_canopy(x[0], x[1], x[2]);
        case 46: return canopy_ground_resistance(p, x[0], x[1], x[2]);
        default: return std::nan("");
    }
}

}  // extern "C"
