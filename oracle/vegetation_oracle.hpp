// oracle/vegetation_oracle.hpp
//
// TEST INFRASTRUCTURE -- NOT PRODUCT CODE (see terrarium_oracle.hpp).
//
// CPU restatement of the reference's 0-D vegetation processes: `VegetationCarbon`
// (src/processes/vegetation/vegetation_carbon.jl:66-118) with LUEPhotosynthesis, MedlynStomatalConductance,
// PALADYNAutotrophicRespiration, PALADYNPhenology, PALADYNCarbonDynamics, PALADYNVegetationDynamics,
// StaticExponentialRootDistribution and FieldCapacityLimitedPAW, driven as the standalone `VegetationModel`
// (src/models/vegetation/vegetation_model.jl:34-49) by the explicit time steppers.  Written from the Julia sources as
// text; pinned by the reference's unit tests under test/vegetation/ (tests/test_oracle_vegetation.py).
// (included by terrarium_oracle.hpp just in front of its Oracle class: the Julia Base helpers, Params and compute_vpd come from there)
#pragma once
#include <algorithm>
#include <cmath>
#include <limits>
#include <vector>

namespace trm_oracle {

struct VegParamsD {   // plain doubles across the C API; field order = include/terrarium_hip.h trm_vegetation_params
    // LUEPhotosynthesis (photosynthesis.jl:17-68)
    double tau25, Kc25, Ko25, q10_tau, q10_Kc, q10_Ko, alpha_leaf, alpha_a, alpha_C3, cq, k_ext, T_CO2_high, T_CO2_low,
        T_photos_high, T_photos_low, theta_r;
    // MedlynStomatalConductance (stomatal_conductance.jl:16-24)
    double g1, g_min;
    // PALADYNAutotrophicRespiration (autotrophic_respiration.jl:14-23)
    double cn_sapwood, cn_root, aws;
    // PALADYNCarbonDynamics (carbon_dynamics.jl:18-43)
    double SLA, awl, LAI_min, LAI_max, gamma_L, gamma_R, gamma_S;
    // PALADYNVegetationDynamics (vegetation_dynamics.jl:15-22)
    double nu_seed, gamma_v_min;
    // StaticExponentialRootDistribution (root_distribution.jl:23-29)
    double root_a, root_b;
    // field capacity / wilting point of the soil's hydraulic properties (soil_hydraulic_properties.jl:93-97,143-155)
    double wilting_point, field_capacity;
    // PhysicalConstants.C_mass (physical_constants.jl:50)
    double C_mass;
    // PALADYNCanopyInterception (canopy_interception.jl:37-49) and PALADYNCanopyEvapotranspiration.C_can (canopy_evapotranspiration.jl:33-40)
    double alpha_int, canopy_k_ext, w_can_max, tau_w, C_can;
};

template <class NF> struct VegParams {
    NF tau25, Kc25, Ko25, q10_tau, q10_Kc, q10_Ko, alpha_leaf, alpha_a, alpha_C3, cq, k_ext, T_CO2_high, T_CO2_low, T_photos_high,
        T_photos_low, theta_r, g1, g_min, cn_sapwood, cn_root, aws, SLA, awl, LAI_min, LAI_max, gamma_L, gamma_R, gamma_S, nu_seed,
        gamma_v_min, root_a, root_b, wilting_point, field_capacity, C_mass, alpha_int, canopy_k_ext, w_can_max, tau_w, C_can;
    VegParams() {}
    explicit VegParams(const VegParamsD& d) {
        const double* s = &d.tau25;
        NF* t = &tau25;
        for (int n = 0; n < 40; ++n) t[n] = NF(s[n]);
    }
};

// ---- scalar formulas ----------------------------------------------------------------------------------------------------
// carbon_dynamics.jl:62-73 compute_λ_NPP
template <class NF> inline NF veg_lambda_NPP(const VegParams<NF>& p, NF LAI_b) {
    if (LAI_b < p.LAI_min) return NF(0);
    if (LAI_b <= p.LAI_max) return (LAI_b - p.LAI_min) / (p.LAI_max - p.LAI_min);
    return NF(1);
}
// carbon_dynamics.jl:84-87 compute_LAI_b
template <class NF> inline NF veg_LAI_b(const VegParams<NF>& p, NF C_veg) { return C_veg / ((NF(2) / p.SLA) + p.awl); }
// carbon_dynamics.jl:98-105 compute_Λ_loc
template <class NF> inline NF veg_Lambda_loc(const VegParams<NF>& p, NF LAI_b) {
    return (p.gamma_L / p.SLA + p.gamma_R / p.SLA + p.gamma_S * p.awl) * LAI_b;
}
// carbon_dynamics.jl:116-126 compute_C_veg_tend
template <class NF> inline NF veg_C_veg_tend(const VegParams<NF>& p, NF LAI_b, NF NPP) {
    NF lam = veg_lambda_NPP(p, LAI_b);
    NF Lloc = veg_Lambda_loc(p, LAI_b);
    return (NF(1) - lam) * NPP - Lloc;
}
// phenology.jl:32-63: evergreen placeholder (f_deciduous = 0, phen = 1)
template <class NF> inline NF veg_f_deciduous() { return NF(0); }
template <class NF> inline NF veg_phenology_factor() { return NF(1); }
template <class NF> inline NF veg_LAI(NF LAI_b) {
    NF f = veg_f_deciduous<NF>(), phen = veg_phenology_factor<NF>();
    return (f * phen + (NF(1) - f)) * LAI_b;
}
// vegetation_dynamics.jl:41-57
template <class NF> inline NF veg_gamma_v(const VegParams<NF>& p) { return p.gamma_v_min; }
template <class NF> inline NF veg_nu_star(const VegParams<NF>& p, NF nu) { return std::max(nu, p.nu_seed); }
// vegetation_dynamics.jl:68-88 compute_ν_tendency
template <class NF> inline NF veg_nu_tendency(const VegParams<NF>& p, NF LAI_b, NF C_veg, NF NPP, NF nu) {
    NF lam = veg_lambda_NPP(p, LAI_b);
    NF gv = veg_gamma_v(p);
    NF ns = veg_nu_star(p, nu);
    return (lam * NPP / C_veg) * ns * (NF(1) - nu) - gv * ns;
}
// stomatal_conductance.jl:45-65 compute_gw_can (its @assert preconditions are not restated)
template <class NF> inline NF veg_gw_can(const VegParams<NF>& p, NF vpd, NF An, NF co2, NF LAI, NF beta) {
    NF g_min = p.g_min / 1000;
    NF g0 = g_min * (1 - std::exp(-p.k_ext * LAI)) * beta;
    return g0 + NF(1.6) * (1 + p.g1 / std::sqrt(vpd)) * An / co2 * NF(1.0e6);
}
// stomatal_conductance.jl:78-81 compute_λc
template <class NF> inline NF veg_lambda_c(const VegParams<NF>& p, NF vpd) {
    return NF(1) - NF(1) / (NF(1) + p.g1 / std::sqrt(vpd * NF(1.0e-3)));
}
// photosynthesis.jl:93-98 compute_kinetic_parameters
template <class NF> inline void veg_kinetic(const VegParams<NF>& p, NF T_air, NF& tau, NF& Kc, NF& Ko) {
    NF e = (T_air - NF(25)) * NF(0.1);
    tau = p.tau25 * jl_pow(p.q10_tau, e);
    Kc = p.Kc25 * jl_pow(p.q10_Kc, e);
    Ko = p.Ko25 * jl_pow(p.q10_Ko, e);
}
template <class NF> inline NF veg_Gamma_star(NF tau, NF pres_O2) { return pres_O2 / (NF(2) * tau); }          // :111-114
template <class NF> inline NF veg_PAR(const VegParams<NF>& p, NF swdown) {                                       // :122-126
    return NF(0.5) * swdown * (NF(1) - p.alpha_leaf) * p.cq;
}
template <class NF> inline NF veg_APAR(const VegParams<NF>& p, NF swdown, NF LAI) {                              // :138-143
    NF PAR = veg_PAR(p, swdown);
    return p.alpha_a * PAR * (NF(1) - std::exp(-p.k_ext * LAI));
}
template <class NF> inline NF veg_pres_i(NF lambda_c, NF pres_a) { return lambda_c * pres_a; }                  // :155-158
// photosynthesis.jl:165-188 compute_temperature_stress
template <class NF> inline NF veg_temperature_stress(const VegParams<NF>& p, NF T_air) {
    NF k1 = NF(2) * std::log(NF(1) / NF(0.99) - NF(1)) / (p.T_CO2_low - p.T_photos_low);
    NF k2 = NF(0.5) * (p.T_CO2_low + p.T_photos_low);
    NF k3 = std::log(NF(0.99) / NF(0.01)) / (p.T_CO2_high - p.T_photos_high);
    if (p.T_CO2_low < T_air && T_air < p.T_CO2_high) {
        NF low = NF(1) / (NF(1) + std::exp(k1 * (k2 - T_air)));
        NF high = NF(1) - NF(0.01) * std::exp(k3 * (T_air - p.T_photos_high));
        return low * high;
    }
    return NF(0);
}
// photosynthesis.jl:206-217 compute_assimilation_factors
template <class NF> inline void veg_assimilation_factors(const VegParams<NF>& p, NF Gs, NF T_stress, NF Kc, NF Ko, NF pres_i, NF pres_O2, NF& c1, NF& c2) {
    c1 = p.alpha_C3 * T_stress * p.C_mass * (pres_i - Gs) / (pres_i + NF(2) * Gs);
    c2 = (pres_i - Gs) / (pres_i + Kc * (NF(1) + pres_O2 / Ko));
}
// photosynthesis.jl:230-234 compute_Vc_max
template <class NF> inline NF veg_Vc_max(NF c1, NF PAR, NF Kc, NF Ko, NF Gs, NF pres_i, NF pres_O2) {
    return c1 * PAR * (pres_i + Kc * (NF(1) + pres_O2 / Ko)) / (pres_i - Gs);
}
template <class NF> inline void veg_JE_JC(NF c1, NF c2, NF APAR, NF Vc_max, NF& JE, NF& JC) { JE = c1 * APAR; JC = c2 * Vc_max; }   // :245-250
template <class NF> inline NF veg_Rd(const VegParams<NF>& p, NF Vc_max, NF beta) { return p.alpha_C3 * Vc_max * beta; }               // :263-267
// photosynthesis.jl:278-283 compute_Ag
template <class NF> inline NF veg_Ag(const VegParams<NF>& p, NF c1, NF c2, NF APAR, NF Vc_max, NF beta) {
    NF JE, JC;
    veg_JE_JC(c1, c2, APAR, Vc_max, JE, JC);
    NF s = JE + JC;
    return (s - std::sqrt(s * s - NF(4) * p.theta_r * JE * JC)) / (NF(2) * p.theta_r) * beta;
}
// photosynthesis.jl:290-337 compute_respiration_assimilation -> (Rd, An)
template <class NF> inline void veg_respiration_assimilation(const VegParams<NF>& p, NF T_air, NF swdown, NF pres, NF co2, NF LAI, NF lambda_c, NF beta, NF& Rd, NF& An) {
    NF pres_O2 = NF(0.209) * pres;                 // physics_utils.jl:16-20
    NF pres_a = co2 * NF(1.0e-6) * pres;           // physics_utils.jl:27-30
    Rd = NF(0);
    An = NF(0);
    if (swdown > NF(0) && T_air > NF(-3)) {
        NF tau, Kc, Ko;
        veg_kinetic(p, T_air, tau, Kc, Ko);
        NF Gs = veg_Gamma_star(tau, pres_O2);
        if (LAI > NF(0)) {
            NF APAR = veg_APAR(p, swdown, LAI);
            NF pres_i = veg_pres_i(lambda_c, pres_a);
            NF T_stress = veg_temperature_stress(p, T_air);
            NF c1, c2;
            veg_assimilation_factors(p, Gs, T_stress, Kc, Ko, pres_i, pres_O2, c1, c2);
            NF Vc_max = veg_Vc_max(c1, APAR, Kc, Ko, Gs, pres_i, pres_O2);
            Rd = veg_Rd(p, Vc_max, beta);
            NF Ag = veg_Ag(p, c1, c2, APAR, Vc_max, beta);
            An = Ag - Rd;
        }
    }
}
template <class NF> inline NF veg_GPP(NF An) { return An * NF(1.0e-3); }                                       // :345-348
// autotrophic_respiration.jl:46-56 compute_f_temp
template <class NF> inline void veg_f_temp(NF T_air, NF T_soil, NF& f_air, NF& f_soil) {
    auto f = [](NF T) { return std::exp(NF(308.56) * (NF(1) / NF(56.02) - NF(1) / (NF(46.02) + T))); };
    f_soil = jl_boolmul(T_soil > 7, f(T_soil));
    f_air = f(T_air);
}
template <class NF> inline NF veg_resp10() { return NF(0.066); }                                                // :63-66
// autotrophic_respiration.jl:73-93 compute_Rm
template <class NF> inline NF veg_Rm(const VegParams<NF>& p, NF T_air, NF T_soil, NF Rd, NF phen, NF C_veg) {
    NF fa, fs;
    veg_f_temp(T_air, T_soil, fa, fs);
    NF resp10 = veg_resp10<NF>();
    NF R_leaf = Rd / NF(1000);
    NF R_stem = resp10 * fa * (p.awl * ((NF(2) / p.SLA) + p.awl)) / (C_veg * p.aws * p.cn_sapwood);
    NF R_root = resp10 * fs * phen * (NF(2) / p.SLA) / (p.SLA * C_veg * p.cn_root);
    return R_leaf + R_stem + R_root;
}
template <class NF> inline NF veg_Rg(NF GPP, NF Rm) { return NF(0.25) * (GPP - Rm); }                           // :100-103
template <class NF> inline NF veg_Ra(const VegParams<NF>& p, NF T_air, NF T_soil, NF Rd, NF phen, NF C_veg, NF GPP) {   // :110-115
    NF Rm = veg_Rm(p, T_air, T_soil, Rd, phen, C_veg);
    NF Rg = veg_Rg(GPP, Rm);
    return Rm + Rg;
}
template <class NF> inline NF veg_NPP(NF GPP, NF Ra) { return GPP - Ra; }                                       // :123-126
// root_distribution.jl:38-41 root_density
template <class NF> inline NF veg_root_density(const VegParams<NF>& p, NF z) {
    return NF(0.5) * (p.root_a * std::exp(p.root_a * z) + p.root_b * std::exp(p.root_b * z));
}
// plant_available_water.jl:77-94 compute_plant_available_water
template <class NF> inline NF veg_plant_available_water(const VegParams<NF>& p, NF theta_w) {
    return jl_max(jl_min(NF(1), (theta_w - p.wilting_point) / (p.field_capacity - p.wilting_point)), NF(0));
}

// ---- canopy hydrology (surface_hydrology/canopy_interception/canopy_interception.jl:64-118) --------------------------------
template <class NF> inline NF canopy_interception(const VegParams<NF>& p, NF precip, NF LAI, NF SAI) {            // :64-67
    return p.alpha_int * precip * (NF(1) - std::exp(-p.canopy_k_ext * (LAI + SAI)));
}
template <class NF> inline NF canopy_saturation_fraction(const VegParams<NF>& p, NF w_can, NF LAI, NF SAI) {      // :74-78
    NF w_max = p.w_can_max * (LAI + SAI);
    return w_max > 0 ? w_can / w_max : NF(0);
}
template <class NF> inline NF canopy_water_removal(const VegParams<NF>& p, NF w_can) { return jl_max(w_can, NF(0)) / p.tau_w; }   // :85-91
template <class NF> inline NF canopy_w_can_tendency(NF I_can, NF E_can, NF R_can) { return I_can - E_can - R_can; }              // :100-106
template <class NF> inline NF canopy_precip_ground(NF precip, NF I_can, NF R_can) { return precip - I_can + R_can; }             // :118-124
// canopy evapotranspiration (evapotranspiration/canopy_evapotranspiration.jl:52-82, 163-176)
template <class NF> inline NF canopy_transpiration(NF dq, NF ra, NF gw_can) {
    NF rs = 1 / jl_max(gw_can, std::sqrt(std::numeric_limits<NF>::epsilon()));
    return dq / (ra + rs);
}
template <class NF> inline NF canopy_evaporation_ground(NF dq, NF beta, NF ra, NF re) { return beta * dq / (ra + re); }
template <class NF> inline NF canopy_evaporation_canopy(NF dq, NF f_can, NF ra) { return f_can * dq / ra; }
template <class NF> inline NF canopy_ground_resistance(const VegParams<NF>& p, NF LAI, NF SAI, NF wind) {
    return (1 - std::exp(-LAI - SAI)) / (p.C_can * wind);
}

// ---- the standalone VegetationModel on Nh columns ------------------------------------------------------------------------
template <class NF> class VegetationOracle {
  public:
    long Nh;
    VegParams<NF> p;
    Params<NF> c;   // physical constants + PrescribedAtmosphere parameters (compute_vpd)
    // prognostic (+ tendencies), auxiliaries, inputs: one value per column
    FieldVec<NF> C_veg, nu, G_C_veg, G_nu, LAI_b, phen, LAI, gw_can, lambda_c, An, Rd, GPP, Ra, NPP;
    FieldVec<NF> Tair, pres, qair, swd, CO2, smlf, daily_Rd, Tground;
    double time = 0.0;
    long long iteration = 0;

    VegetationOracle() : Nh(0), c(ParamsD{}) {}
    VegetationOracle(long nh, const VegParamsD& vp, const ParamsD& cp) : Nh(nh), p(vp), c(cp) {
        for (auto* v : {&C_veg, &nu, &G_C_veg, &G_nu, &LAI_b, &phen, &LAI, &gw_can, &lambda_c, &An, &Rd, &GPP, &Ra, &NPP, &daily_Rd}) v->assign(nh, NF(0));
        // input defaults (prescribed_atmosphere.jl:90-92,148,221-223,14; photosynthesis.jl:76; autotrophic_respiration.jl:38)
        Tair.assign(nh, NF(10));
        pres.assign(nh, NF(101325));
        qair.assign(nh, NF(1.0e-3));
        swd.assign(nh, NF(300));
        CO2.assign(nh, NF(380));
        smlf.assign(nh, NF(1));
        Tground.assign(nh, NF(10));
    }
    FieldVec<NF>* field(int id) {
        FieldVec<NF>* all[] = {&C_veg, &nu, &G_C_veg, &G_nu, &LAI_b, &phen, &LAI, &gw_can, &lambda_c, &An, &Rd, &GPP, &Ra, &NPP,
                                  &Tair, &pres, &qair, &swd, &CO2, &smlf, &daily_Rd, &Tground};
        return (id >= 0 && id < 22) ? all[id] : nullptr;
    }
    // compute_auxiliary!(state, grid, veg::VegetationCarbon, constants, atmos, soil = nothing) (vegetation_carbon.jl:66-104)
    void compute_auxiliary(long lo = 0, long hi = -1) {
        if (hi < 0) hi = Nh;
        for (long i = lo; i < hi; ++i) {
            // (plant available water: no-op without soil, the limiting factor is an input)
            LAI_b[i] = veg_LAI_b(p, C_veg[i]);                                         // carbon dynamics
            phen[i] = veg_phenology_factor<NF>();                                      // phenology
            LAI[i] = veg_LAI(LAI_b[i]);
            NF vpd = compute_vpd(c, pres[i], qair[i], Tair[i]);                        // stomatal conductance, with An of the
            gw_can[i] = veg_gw_can(p, vpd, An[i], CO2[i], LAI[i], smlf[i]);            // previous evaluation (vegetation_carbon.jl:88-92)
            lambda_c[i] = veg_lambda_c(p, vpd);
            NF rd, an;                                                                 // photosynthesis
            veg_respiration_assimilation(p, Tair[i], swd[i], pres[i], CO2[i], LAI[i], lambda_c[i], smlf[i], rd, an);
            Rd[i] = rd;
            An[i] = an;
            GPP[i] = veg_GPP(an);
            Ra[i] = veg_Ra(p, Tair[i], Tground[i], daily_Rd[i], phen[i], C_veg[i], GPP[i]);   // autotrophic respiration
            NPP[i] = veg_NPP(GPP[i], Ra[i]);
        }
    }
    // compute_tendencies!(state, grid, veg::VegetationCarbon) (vegetation_carbon.jl:111-118); tendencies accumulate on reset fields
    void compute_tendencies(long lo = 0, long hi = -1) {
        if (hi < 0) hi = Nh;
        for (long i = lo; i < hi; ++i) {
            G_C_veg[i] = veg_C_veg_tend(p, LAI_b[i], NPP[i]);
            G_nu[i] = veg_nu_tendency(p, LAI_b[i], C_veg[i], NPP[i], nu[i]);
        }
    }
    void explicit_step(NF dt, long lo = 0, long hi = -1) {
        if (hi < 0) hi = Nh;
        for (long i = lo; i < hi; ++i) {
            C_veg[i] = C_veg[i] + G_C_veg[i] * dt;
            nu[i] = nu[i] + G_nu[i] * dt;
        }
    }
    void timestep_euler(double dt, bool finalize) {   // forward_euler.jl:19-31 on the VegetationModel (no closure)
        compute_auxiliary();
        compute_tendencies();
        explicit_step(NF(dt));
        time += dt;
        iteration += 1;
        if (finalize) compute_auxiliary();
    }
    void timestep_heun(double dt, bool finalize) {    // heun.jl:37-71
        compute_auxiliary();
        compute_tendencies();
        VegetationOracle stage = *this;
        stage.explicit_step(NF(dt));
        stage.compute_auxiliary();
        stage.compute_tendencies();
        for (long i = 0; i < Nh; ++i) {
            G_C_veg[i] = (G_C_veg[i] + stage.G_C_veg[i]) / NF(2);
            G_nu[i] = (G_nu[i] + stage.G_nu[i]) / NF(2);
        }
        explicit_step(NF(dt));
        time += dt;
        iteration += 1;
        if (finalize) compute_auxiliary();
    }
};

}  // namespace trm_oracle
