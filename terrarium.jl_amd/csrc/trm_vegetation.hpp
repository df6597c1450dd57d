// trm_vegetation.hpp -- the 0-D vegetation processes of VegetationCarbon on the device.
//
// One thread per column: plant functional type parameters are launch constants, every process is a handful of scalar
// formulas (src/processes/vegetation/*.jl), so a whole `update_state!` + `explicit_step!` of the vegetation state is one
// pass over ~11 per-column inputs and ~14 outputs -- a streaming kernel with ~150 flops per column, far below any
// roofline at N145 (56 951 threads: latency-bound, ~3 us).  Order of the processes and the deliberate use of the PREVIOUS
// evaluation's net assimilation by the stomatal conductance follow compute_auxiliary!(state, grid, veg::VegetationCarbon,
// ...) (vegetation_carbon.jl:66-104).
//
// TRM_VEGETATION_COUPLED (LandModel with vegetation, land_model.jl:79-97): k_surface_veg is the 0-D part of one step --
// plant available water from the soil column, the vegetation processes, canopy interception, canopy evapotranspiration,
// runoff, the surface energy balance x2 and (ADVANCE) the tendencies and explicit step of canopy_water,
// carbon_vegetation and vegetation_area_fraction -- in one launch in front of the soil column kernel, which reads only
// the ground heat flux and the infiltration it leaves behind.
#pragma once
#include "trm_kernels.hpp"

namespace trm {

// trm_vegetation_params converted to NF (field order = include/terrarium_hip.h)
template <class NF> struct VegDev {
    NF tau25, Kc25, Ko25, q10_tau, q10_Kc, q10_Ko, alpha_leaf, alpha_a, alpha_C3, cq, k_ext, T_CO2_high, T_CO2_low, T_photos_high,
        T_photos_low, theta_r, g1, g_min, cn_sapwood, cn_root, aws, SLA, awl, LAI_min, LAI_max, gamma_L, gamma_R, gamma_S, nu_seed,
        gamma_v_min, root_a, root_b, wilting_point, field_capacity, C_mass, alpha_int, canopy_k_ext, w_can_max, tau_w, C_can;
    // physical constants / atmosphere parameters the processes read
    NF eps_mw, one_minus_eps_mw;
    NF sqrt_eps;   // sqrt(eps(NF)): floor of the stomatal conductance (canopy_evapotranspiration.jl:53)
    NF paw_span, rpaw_span;   // field_capacity - wilting_point and its reciprocal, formed on the host in NF
    // launch constants of the Q10 kinetics and the temperature stress (photosynthesis.jl:93-98, 165-188), formed on the host
    NF ln_q10_tau, ln_q10_Kc, ln_q10_Ko, ts_k1, ts_k2, ts_k3;
};
template <class NF> struct VegView {
    long Nh;
    NF *C_veg, *nu, *G_C_veg, *G_nu, *LAI_b, *phen, *LAI, *gw_can, *lambda_c, *An, *Rd, *GPP, *Ra, *NPP;
    const NF *Tair, *pres, *qair, *swd, *CO2, *smlf, *daily_Rd, *Tground;
    long Tground_stride;   // 1 for the input field; the soil's top temperature is read with the level pitch
    // canopy hydrology of the coupled LandModel (null in the standalone VegetationModel)
    NF *w_can, *G_w_can, *I_can, *R_can, *f_can, *rain_ground, *E_can, *transp;
    const NF* SAI;
    NF* paw;                   // plant_available_water [Nh][Nzp]
    const NF* rootf;           // static root fraction per level [Nz] (root_distribution.jl:45-63)
    TRM_DEV NF* smlf_out() const { return const_cast<NF*>(smlf); }   // the coupled model computes the limiting factor itself
};

TRM_DEV double log_(double x) { return log(x); }
TRM_DEV float log_(float x) { return logf(x); }

// carbon_dynamics.jl:62-73, 84-87, 98-105, 116-126
template <class NF> TRM_DEV NF veg_lambda_NPP(const VegDev<NF>& p, NF LAI_b) {
    if (LAI_b < p.LAI_min) return NF(0);
    if (LAI_b <= p.LAI_max) return (LAI_b - p.LAI_min) / (p.LAI_max - p.LAI_min);
    return NF(1);
}
template <class NF> TRM_DEV NF veg_LAI_b(const VegDev<NF>& p, NF C_veg) { return C_veg / ((NF(2) / p.SLA) + p.awl); }
template <class NF> TRM_DEV NF veg_C_veg_tendency(const VegDev<NF>& p, NF LAI_b, NF NPP) {
    const NF lambda = veg_lambda_NPP(p, LAI_b);
    const NF litter = (p.gamma_L / p.SLA + p.gamma_R / p.SLA + p.gamma_S * p.awl) * LAI_b;
    return (NF(1) - lambda) * NPP - litter;
}
// vegetation_dynamics.jl:41-88 (disturbance rate = its minimum, seeding through nu_star)
template <class NF> TRM_DEV NF veg_nu_tendency(const VegDev<NF>& p, NF LAI_b, NF C_veg, NF NPP, NF nu) {
    const NF lambda = veg_lambda_NPP(p, LAI_b);
    const NF nu_star = jl_max(nu, p.nu_seed);
    return (lambda * NPP / C_veg) * nu_star * (NF(1) - nu) - p.gamma_v_min * nu_star;
}
// phenology.jl:32-63: evergreen placeholder, f_deciduous = 0 and phen = 1
template <class NF> TRM_DEV NF veg_LAI(NF LAI_b, NF& phen) {
    const NF f_deciduous = NF(0);
    phen = NF(1);
    return (f_deciduous * phen + (NF(1) - f_deciduous)) * LAI_b;
}
// stomatal_conductance.jl:45-81
template <class NF> TRM_DEV NF veg_gw_can(const VegDev<NF>& p, NF vpd, NF An, NF co2, NF LAI, NF beta) {
    const NF g_min = p.g_min / NF(1000);
    const NF g0 = g_min * (NF(1) - exp_(-p.k_ext * LAI)) * beta;
    return g0 + NF(1.6) * (NF(1) + p.g1 / sqrt_(vpd)) * An / co2 * NF(1.0e6);
}
template <class NF> TRM_DEV NF veg_lambda_c(const VegDev<NF>& p, NF vpd) {
    return NF(1) - NF(1) / (NF(1) + p.g1 / sqrt_(vpd * NF(1.0e-3)));
}
// photosynthesis.jl:165-188
template <class NF> TRM_DEV NF veg_temperature_stress(const VegDev<NF>& p, NF T_air) {
    const NF k1 = p.ts_k1, k2 = p.ts_k2, k3 = p.ts_k3;
    if (p.T_CO2_low < T_air && T_air < p.T_CO2_high) {
        const NF low = NF(1) / (NF(1) + exp_(k1 * (k2 - T_air)));
        const NF high = NF(1) - NF(0.01) * exp_(k3 * (T_air - p.T_photos_high));
        return low * high;
    }
    return NF(0);
}
// q10^e of the Q10 responses (photosynthesis.jl:93-98) as exp(e ln q10), ln q10 a launch constant: |e ln q10| < 4 over
// the temperature range of the model, so the product's rounding moves the result by < 1e-15 relative (fp64) -- the
// generic x^y of Base costs ~4x the instructions on the serial 0-D chain that bounds the kernel.  e = 0 gives exactly 1.
template <class NF> TRM_DEV NF veg_q10(NF ln_q10, NF e) { return exp_(e * ln_q10); }
// photosynthesis.jl:290-337 compute_respiration_assimilation -> (Rd, An) [gC/m^2/s]
template <class NF> TRM_DEV void veg_respiration_assimilation(const VegDev<NF>& p, NF T_air, NF swdown, NF pres, NF co2, NF LAI, NF lambda_c, NF beta, NF& Rd, NF& An) {
    const NF pres_O2 = NF(0.209) * pres;            // physics_utils.jl:16-20
    const NF pres_a = co2 * NF(1.0e-6) * pres;      // physics_utils.jl:27-30
    Rd = NF(0);
    An = NF(0);
    if (swdown > NF(0) && T_air > NF(-3) && LAI > NF(0)) {
        const NF e = (T_air - NF(25)) * NF(0.1);    // Q10 kinetics, :93-98
        const NF tau = p.tau25 * veg_q10(p.ln_q10_tau, e), Kc = p.Kc25 * veg_q10(p.ln_q10_Kc, e), Ko = p.Ko25 * veg_q10(p.ln_q10_Ko, e);
        const NF Gamma = pres_O2 / (NF(2) * tau);                                                        // :111-114
        const NF PAR = NF(0.5) * swdown * (NF(1) - p.alpha_leaf) * p.cq;                                 // :122-126
        const NF APAR = p.alpha_a * PAR * (NF(1) - exp_(-p.k_ext * LAI));                                // :138-143
        const NF pres_i = lambda_c * pres_a;                                                             // :155-158
        const NF T_stress = veg_temperature_stress(p, T_air);
        const NF c1 = p.alpha_C3 * T_stress * p.C_mass * (pres_i - Gamma) / (pres_i + NF(2) * Gamma);    // :206-217
        const NF c2 = (pres_i - Gamma) / (pres_i + Kc * (NF(1) + pres_O2 / Ko));
        const NF Vc_max = c1 * APAR * (pres_i + Kc * (NF(1) + pres_O2 / Ko)) / (pres_i - Gamma);         // :230-234
        Rd = p.alpha_C3 * Vc_max * beta;                                                                 // :263-267
        const NF JE = c1 * APAR, JC = c2 * Vc_max, s = JE + JC;                                          // :245-250, 278-283
        const NF Ag = (s - sqrt_(s * s - NF(4) * p.theta_r * JE * JC)) / (NF(2) * p.theta_r) * beta;
        An = Ag - Rd;
    }
}
// autotrophic_respiration.jl:46-126 -> Ra [kgC/m^2/s]
template <class NF> TRM_DEV NF veg_autotrophic_respiration(const VegDev<NF>& p, NF T_air, NF T_soil, NF Rd_daily, NF phen, NF C_veg, NF GPP) {
    const NF f_air = exp_(NF(308.56) * (NF(1) / NF(56.02) - NF(1) / (NF(46.02) + T_air)));
    const NF f_soil = boolmul(T_soil > NF(7), exp_(NF(308.56) * (NF(1) / NF(56.02) - NF(1) / (NF(46.02) + T_soil))));
    const NF resp10 = NF(0.066);
    const NF R_leaf = Rd_daily / NF(1000);
    const NF R_stem = resp10 * f_air * (p.awl * ((NF(2) / p.SLA) + p.awl)) / (C_veg * p.aws * p.cn_sapwood);
    const NF R_root = resp10 * f_soil * phen * (NF(2) / p.SLA) / (p.SLA * C_veg * p.cn_root);
    const NF Rm = R_leaf + R_stem + R_root;
    const NF Rg = NF(0.25) * (GPP - Rm);
    return Rm + Rg;
}
// plant_available_water.jl:77-94
template <class NF> TRM_DEV NF veg_plant_available_water(const VegDev<NF>& p, NF theta_w) {
    return jl_max(jl_min(NF(1), div_const(theta_w - p.wilting_point, p.paw_span, p.rpaw_span)), NF(0));
}

// what one vegetation column carries through a step
template <class NF> struct VegColumn {
    NF C_veg, nu, An;                     // prognostic state + the carried net assimilation
    NF LAI_b, phen, LAI, gw_can, lambda_c, Rd, GPP, Ra, NPP, G_C_veg, G_nu;
};
template <class NF> struct VegInputs { NF Tair, pres, qair, swd, CO2, smlf, daily_Rd, Tground; };

// compute_auxiliary!(state, grid, veg, constants, atmos, soil) for one column (vegetation_carbon.jl:66-104)
// ... in two halves: carbon pools -> leaf area -> stomatal conductance (what the canopy evapotranspiration needs), then
// photosynthesis -> respiration -> net primary production (what the carbon tendencies need)
template <class NF> TRM_DEV void veg_auxiliary_conductance(const VegDev<NF>& p, const VegInputs<NF>& in, VegColumn<NF>& c) {
    c.LAI_b = veg_LAI_b(p, c.C_veg);
    c.LAI = veg_LAI(c.LAI_b, c.phen);
    // compute_vpd at the air temperature (prescribed_atmosphere.jl:176-182, physical_constants.jl:83-97)
    const NF e_sat = saturation_vapor_pressure(in.Tair);
    const NF e_air = in.qair * in.pres / (p.eps_mw + p.one_minus_eps_mw * in.qair);
    const NF vpd = jl_max(e_sat - e_air, NF(0.1));
    c.gw_can = veg_gw_can(p, vpd, c.An, in.CO2, c.LAI, in.smlf);     // An of the previous evaluation
    c.lambda_c = veg_lambda_c(p, vpd);
}
template <class NF> TRM_DEV void veg_auxiliary_carbon(const VegDev<NF>& p, const VegInputs<NF>& in, VegColumn<NF>& c) {
    veg_respiration_assimilation(p, in.Tair, in.swd, in.pres, in.CO2, c.LAI, c.lambda_c, in.smlf, c.Rd, c.An);
    c.GPP = c.An * NF(1.0e-3);
    c.Ra = veg_autotrophic_respiration(p, in.Tair, in.Tground, in.daily_Rd, c.phen, c.C_veg, c.GPP);
    c.NPP = c.GPP - c.Ra;
}
template <class NF> TRM_DEV void veg_auxiliary(const VegDev<NF>& p, const VegInputs<NF>& in, VegColumn<NF>& c) {
    veg_auxiliary_conductance(p, in, c);
    veg_auxiliary_carbon(p, in, c);
}
template <class NF> TRM_DEV void veg_tendencies(const VegDev<NF>& p, VegColumn<NF>& c) {
    c.G_C_veg = veg_C_veg_tendency(p, c.LAI_b, c.NPP);
    c.G_nu = veg_nu_tendency(p, c.LAI_b, c.C_veg, c.NPP, c.nu);
}

// MODE: 0 compute_auxiliary!, 1 + compute_tendencies! (update_state!), 2 one ForwardEuler step, 3 one Heun step;
// nsteps > 1 (modes 2, 3) keeps the column in registers.  `finalize`: compute_auxiliary! once more at the end.
// 4 compute_tendencies! alone (from the stored auxiliaries), 5 explicit_step! alone.
enum { VEG_AUX = 0, VEG_UPDATE = 1, VEG_EULER = 2, VEG_HEUN = 3, VEG_TEND = 4, VEG_EXPLICIT = 5 };
template <class NF, int MODE>
__global__ void __launch_bounds__(256) k_vegetation(VegView<NF> v, VegDev<NF> p, NF dt, int nsteps, int finalize) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= v.Nh) return;
    if (MODE == VEG_TEND) {
        VegColumn<NF> t;
        t.C_veg = v.C_veg[i]; t.nu = v.nu[i]; t.LAI_b = v.LAI_b[i]; t.NPP = v.NPP[i];
        veg_tendencies(p, t);
        v.G_C_veg[i] = t.G_C_veg;       // (assigned, not accumulated: carbon_dynamics.jl:184, vegetation_dynamics.jl:150)
        v.G_nu[i] = t.G_nu;
        if (v.w_can) v.G_w_can[i] = v.I_can[i] - v.E_can[i] - v.R_can[i];   // canopy_interception.jl:100-106, 201-215
        return;
    }
    if (MODE == VEG_EXPLICIT) {
        v.C_veg[i] = v.C_veg[i] + v.G_C_veg[i] * dt;
        v.nu[i] = v.nu[i] + v.G_nu[i] * dt;
        if (v.w_can) v.w_can[i] = v.w_can[i] + v.G_w_can[i] * dt;
        return;
    }
    VegInputs<NF> in = {v.Tair[i], v.pres[i], v.qair[i], v.swd[i], v.CO2[i], v.smlf[i], v.daily_Rd[i], v.Tground[i * v.Tground_stride]};
    VegColumn<NF> c;
    c.C_veg = v.C_veg[i];
    c.nu = v.nu[i];
    c.An = v.An[i];
    c.G_C_veg = c.G_nu = NF(0);
    if (MODE == VEG_AUX || MODE == VEG_UPDATE) {
        veg_auxiliary(p, in, c);
        if (MODE == VEG_UPDATE) veg_tendencies(p, c);
    } else {
        for (int s = 0; s < nsteps; ++s) {
            veg_auxiliary(p, in, c);
            veg_tendencies(p, c);
            if (MODE == VEG_HEUN) {   // heun.jl:37-71: predictor, tendencies at the stage, average
                VegColumn<NF> st = c;
                st.C_veg = c.C_veg + c.G_C_veg * dt;
                st.nu = c.nu + c.G_nu * dt;
                veg_auxiliary(p, in, st);
                veg_tendencies(p, st);
                c.G_C_veg = (c.G_C_veg + st.G_C_veg) / NF(2);
                c.G_nu = (c.G_nu + st.G_nu) / NF(2);
            }
            c.C_veg = c.C_veg + c.G_C_veg * dt;
            c.nu = c.nu + c.G_nu * dt;
        }
        if (finalize) veg_auxiliary(p, in, c);
        v.C_veg[i] = c.C_veg;
        v.nu[i] = c.nu;
    }
    if (MODE != VEG_AUX) { v.G_C_veg[i] = c.G_C_veg; v.G_nu[i] = c.G_nu; }
    v.LAI_b[i] = c.LAI_b; v.phen[i] = c.phen; v.LAI[i] = c.LAI; v.gw_can[i] = c.gw_can; v.lambda_c[i] = c.lambda_c;
    v.An[i] = c.An; v.Rd[i] = c.Rd; v.GPP[i] = c.GPP; v.Ra[i] = c.Ra; v.NPP[i] = c.NPP;
}

// FieldCapacityLimitedPAW + StaticExponentialRootDistribution (plant_available_water.jl:36-94, root_distribution.jl:38-63):
// plant_available_water per cell from the liquid water content, and its root-weighted column integral
// Integral(PAW * root_fraction / dz) dz = sum_k PAW_k * root_fraction_k, the soil moisture limiting factor.
template <class NF> __global__ void __launch_bounds__(256) k_plant_available_water(const NF* sat, const NF* liq, const NF* root_fraction, NF* paw, NF* smlf,
                                                                                   long Nh, int Nz, int Nzp, NF por, VegDev<NF> p, const NF* dzc) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Nh) return;
    NF acc = NF(0);
    for (int k = 0; k < Nz; ++k) {      // Oceananigans' Integral sums from the bottom cell up
        const long c = i * Nzp + k;
        const NF water = (sat[c] * por) * liq[c];
        const NF w = veg_plant_available_water(p, water);
        paw[c] = w;
        acc = acc + (w * root_fraction[c] / dzc[k]) * dzc[k];
    }
    smlf[i] = acc;
}

// ---- canopy hydrology (canopy_interception.jl:64-124, canopy_evapotranspiration.jl:52-82,163-176) -----------------------
template <class NF> struct CanopyOut { NF f_can, I_can, R_can, rain_ground, E_can, transp; };
template <class NF> TRM_DEV void canopy_interception(const VegDev<NF>& p, NF rain, NF LAI, NF SAI, NF w_can, CanopyOut<NF>& o) {
    const NF w_max = p.w_can_max * (LAI + SAI);
    o.f_can = w_max > NF(0) ? w_can / w_max : NF(0);
    o.I_can = p.alpha_int * rain * (NF(1) - exp_(-p.canopy_k_ext * (LAI + SAI)));
    o.R_can = jl_max(w_can, NF(0)) / p.tau_w;
    o.rain_ground = rain - o.I_can + o.R_can;
}

// The 0-D part of compute_auxiliary!(state, model::LandModel) with vegetation.
//
// NZP = 32 / 64 (the level pitch, Nz <= 64): a workgroup of 256 threads owns 64 consecutive columns.  Phase 1, all four
// waves: the columns' cells are one contiguous run of the z-fastest layout, so sat / liq are read (and
// plant_available_water written) fully coalesced; each cell's W r / dz * dz term goes to LDS (row pitch NZP + 1: the
// phase-2 reads of 64 different rows then spread over the banks).  Phase 2, the first wave, one column per lane: the
// bottom-up sum of the column's terms in the sequential order of Oceananigans' Integral (a DPP / tree reduction would
// change the rounding; a per-level readlane or shuffle loop costs ~15 VALU per level and wave -- measured 23-27 us for
// the N145 columns against 37 us for the whole soil step), then every 0-D process of the column.
// NZP = 0 (Nz > 64, reference-order path only): one thread per column walks its cells.
// Run-time switches (wave-uniform): from_state -- the top-face hydraulic conductivity is formed from the top cell (in
// front of the fused column kernel, which does not materialise K) instead of read from the hydraulic_conductivity
// field; top_arrays -- the top cell's (T, sat, liq) come from the compact per-column arrays the fused step wrote;
// advance -- compute_tendencies! + explicit_step! of the 0-D prognostics follow in the same launch; store_paw -- the
// per-cell plant_available_water field is materialised.
// advance: 0 compute_auxiliary! only; 1 + compute_tendencies! + explicit_step! of the 0-D prognostics; 2 + compute_tendencies!
// alone (Heun's evaluation at the stage); 3 Heun's first stage on the state: tendencies stored, the PREDICTED 0-D prognostics,
// the net assimilation and the skin temperature written to the stage's arrays (st_*), the state's left as they are.
template <class NF> struct SurfaceVegArgs {
    NF dt;
    int richards, from_state, top_arrays, advance, store_paw;
    NF *st_w_can, *st_C_veg, *st_nu, *st_An, *st_Ts;
};

template <class NF> TRM_DEV NF paw_term(const VegDev<NF>& vp, NF por, NF sat, NF liq, NF rootf, NF dz, NF rdz, NF& w) {
    w = jl_max(jl_min(NF(1), div_const((sat * por) * liq - vp.wilting_point, vp.paw_span, vp.rpaw_span)), NF(0));
    return div_const(w * rootf, dz, rdz) * dz;
}

// Phases 1 and 2 of the workgroup-cooperative plant available water (see k_surface_veg): returns true in the threads
// that own a column (waves 0 and 1, lane = column i) with the column's soil moisture limiting factor in `smlf`.
template <class NF, int NZP>
TRM_DEV bool block_soil_moisture_limit(const NF* __restrict__ sat, const NF* __restrict__ liq, const NF* __restrict__ rootf, const NF* __restrict__ dzc,
                                       const NF* __restrict__ rdzc, NF* __restrict__ paw, long Nh, int Nz, NF por, const VegDev<NF>& vp, long& i, NF& smlf) {
    __shared__ NF terms[64 * (NZP + 1)];
    const long i0 = (long)blockIdx.x * 64;
    const int ncol = (int)((Nh - i0) < 64 ? (Nh - i0) : 64);
    const long base = i0 * NZP;
    // a thread keeps its level k and walks the columns c0, c0 + 256 / NZP, ...: all of its loads are issued before the
    // first use (one memory round trip instead of one per cell), the per-level constants are read once
    constexpr int PASSES = NZP / 4, CPP = 256 / NZP;
    const int k = threadIdx.x % NZP, c0 = threadIdx.x / NZP;
    const bool level = k < Nz;
    const NF rf = level ? rootf[k] : NF(0), dz = level ? dzc[k] : NF(1), rdz = level ? rdzc[k] : NF(1);
    // Loads and stores WITHOUT a branch per pass: a conditional load or store is its own basic block, and where such blocks meet the
    // compiler waits for every vector memory operation that may be pending (`s_waitcnt vmcnt(0)`) -- with the store of pass j
    // conditional, pass j + 1 waited for it: eight serialised trips to memory per thread (round 4, from the disassembly).  The loads
    // take clamped indices (a lane beyond the column count or the level count re-reads a valid cell), the terms of rows beyond the
    // column count go to LDS rows nobody sums, and the stores of a FULL block (every block but the last) sit in one block.
    const int kc = level ? k : Nz - 1;
    NF s_[PASSES], l_[PASSES], w_[PASSES], t_[PASSES];
#pragma unroll
    for (int j = 0; j < PASSES; ++j) {
        const int col = c0 + j * CPP, colc = col < ncol ? col : ncol - 1;
        s_[j] = sat[base + colc * NZP + kc];
        l_[j] = liq[base + colc * NZP + kc];
    }
#pragma unroll
    for (int j = 0; j < PASSES; ++j) t_[j] = paw_term(vp, por, s_[j], l_[j], rf, dz, rdz, w_[j]);
    if (level) {
#pragma unroll
        for (int j = 0; j < PASSES; ++j) terms[(c0 + j * CPP) * (NZP + 1) + k] = t_[j];
        if (paw) {
            if (ncol == 64) {
#pragma unroll
                for (int j = 0; j < PASSES; ++j) paw[base + (c0 + j * CPP) * NZP + k] = w_[j];
            } else {
#pragma unroll
                for (int j = 0; j < PASSES; ++j)
                    if (c0 + j * CPP < ncol) paw[base + (c0 + j * CPP) * NZP + k] = w_[j];
            }
        }
    }
    __syncthreads();
    // lane = column in the first TWO waves (k_surface_veg runs one half of the 0-D work in each; both need the factor)
    const int row = threadIdx.x & 63;
    if (threadIdx.x >= 128 || row >= ncol) return false;
    i = i0 + row;
    smlf = NF(0);
    for (int kk = 0; kk < Nz; ++kk) smlf = smlf + terms[row * (NZP + 1) + kk];
    return true;
}
// trm_compute_plant_available_water for Nz <= 64
template <class NF, int NZP>
__global__ void __launch_bounds__(256) k_plant_available_water_block(const NF* sat, const NF* liq, const NF* rootf, const NF* dzc, const NF* rdzc, NF* paw, NF* smlf,
                                                                     long Nh, int Nz, NF por, VegDev<NF> p) {
    long i;
    NF acc;
    if (block_soil_moisture_limit<NF, NZP>(sat, liq, rootf, dzc, rdzc, paw, Nh, Nz, por, p, i, acc) && threadIdx.x < 64) smlf[i] = acc;
}

template <class NF, int NZP, int HYD>
__global__ void __launch_bounds__(256) k_surface_veg(View<NF> v_arg, DevParams<NF> p_arg, VegView<NF> vv_arg, VegDev<NF> vp_arg, SurfaceVegArgs<NF> a) {
    // ~330 scalar kernel arguments: every section below reads the ones it needs afresh from the kernarg segment
    // (kernarg_reload, trm_kernels.hpp), so that their live ranges end with the section instead of spilling to VGPR lanes
    // (measured: 501 v_readlane + 190 v_writelane in the first version of this kernel)
    constexpr unsigned off_p = round_up_to((unsigned)sizeof(View<NF>), (unsigned)alignof(DevParams<NF>));
    constexpr unsigned off_vv = round_up_to(off_p + (unsigned)sizeof(DevParams<NF>), (unsigned)alignof(VegView<NF>));
    constexpr unsigned off_vp = round_up_to(off_vv + (unsigned)sizeof(VegView<NF>), (unsigned)alignof(VegDev<NF>));
    constexpr bool BLOCK = NZP > 0;
    const View<NF>& v = v_arg;
    const DevParams<NF>& p = p_arg;
    const VegView<NF>& vv = vv_arg;
    // role 0 (wave 0): stomatal conductance, canopy hydrology, runoff, surface energy balance;  role 1 (wave 1):
    // photosynthesis, respiration, carbon and area-fraction tendencies.  The two chains are independent (the conductance
    // uses the PREVIOUS evaluation's net assimilation) and each is a serial latency chain, so they run side by side.
    // role 2 (one thread per column, Nz > 64): both.
    const int role = BLOCK ? (int)(threadIdx.x >> 6) : 2;
    long i = BLOCK ? (long)blockIdx.x * 64 + (threadIdx.x & 63) : (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool owner = (role < 2 || !BLOCK) && i < v.Nh;
    if (!BLOCK && !owner) return;
    uint32_t viol = 0;
    // per-column inputs first: their latency overlaps the cell phase, and the previous net assimilation is read before
    // the barrier, i.e. before the other wave can store the new one
    VegInputs<NF> vin = {};
    VegColumn<NF> vc = {};
    SebIn<NF> in = {};
    NF SAI = NF(0), w_can = NF(0), Ts = NF(0), S = NF(0), sat_top = NF(0), liq_top = NF(0), T_ground = NF(0), Kf_field = NF(0);
    if (owner) {
        const long top = i * v.Nzp + (v.Nz - 1);
        sat_top = a.top_arrays ? v.top_sat[i] : v.sat[top];
        liq_top = a.top_arrays ? v.top_liq[i] : v.liq[top];
        T_ground = a.top_arrays ? v.top_T[i] : v.T[top];       // ground_temperature = the top soil cell (soil_energy.jl:48-57)
        vin = {v.Tair[i], v.pres[i], v.qair[i], v.swd[i], vv.CO2[i], NF(0), vv.daily_Rd[i], T_ground};
        vc.C_veg = vv.C_veg[i];
        vc.nu = vv.nu[i];
        vc.An = vv.An[i];
        if (role != 1) {
            in = {vin.Tair, vin.pres, v.wind[i], vin.qair, v.rain[i], vin.swd, v.lwd[i], NF(0), NF(0), NF(0)};
            seb_radiation_inputs(p, v.albedo, v.emissivity, (unsigned)i * (unsigned)sizeof(NF), in);
            SAI = vv.SAI[i];
            w_can = vv.w_can[i];
            Ts = v.Ts[i];
            S = v.S[i];
            if (!a.from_state) Kf_field = v.Kf[top];
        }
    }
    NF smlf = NF(0);
    if constexpr (BLOCK) {
        long i_;
        const View<NF>& v = kernarg_reload<View<NF>>(0);
        const VegView<NF>& vv = kernarg_reload<VegView<NF>>(off_vv);
        if (!block_soil_moisture_limit<NF, NZP>(v.sat, v.liq, vv.rootf, v.dzc, v.rdzc, a.store_paw ? vv.paw : nullptr, v.Nh, v.Nz,
                                                kernarg_reload<DevParams<NF>>(off_p).por, kernarg_reload<VegDev<NF>>(off_vp), i_, smlf)) return;
    } else {
        const VegDev<NF>& vp = vp_arg;
        for (int k = 0; k < v.Nz; ++k) {
            const long c = i * v.Nzp + k;
            NF w;
            smlf = smlf + paw_term(vp, p.por, v.sat[c], v.liq[c], vv.rootf[k], v.dzc[k], v.rdzc[k], w);
            if (a.store_paw) vv.paw[c] = w;
        }
    }
    vin.smlf = smlf;
    veg_auxiliary_conductance(kernarg_reload<VegDev<NF>>(off_vp), vin, vc);       // (cheap; both roles need the leaf area and the CO2 ratio)
    if (role != 1) {
        const DevParams<NF>& p = kernarg_reload<DevParams<NF>>(off_p);
        const VegDev<NF>& vp = kernarg_reload<VegDev<NF>>(off_vp);
        const NF dz_top = kernarg_reload<View<NF>>(0).g.dzc_top;
        // canopy interception, then canopy evapotranspiration (surface_hydrology.jl:36-49)
        CanopyOut<NF> co;
        canopy_interception(vp, in.rain, vc.LAI, SAI, w_can, co);
        SebOut<NF> o;
        o.Ts = Ts;
        const NF ra = aerodynamic_resistance(p, in.wind);
        {
            const NF dqs = humidity_vpd(p, in.pres, in.qair, o.Ts);       // canopy -> atmosphere
            const NF dqg = humidity_vpd(p, in.pres, in.qair, T_ground);   // ground -> canopy
            const NF re = (NF(1) - exp_(-vc.LAI - SAI)) / (vp.C_can * jl_max(in.wind, p.min_windspeed));
            const NF beta = evaporation_resistance_factor(p, sat_top, liq_top);
            const NF rs = NF(1) / jl_max(vc.gw_can, vp.sqrt_eps);
            co.transp = dqs / (ra + rs);
            o.evap = beta * dqg / (ra + re);
            co.E_can = co.f_can * dqs / ra;
        }
        // runoff of the rain that reaches the ground (direct_surface_runoff.jl:87-117)
        const NF Kf_top = a.from_state ? conductivity_hydraulic<NF, HYD, false>(p, liq_top, fractions(p, sat_top, liq_top, viol)) : Kf_field;
        surface_runoff(p, co.rain_ground, sat_top, Kf_top, S, a.richards != 0, o);
        // surface energy balance x2 with the humidity flux of all three pathways (canopy_evapotranspiration.jl:97-102)
        const NF Q_h = o.evap + co.E_can + co.transp;
        // (one copy of the flux code, executed four times: the kernel is bound by cold instruction fetch, not by issue)
#pragma clang loop unroll(disable)
        for (int n = 0; n < 4; ++n) {
            seb_fluxes_humidity(p, in, ra, Q_h, o);
            if ((n & 1) == 0) o.Ts = T_ground - div_const(o.ghf * dz_top, p.kappa_s2, p.rkappa_s2);
        }
        const View<NF>& v = kernarg_reload<View<NF>>(0);
        const VegView<NF>& vv = kernarg_reload<VegView<NF>>(off_vv);
        vv.smlf_out()[i] = smlf;
        vv.LAI_b[i] = vc.LAI_b; vv.phen[i] = vc.phen; vv.LAI[i] = vc.LAI; vv.gw_can[i] = vc.gw_can; vv.lambda_c[i] = vc.lambda_c;
        v.Ts[i] = o.Ts; v.ghf[i] = o.ghf; v.swu[i] = o.swu; v.lwu[i] = o.lwu; v.rnet[i] = o.rnet;
        v.Hs[i] = o.Hs; v.Hl[i] = o.Hl; v.evap[i] = o.evap; v.infil[i] = o.infil; v.runoff[i] = o.runoff;
        vv.f_can[i] = co.f_can; vv.I_can[i] = co.I_can; vv.R_can[i] = co.R_can; vv.rain_ground[i] = co.rain_ground;
        vv.E_can[i] = co.E_can; vv.transp[i] = co.transp;
        if (a.advance) {   // compute_tendencies! (+ explicit_step!) of the canopy water (land_model.jl:90-97)
            const NF G_w = co.I_can - co.E_can - co.R_can;
            vv.G_w_can[i] = G_w;
            if (a.advance == 1) vv.w_can[i] = w_can + G_w * a.dt;
            if (a.advance == 3) {
                a.st_w_can[i] = w_can + G_w * a.dt;
                a.st_Ts[i] = o.Ts + NF(0) * a.dt;          // the stage's zero-tendency skin temperature
            }
        }
    }
    if (role != 0) {
        const VegDev<NF>& vp = kernarg_reload<VegDev<NF>>(off_vp);
        veg_auxiliary_carbon(vp, vin, vc);
        const VegView<NF>& vv = kernarg_reload<VegView<NF>>(off_vv);
        vv.An[i] = vc.An; vv.Rd[i] = vc.Rd; vv.GPP[i] = vc.GPP; vv.Ra[i] = vc.Ra; vv.NPP[i] = vc.NPP;
        if (a.advance) {   // ... and of the vegetation carbon and area fraction
            veg_tendencies(vp, vc);
            vv.G_C_veg[i] = vc.G_C_veg; vv.G_nu[i] = vc.G_nu;
            if (a.advance == 1) {
                vv.C_veg[i] = vc.C_veg + vc.G_C_veg * a.dt;
                vv.nu[i] = vc.nu + vc.G_nu * a.dt;
            }
            if (a.advance == 3) {
                a.st_C_veg[i] = vc.C_veg + vc.G_C_veg * a.dt;
                a.st_nu[i] = vc.nu + vc.G_nu * a.dt;
                a.st_An[i] = vc.An;                        // the stage starts from the state's evaluation (copyto!(stage, state))
            }
        }
    }
    if (viol) atomicOr(kernarg_reload<View<NF>>(0).status, viol);
}

// Heun's average_tendencies! + explicit_step! for the three 0-D prognostics of the coupled LandModel (heun.jl:27-35, 64-69):
// G <- (G_state + G_stage) / 2, u <- u + G dt.
template <class NF> __global__ void __launch_bounds__(256) k_heun_average_0d(VegView<NF> st, VegView<NF> sg, NF dt, long Nh) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Nh) return;
    const NF Gw = (st.G_w_can[i] + sg.G_w_can[i]) / NF(2), Gc = (st.G_C_veg[i] + sg.G_C_veg[i]) / NF(2), Gn = (st.G_nu[i] + sg.G_nu[i]) / NF(2);
    st.G_w_can[i] = Gw; st.G_C_veg[i] = Gc; st.G_nu[i] = Gn;
    st.w_can[i] = st.w_can[i] + Gw * dt;
    st.C_veg[i] = st.C_veg[i] + Gc * dt;
    st.nu[i] = st.nu[i] + Gn * dt;
}

}  // namespace trm
