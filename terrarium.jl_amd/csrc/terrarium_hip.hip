// terrarium_hip.hip -- context management, the step sequences and the C ABI of libterrarium_hip.so
// (include/terrarium_hip.h).  gfx950 only; no CPU fallback.  The kernel instantiations live in the trm_launch_*.hip files
// (trm_host.hpp declares their launchers); this file launches only the small data-movement kernels (transposition, ring
// scatter / gather, reductions).
#include "trm_host.hpp"

#include <dlfcn.h>

using namespace trm;
using namespace trmh;

namespace {
thread_local std::string g_create_error;
}  // namespace

namespace trmh {
int fail(trm_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg;
    else g_create_error = msg;
    return code;
}
}  // namespace trmh

namespace {

// ---- grid: ColumnGrid (column_grid.jl:20-34) + Oceananigans' stretched-coordinate recipe --------------
// thickness[0] is the surface layer.  All derived quantities are formed in NF.
template <class NF> struct HostGrid {
    std::vector<NF> zF, zC, dzc, rdzc, rdzf, dzf, psiz;  // faces 0..Nz ; centres 0..Nz-1
    NF dzf_bot, dzf_top;
    void build(int Nz, const double* thickness) {
        // z_coords = convert.(NF, vcat(-reverse(cumsum(z_thick)), 0))
        std::vector<double> cs(Nz);
        double acc = thickness[0];
        cs[0] = acc;
        for (int n = 1; n < Nz; ++n) { acc = acc + thickness[n]; cs[n] = acc; }
        zF.resize(Nz + 1);
        for (int f = 0; f < Nz; ++f) zF[f] = (NF)(-cs[Nz - 1 - f]);
        zF[Nz] = NF(0);
        // halo faces continue with the boundary cell's spacing (Bounded topology)
        NF Fm1 = zF[0] - (zF[1] - zF[0]);
        NF Fp1 = zF[Nz] + (zF[Nz] - zF[Nz - 1]);
        zC.resize(Nz);
        dzc.resize(Nz);
        rdzc.resize(Nz);
        for (int k = 0; k < Nz; ++k) {
            zC[k] = (zF[k + 1] + zF[k]) / NF(2);
            dzc[k] = zF[k + 1] - zF[k];
            rdzc[k] = NF(1) / dzc[k];
        }
        NF Cm1 = (zF[0] + Fm1) / NF(2);
        NF Cp1 = (Fp1 + zF[Nz]) / NF(2);
        dzf.resize(Nz + 1);
        rdzf.resize(Nz + 1);
        for (int f = 0; f <= Nz; ++f) {
            NF hi = (f < Nz) ? zC[f] : Cp1;
            NF lo = (f > 0) ? zC[f - 1] : Cm1;
            dzf[f] = hi - lo;
            rdzf[f] = NF(1) / dzf[f];
        }
        dzf_bot = dzf[0];
        dzf_top = dzf[Nz];
        // elevation head psi_z = z - z_ref with z_ref the surface face (soil_hydraulic_closures.jl:112-121)
        psiz.resize(Nz);
        for (int k = 0; k < Nz; ++k) psiz[k] = zC[k] - zF[Nz];
    }
};

template <class NF> DevParams<NF> make_dev_params(const trm_params& q) {
    DevParams<NF> p;
    NF rho_soc = (NF)q.rho_soc, rho_org = (NF)q.rho_org, por_m = (NF)q.por_mineral, por_o = (NF)q.por_organic;
    p.org = rho_soc / ((NF(1) - por_o) * rho_org);               // homogeneous_strat.jl:34-44
    p.por = (NF(1) - p.org) * por_m + p.org * por_o;             // homogeneous_strat.jl:51-61
    p.rpor = NF(1) / p.por;
    p.solid_frac = NF(1) - p.por;                                // soil_volume.jl:62
    p.frac_organic = p.solid_frac * p.org;                       // soil_volume.jl:103-107
    p.frac_mineral = p.solid_frac * (NF(1) - p.org);
    p.sk_water = std::sqrt((NF)q.k_water);
    p.sk_ice = std::sqrt((NF)q.k_ice);
    p.sk_air = std::sqrt((NF)q.k_air);
    p.kterm_mineral = std::sqrt((NF)q.k_mineral) * p.frac_mineral;
    p.kterm_organic = std::sqrt((NF)q.k_organic) * p.frac_organic;
    p.c_water = (NF)q.c_water;
    p.c_ice = (NF)q.c_ice;
    p.c_air = (NF)q.c_air;
    p.cterm_mineral = (NF)q.c_mineral * p.frac_mineral;
    p.cterm_organic = (NF)q.c_organic * p.frac_organic;
    p.L = (NF)q.rho_w * (NF)q.Lsl;
    p.K_sat = (NF)q.K_sat;
    p.theta_res = (NF)q.theta_res;
    p.theta_span = p.por - p.theta_res;   // (theta_sat - theta_res) with theta_sat = por at every call site
    p.rtheta_span = NF(1) / p.theta_span;
    p.bc_psi_s = (NF)q.bc_psi_s;
    p.vg_alpha = (NF)q.vg_alpha;
    p.impedance = (NF)q.impedance;
    {   // fully frozen cells: y = -impedance * (1 - 0); Base.:^ takes pow_body(10.0, Int(y)) when y is integer-valued
        const NF y = -p.impedance * (NF(1) - NF(0));
        p.impedance_int = ((NF)(int)y == y && y > NF(-4096) && y < NF(4096)) ? 1 : 0;
        p.I_ice_frozen = p.impedance_int ? pow_int(NF(10), (int)y) : NF(0);
    }
    p.vwc_forcing = (NF)q.vwc_forcing;
    p.neg_inv_alpha = NF(-1) / p.vg_alpha;
    NF lam = (NF)q.bc_lambda, n = (NF)q.vg_n;
    NF m = NF(1) - NF(1) / n;
    p.bc_lambda = make_pow_spec<NF>(lam);
    p.bc_neg_inv_lambda = make_pow_spec<NF>(NF(-1) / lam);
    p.vg_n = make_pow_spec<NF>(n);
    p.vg_neg_m = make_pow_spec<NF>(-m);
    p.vg_neg_inv_m = make_pow_spec<NF>(NF(-1) / m);
    p.vg_inv_n = make_pow_spec<NF>(NF(1) / n);
    p.vgk_e1 = make_pow_spec<NF>(n / (n + NF(1)));
    p.vgk_e2 = make_pow_spec<NF>((n - NF(1)) / n);
    p.albedo = (NF)q.albedo;
    p.emissivity = (NF)q.emissivity;
    p.one_minus_emissivity = NF(1) - p.emissivity;
    p.eps_sigma = p.emissivity * (NF)q.sigma;
    p.sigma = (NF)q.sigma;
    p.prescribed_albedo = q.prescribed_albedo;
    p.kappa_s2 = NF(2) * (NF)q.kappa_s;
    p.rkappa_s2 = NF(1) / p.kappa_s2;
    p.C_h = (NF)q.C_h;
    p.min_windspeed = (NF)q.min_windspeed;
    p.tau_r = (NF)q.tau_r;
    p.rtau_r = NF(1) / p.tau_r;
    p.beta_evap = (NF)q.beta_evap;
    p.field_capacity = (NF)q.field_capacity;
    p.evap_resistance = q.evap_resistance;
    p.Tref = (NF)q.Tref;
    p.eps_mw = (NF)q.eps_mw;
    p.one_minus_eps_mw = NF(1) - p.eps_mw;
    p.ca_rhoa = (NF)q.c_a * (NF)q.rho_a;
    p.Llg_rhoa = (NF)q.Llg * (NF)q.rho_a;
    p.flow = q.flow;
    p.swrc = q.swrc;
    p.unsat_k = q.unsat_k;
    p.seb = q.seb;
    p.halo_policy = q.halo_policy;
    return p;
}

template <class NF> View<NF> make_view(const trm_ctx* c, const FieldSet& s) {
    View<NF> v;
    v.Nh = c->Nh;
    v.Nz = c->Nz;
    v.Nzp = c->Nzp;
    auto F = [&](int id) { return (NF*)s.f[id]; };
    v.U = F(TRM_FIELD_INTERNAL_ENERGY);
    v.sat = F(TRM_FIELD_SATURATION_WATER_ICE);
    v.T = F(TRM_FIELD_TEMPERATURE);
    v.liq = F(TRM_FIELD_LIQUID_WATER_FRACTION);
    v.psi = F(TRM_FIELD_PRESSURE_HEAD);
    v.Kf = F(TRM_FIELD_HYDRAULIC_CONDUCTIVITY);
    v.Kf_top = (NF*)s.kf_top;
    v.top_T = (NF*)c->d_top3;
    v.top_sat = v.top_T ? v.top_T + c->Nh : nullptr;
    v.top_liq = v.top_T ? v.top_T + 2 * c->Nh : nullptr;
    v.G_U = F(TRM_FIELD_TEND_INTERNAL_ENERGY);
    v.G_sat = F(TRM_FIELD_TEND_SATURATION_WATER_ICE);
    v.S = F(TRM_FIELD_SURFACE_EXCESS_WATER);
    v.G_S = F(TRM_FIELD_TEND_SURFACE_EXCESS_WATER);
    v.wt = F(TRM_FIELD_WATER_TABLE);
    v.Ts = F(TRM_FIELD_SKIN_TEMPERATURE);
    v.ghf = F(TRM_FIELD_GROUND_HEAT_FLUX);
    v.swu = F(TRM_FIELD_SURFACE_SHORTWAVE_UP);
    v.lwu = F(TRM_FIELD_SURFACE_LONGWAVE_UP);
    v.rnet = F(TRM_FIELD_SURFACE_NET_RADIATION);
    v.Hs = F(TRM_FIELD_SENSIBLE_HEAT_FLUX);
    v.Hl = F(TRM_FIELD_LATENT_HEAT_FLUX);
    v.evap = F(TRM_FIELD_EVAPORATION_GROUND);
    v.infil = F(TRM_FIELD_INFILTRATION);
    v.runoff = F(TRM_FIELD_SURFACE_RUNOFF);
    v.Tair = F(TRM_FIELD_AIR_TEMPERATURE);
    v.pres = F(TRM_FIELD_AIR_PRESSURE);
    v.wind = F(TRM_FIELD_WINDSPEED);
    v.qair = F(TRM_FIELD_SPECIFIC_HUMIDITY);
    v.rain = F(TRM_FIELD_RAINFALL);
    v.swd = F(TRM_FIELD_SURFACE_SHORTWAVE_DOWN);
    v.lwd = F(TRM_FIELD_SURFACE_LONGWAVE_DOWN);
    v.albedo = F(TRM_FIELD_ALBEDO);
    v.emissivity = F(TRM_FIELD_EMISSIVITY);
    v.zC = (const NF*)c->d_zC;
    v.zF = (const NF*)c->d_zF;
    v.dzc = (const NF*)c->d_dzc;
    v.rdzc = (const NF*)c->d_rdzc;
    v.rdzf = (const NF*)c->d_rdzf;
    v.psiz = (const NF*)c->d_psiz;
    v.lvl = (const NF*)c->d_lvl;
    // (static between the stages unless the caller evaluates a state-dependent forcing at the stage: trm_stage_field_device_ptr)
    v.Fvwc = c->opt_vwc_field ? (const NF*)((&s == &c->stage && c->stage_vwc_own) ? c->stage.f[TRM_FIELD_VWC_FORCING] : c->state.f[TRM_FIELD_VWC_FORCING]) : nullptr;
    BcGeom<NF>& g = v.g;
    g.dzf_bot = (NF)c->dzf_bot;
    g.dzf_top = (NF)c->dzf_top;
    g.hdzf_bot = g.dzf_bot / NF(2);
    g.hdzf_top = g.dzf_top / NF(2);
    g.rhdzf_bot = NF(1) / g.hdzf_bot;
    g.rhdzf_top = NF(1) / g.hdzf_top;
    g.Az = (NF)c->Az;                       // Flat y => Az = dx
    g.V_bot = g.Az * (NF)c->dzc_bot;        // V = Az * dz
    g.V_top = g.Az * (NF)c->dzc_top;
    g.rV_bot = NF(1) / g.V_bot;
    g.rV_top = NF(1) / g.V_top;
    g.zF_top = (NF)c->h_zF[c->Nz];
    g.dzc_top = (NF)c->dzc_top;
    v.status = c->d_status;
    for (int a = 0; a < TRM_BCV_COUNT; ++a)
        for (int b = 0; b < 2; ++b) {
            v.bc.kind[a][b] = c->bc_kind[a][b];
            void* val = (&s == &c->stage && c->bc_value_stage[a][b]) ? c->bc_value_stage[a][b] : c->bc_value[a][b];
            v.bc.value[a][b] = val ? val : c->d_zero;
        }
    fill_small_table(v);
    return v;
}

// The view of columns [lo, lo + n) of `v`: every per-column / per-cell pointer advanced, the per-level tables shared.
// (3-D fields are [Nh][Nzp], 2-D fields and boundary value arrays [Nh]; top_* are three [Nh] rows of one buffer.)
template <class NF> View<NF> sub_view(const View<NF>& v, long lo, long n) {
    View<NF> s = v;
    s.Nh = n;
    const long c3 = lo * v.Nzp;
    auto p3 = [&](NF*& q) { if (q) q += c3; };
    auto p2 = [&](NF*& q) { if (q) q += lo; };
    auto c2 = [&](const NF*& q) { if (q) q += lo; };
    p3(s.U); p3(s.sat); p3(s.T); p3(s.liq); p3(s.psi); p3(s.Kf); p3(s.G_U); p3(s.G_sat);
    if (s.Fvwc) s.Fvwc += c3;
    p2(s.S); p2(s.wt); p2(s.Kf_top); p2(s.G_S);
    p2(s.Ts); p2(s.ghf); p2(s.infil); p2(s.swu); p2(s.lwu); p2(s.rnet); p2(s.Hs); p2(s.Hl); p2(s.evap); p2(s.runoff);
    p2(s.top_T); p2(s.top_sat); p2(s.top_liq);
    c2(s.Tair); c2(s.pres); c2(s.wind); c2(s.qair); c2(s.rain); c2(s.swd); c2(s.lwd); c2(s.albedo); c2(s.emissivity);
    for (int a = 0; a < TRM_BCV_COUNT; ++a)
        for (int b = 0; b < 2; ++b)
            if (s.bc.value[a][b]) s.bc.value[a][b] = (const NF*)s.bc.value[a][b] + lo;
    fill_small_table(s);
    return s;
}

template <class NF> StageView<NF> make_stage_view(const trm_ctx* c) {
    // Heun: the stage's temperature boundary values (a series evaluated at t + dt), else the state's
    StageView<NF> w{};
    auto bc = [&](int side) {
        void* q = c->bc_value_stage[TRM_BCV_TEMPERATURE][side] ? c->bc_value_stage[TRM_BCV_TEMPERATURE][side] : c->bc_value[TRM_BCV_TEMPERATURE][side];
        return (const NF*)(q ? q : c->d_zero);
    };
    w.bcT_bot = bc(0);
    w.bcT_top = bc(1);
    return w;
}

}  // namespace

namespace trmh {
int front_epoch_next(trm_ctx* c) {
    const size_t bytes = (size_t)c->Nh * FRONT_GRANULES * sizeof(unsigned long long);
    if (!c->d_gran) {
        TRM_HIP(c, hipMalloc((void**)&c->d_gran, bytes));
        TRM_HIP(c, hipMemsetAsync(c->d_gran, 0, bytes, c->stream));
        c->front_epoch = 0;
    }
    if (++c->front_epoch == 0) {      // (wrapped: stale granules may carry any tag again -- start over)
        TRM_HIP(c, hipMemsetAsync(c->d_gran, 0, bytes, c->stream));
        c->front_epoch = 1;
    }
    return TRM_OK;
}
template <class NF> const LaunchArgs<NF>& launch_args(trm_ctx* c) {
    if (!c->args) {
        c->args = new LaunchArgs<NF>();
        c->args_free = [](void* q) { delete (LaunchArgs<NF>*)q; };
    }
    LaunchArgs<NF>* a = (LaunchArgs<NF>*)c->args;
    if (!c->args_valid) {
        a->p = make_dev_params<NF>(c->params);
        a->state = make_view<NF>(c, c->state);
        a->stage = make_view<NF>(c, c->stage);
        for (int q = 0; q < 2; ++q) a->part[q] = sub_view<NF>(a->state, c->part_lo[q], c->part_n[q]);
        a->w = make_stage_view<NF>(c);
        c->args_valid = true;
    }
    return *a;
}
template const LaunchArgs<double>& launch_args<double>(trm_ctx*);
template const LaunchArgs<float>& launch_args<float>(trm_ctx*);

// Time interpolation indices of a series at time t -- Oceananigans' FieldTimeSeries indexing (Linear / Clamp /
// Cyclical), restated; that package is not part of the reference tree (parity unpinned).
// Returns 0-based nodes n1, n2 and the fraction f: value = v[n2] * f + v[n1] * (1 - f); n1 == n2 means "copy".
void series_time_indices(const std::vector<double>& times, int indexing, double t, int& n1, int& n2, double& f, double& g) {
    const int nt = (int)times.size();
    n1 = n2 = 0;
    f = g = 0.0;
    if (nt == 1) return;
    if (indexing == TRM_TIME_RASTER) {
        // update_from_raster! (TerrariumRastersExt.jl:96-121): searchsorted brackets t; on a node or beyond either end the
        // node's values are taken as they are (flat extrapolation), in between x1 + eps * (x2 - x1) / dt
        const int right = (int)(std::lower_bound(times.begin(), times.end(), t) - times.begin());       // first(indexes) - 1
        const int left = (int)(std::upper_bound(times.begin(), times.end(), t) - times.begin()) - 1;    // last(indexes) - 1
        if (left >= 0 && right <= nt - 1) {
            n1 = left;
            n2 = right;
            g = times[right] - times[left];
            f = t - times[left];
            if (!(g > 0)) n1 = n2 = right;   // (on a node)
        } else {
            n1 = n2 = std::min(right, nt - 1);
        }
        return;
    }
    auto find = [&](double tq) {
        // binary search for the bracketing interval; an interior node hit returns (n, n); outside the range the
        // first / last interval is returned (linear extrapolation)
        int low = 0, high = nt - 1;
        while (low + 1 < high) {
            int mid = (low + high) / 2;
            if (times[mid] == tq) { n1 = n2 = mid; f = 0.0; return; }
            if (times[mid] < tq) low = mid; else high = mid;
        }
        n1 = low;
        n2 = high;
        f = (1.0 / (times[n2] - times[n1])) * (tq - times[n1]);
    };
    if (indexing == TRM_TIME_CYCLICAL) {
        const double t1 = times[0], tN = times[nt - 1];
        const double period = (tN - t1) + (tN - times[nt - 2]);
        double tau = std::fmod(t - t1, period);
        if (tau < 0) tau += period;
        const double tm = tau + t1;
        if (tm > tN) {   // between the last node and the first node of the next cycle
            n1 = nt - 1;
            n2 = 0;
            f = (1.0 / (period - (tN - t1))) * (tm - tN);
            return;
        }
        find(tm);
        return;
    }
    find(t);
    if (indexing == TRM_TIME_CLAMP) {
        if (t >= times[nt - 1]) { n1 = n2 = nt - 1; f = 0.0; }
        else if (t <= times[0]) { n1 = n2 = 0; f = 0.0; }
    }
}
}  // namespace trmh

namespace {

// (measured: profiles/r04/coupling_exchange.log)
#ifndef TRM_SINGLE_STEP_PROGRAM_MAX_COLUMNS
#define TRM_SINGLE_STEP_PROGRAM_MAX_COLUMNS 0
#endif

template <class NF> int upload_impl(trm_ctx* c, int field, const NF* host);

// The step sequences of one precision.  The launches themselves are Unfused / Veg / ColumnLaunch / GenericLaunch / DeepLaunch /
// LandLaunch / PackedLaunch (trm_host.hpp); the forwarders below keep their reference names in the sequences.
template <class NF> struct Ops {
    using P = Policy<NF>;
    using U = Unfused<NF>;
    static bool richards(const trm_ctx* c) { return P::richards(c); }
    static bool coupled(const trm_ctx* c) { return P::coupled(c); }
    static int hyd(const trm_ctx* c) { return P::hyd(c); }
    static bool generic_bcs(const trm_ctx* c) { return P::generic_bcs(c); }
    static bool packed_path(trm_ctx* c) { return P::packed_path(c); }
    static bool deep_columns(const trm_ctx* c) { return P::deep_columns(c); }
    static bool wide_columns(const trm_ctx* c) { return P::wide_columns(c); }
    static int series_slot(const trm_ctx* c, const trm_ctx::Series& sr) { return P::series_slot(c, sr); }
    static bool series_fit_program(const trm_ctx* c) { return P::series_fit_program(c); }
    static VegDev<NF> veg_dev(const trm_ctx* c) { return P::veg_dev(c); }
    static VegView<NF> veg_view(const trm_ctx* c) { return P::veg_view(c); }
    static VegView<NF> veg_view(const trm_ctx* c, const FieldSet& s) { return P::veg_view(c, s); }
    static int await_levels(trm_ctx* c, trm_ctx::Series& sr, int last_level) { return U::await_levels(c, sr, last_level); }
    static int update_inputs(trm_ctx* c, const FieldSet& s, double time) { return U::update_inputs(c, s, time); }
    static int hydraulics(trm_ctx* c, const FieldSet& s) { return U::hydraulics(c, s); }
    static int surface(trm_ctx* c, const FieldSet& s, bool from_state = false) { return U::surface(c, s, from_state); }
    static int compute_auxiliary(trm_ctx* c, const FieldSet& s) { return U::compute_auxiliary(c, s); }
    static int compute_tendencies(trm_ctx* c, const FieldSet& s) { return U::compute_tendencies(c, s); }
    static int reset_tendencies(trm_ctx* c, const FieldSet& s) { return U::reset_tendencies(c, s); }
    static int update_state(trm_ctx* c, const FieldSet& s, bool tendencies) { return U::update_state(c, s, tendencies); }
    static int explicit_step(trm_ctx* c, const FieldSet& s, double dt) { return U::explicit_step(c, s, dt); }
    static int closure(trm_ctx* c, const FieldSet& s) { return U::closure(c, s); }
    static int invclosure(trm_ctx* c, const FieldSet& s) { return U::invclosure(c, s); }
    static int initialize(trm_ctx* c) { return U::initialize(c); }
    static int average(trm_ctx* c, int field) { return U::average(c, field); }
    template <bool FROM_STATE, bool ADVANCE> static int surface_veg(trm_ctx* c, const FieldSet& s, double dt, bool store_paw = true) {
        return Veg<NF>::surface_veg(c, s, FROM_STATE, ADVANCE, dt, store_paw);
    }
    static int surface_veg_launch(trm_ctx* c, const View<NF>& v, const VegView<NF>& vv, const SurfaceVegArgs<NF>& a) { return Veg<NF>::surface_veg_launch(c, v, vv, a); }
    template <int MODE> static int veg_launch(trm_ctx* c, double dt, int nsteps, int finalize) { return Veg<NF>::vegetation(c, c->state, MODE, dt, nsteps, finalize); }
    static int plant_available_water(trm_ctx* c) { return Veg<NF>::plant_available_water(c, c->state, true); }
    static int plant_available_water(trm_ctx* c, const FieldSet& s, bool store_paw) { return Veg<NF>::plant_available_water(c, s, store_paw); }

    // nsteps steps of the standalone VegetationModel; time series inputs are evaluated by the host between launches
    static int veg_step(trm_ctx* c, double dt, int nsteps, int finalize, bool heun) {
        c->last_program = TRM_PROGRAM_VEGETATION;
        int n = 0;
        while (n < nsteps) {
            const int m = c->series.empty() ? nsteps - n : 1;
            const int fin = (finalize && n + m == nsteps) ? 1 : 0;
            int rc = update_inputs(c, c->state, c->time);
            if (!rc) rc = heun ? veg_launch<VEG_HEUN>(c, dt, m, fin) : veg_launch<VEG_EULER>(c, dt, m, fin);
            if (rc) return rc;
            for (int j = 0; j < m; ++j) c->time += dt;
            c->iteration += m;
            n += m;
        }
        return TRM_OK;
    }
    // static root fractions: density at the cell centres x thickness, normalised over the column (root_distribution.jl:45-63)
    static int upload_root_fraction(trm_ctx* c) {
        const VegDev<NF> p = veg_dev(c);
        std::vector<NF> R((size_t)c->Nz);
        NF total = NF(0);
        for (int k = 0; k < c->Nz; ++k) {
            const NF z = (NF)c->h_zC[k], dz = (NF)c->h_dzc[k];
            R[k] = (NF(0.5) * (p.root_a * std::exp(p.root_a * z) + p.root_b * std::exp(p.root_b * z))) * dz;
        }
        for (int k = 0; k < c->Nz; ++k) total = total + R[k];
        std::vector<NF> per_level((size_t)c->Nz);
        for (int k = 0; k < c->Nz; ++k) per_level[k] = R[k] / total;
        if (!c->d_rootf) TRM_HIP(c, hipMalloc(&c->d_rootf, (size_t)c->Nz * sizeof(NF)));
        TRM_HIP(c, hipMemcpyAsync(c->d_rootf, per_level.data(), (size_t)c->Nz * sizeof(NF), hipMemcpyHostToDevice, c->stream));
    TRM_HIP(c, hipStreamSynchronize(c->stream));
        std::vector<NF> host((size_t)c->Nz * c->Nh);
        for (int k = 0; k < c->Nz; ++k)
            for (long i = 0; i < c->Nh; ++i) host[(size_t)k * c->Nh + i] = R[k] / total;
        return upload_impl<NF>(c, TRM_FIELD_ROOT_FRACTION, host.data());
    }
    // slot table + [nsteps][nseries] rows for a multi-step launch that starts at the context clock
    static int upload_series_rows(trm_ctx* c, double dt, int nsteps) {
        const int ns = (int)c->series.size();
        const size_t nrows = (size_t)nsteps * ns, need = sizeof(SeriesTable<NF>) + nrows * sizeof(SeriesRow);
        trm_ctx::RowStage& st = c->row_stage[c->row_stage_next];
        c->row_stage_next = (c->row_stage_next + 1) % 4;
        if (st.pending) {
            TRM_HIP(c, hipEventSynchronize(st.done));
            st.pending = false;
        }
        if (!st.done) TRM_HIP(c, hipEventCreateWithFlags(&st.done, hipEventDisableTiming));
        if (need > st.cap) {
            if (st.h) TRM_HIP(c, hipHostFree(st.h));
            st.h = nullptr;
            st.cap = 0;
            TRM_HIP(c, hipHostMalloc(&st.h, need, hipHostMallocDefault));
            st.cap = need;
        }
        SeriesTable<NF>& tb = *(SeriesTable<NF>*)st.h;
        SeriesRow* rows = (SeriesRow*)((char*)st.h + sizeof(SeriesTable<NF>));
        std::memset(&tb, 0, sizeof(tb));
        for (int j = 0; j < ns; ++j) {
            auto& sr = c->series[j];
            const int slot = series_slot(c, sr);
            tb.base[slot] = (const NF*)sr.d_values;
            tb.row_of[slot] = j;
            tb.raster[slot] = sr.indexing == TRM_TIME_RASTER ? 1 : 0;
            if (sr.is_bc) {
                void*& dst = c->bc_value[sr.var][sr.side];
                if (!dst) {
                    TRM_HIP(c, hipMalloc(&dst, (size_t)c->Nh * sizeof(NF)));
                    c->args_valid = false;
                }
                tb.dst[slot] = (NF*)dst;
            } else {
                tb.dst[slot] = (NF*)c->state.f[sr.field];
            }
            double t = c->time;
            if (sr.trimmed && t < sr.trimmed_before)
                return fail(c, TRM_EINVAL, "a windowed time series was asked for a time before the levels it still holds (trm_series_trim_before released them)");
            for (int s = 0; s < nsteps; ++s) {
                int n1, n2;
                double f, g;
                series_time_indices(sr.times, sr.indexing, t, n1, n2, f, g);
                if (int rw = await_levels(c, sr, std::max(n1, n2))) return rw;
                rows[(size_t)s * ns + j] = SeriesRow{(long long)(sr.slot(n1) * (size_t)c->Nh), (long long)(sr.slot(n2) * (size_t)c->Nh), f, g};
                t += dt;
            }
        }
        if (!c->d_series_table) TRM_HIP(c, hipMalloc(&c->d_series_table, sizeof(SeriesTable<double>)));
        if (nrows * sizeof(SeriesRow) > c->series_rows_cap) {
            if (c->d_series_rows) TRM_HIP(c, hipFree(c->d_series_rows));   // (waits for the launches that read it)
            c->d_series_rows = nullptr;
            c->series_rows_cap = 0;
            TRM_HIP(c, hipMalloc(&c->d_series_rows, nrows * sizeof(SeriesRow)));
            c->series_rows_cap = nrows * sizeof(SeriesRow);
        }
        // in stream order behind the previous launch (which reads the device copies) and in front of the next one; the host
        // does not wait
        TRM_HIP(c, hipMemcpyAsync(c->d_series_table, &tb, sizeof(tb), hipMemcpyHostToDevice, c->stream));
        TRM_HIP(c, hipMemcpyAsync(c->d_series_rows, rows, nrows * sizeof(SeriesRow), hipMemcpyHostToDevice, c->stream));
        TRM_HIP(c, hipEventRecord(st.done, c->stream));
        st.pending = true;
        return TRM_OK;
    }
    // the top-cell arrays (LandModel: T, sat, liq of the top cell, [Nh] each) can describe the state: they exist and no device
    // pointer to T / sat / liq has been handed out.  Every "the next surface evaluation may read the arrays" decision goes
    // through here -- a launch with TOP_ARRAYS on a context without them would read through a null pointer.
    static bool tops_current(const trm_ctx* c) { return c->d_top3 != nullptr && !c->top_escaped; }
    template <int PROG> static int column_program(trm_ctx* c, double dt, int finalize, int nsteps) {
        return richards(c) ? ColumnLaunch<NF, true, PROG>::run(c, dt, finalize, nsteps) : ColumnLaunch<NF, false, PROG>::run(c, dt, finalize, nsteps);
    }
    // the resident multi-step program on deep columns (no surface energy balance, no series: see step())
    static int deep_program(trm_ctx* c, double dt, int finalize, int nsteps) { return DeepLaunch<NF>::run(c, PROG_MULTI, false, dt, finalize, nsteps); }
    // Heun of deep columns: the sequence of heun_step_fused with k_column_deep<PROG_HEUN> as the column program
    static int heun_step_deep(trm_ctx* c, double dt, int finalize) {
        int rc = update_inputs(c, c->state, c->time);
        if (!rc) rc = update_inputs(c, c->stage, c->time + dt);   // boundary value series at the stage's clock (heun.jl:52)
        if (!rc && c->params.seb) rc = surface(c, c->state, true);
        if (!rc) {
            rc = wide_columns(c) ? WideLaunch<NF>::run(c, PROG_HEUN, generic_bcs(c), dt, finalize) : DeepLaunch<NF>::run(c, PROG_HEUN, generic_bcs(c), dt, finalize, 1);
        }
        if (!rc) c->closure_consistent = true;
        c->tend_valid = finalize != 0;
        c->top_valid = c->params.seb != 0 && !rc && tops_current(c);
        if (!rc && finalize && c->params.seb) rc = surface(c, c->state, true);
        return rc;
    }
    // one fused ForwardEuler step (the state's surface processes have run)
    static int wave_step(trm_ctx* c, double dt, int finalize) {
        int rc = TRM_OK;
        if (wide_columns(c)) {
            rc = WideLaunch<NF>::run(c, PROG_EULER, generic_bcs(c), dt, finalize);
            if (!rc) c->closure_consistent = true;
            return rc;
        }
        if (deep_columns(c)) {
            rc = DeepLaunch<NF>::run(c, PROG_EULER, generic_bcs(c), dt, finalize, 1);
            if (!rc) c->closure_consistent = true;
            return rc;
        }
        if (packed_path(c)) {
            rc = PackedLaunch::step(c, dt, finalize);
        } else if (generic_bcs(c)) {
            rc = GenericLaunch<NF>::step(c, dt, finalize);
        } else {
            rc = column_program<PROG_EULER>(c, dt, finalize, 1);
        }
        if (!rc) c->closure_consistent = true;
        return rc;
    }
    static int unfused_step(trm_ctx* c, double dt, int finalize) {
        c->last_program = TRM_PROGRAM_UNFUSED;
        int rc = update_state(c, c->state, true);
        if (!rc) rc = explicit_step(c, c->state, dt);
        if (!rc) rc = closure(c, c->state);
        if (!rc && finalize) rc = compute_auxiliary(c, c->state);
        return rc;
    }
    // TRM_OPT_STEPS_PER_LAUNCH = 0: run!'s loop (model_integrator.jl:72-88) is exactly trm_step(ctx, dt, nsteps, 0), so the
    // resident-column program is what a plain call gets whenever it is legal.  Measured: 2.0-4.6 us per step
    // against 6.8-12 on N72 / 7 119-column shards, 12.3 against 24-26 at N145 -- and for fp32 contexts whose per-step path is
    // the packed kernel as well: C5 371 against 447-451 us, C5-VG 479 against 489, a 12 696-column fp32 shard 7.3 against 13.3
    // (r3: the rule used to keep the packed kernel there).
    static int auto_steps_per_launch(trm_ctx*) { return 50; }
    // ---- LandModel, per-step path: the surface processes of one half of the columns UNDER the column program of the other ----
    // (k_land_euler / k_land_pk, trm_column.hpp: one stream, two launches per step as before, each covering the soil columns
    // of one half and the 0-D surface processes of the other)
    static bool interleave_now(trm_ctx* c, int steps_left) {
        if (c->opt_pipeline == 0 || steps_left < 2 || !c->params.seb || !richards(c) || coupled(c) || c->part_n[1] <= 0) return false;
        if (!c->series.empty() || generic_bcs(c) || c->Nz > 64) return false;      // (inputs constant over the call; one level per lane)
        if (std::is_same<NF, float>::value && !packed_path(c)) return false;        // (fp32 off the packed kernel: not instantiated)
        // Measured (profiles/r03/exp6_ab_land_interleaved.log, bench_default.json: land_interleaved): within +-2 % at 812 500 columns
        // (492 vs 497, 503 vs 511 us on one box; 523 vs 512 on another), +10 % at N145 and on its shards -- every launch carries ~4 us
        // of fixed cost, and two half-size column launches pay it twice where the k_surface launch they absorb was little more than
        // that fixed cost itself.  Not a win anywhere it was measured: AUTO (2) leaves it off; 1 forces it.
        return c->opt_pipeline == 1;
    }
    // columns of part `qcol` step; the surface processes of part `qsurf` run beside them for ITS next column step
    static int land_launch(trm_ctx* c, int qcol, int qsurf, double dt, int finalize, bool top_arrays) { return LandLaunch<NF>::run(c, qcol, qsurf, dt, finalize, top_arrays); }
    // `nsteps` >= 2 ForwardEuler steps of a bare-ground LandModel with constant inputs:
    //     surf(A, 0) | col(A, 0) + surf(B, 0) | col(B, 0) + surf(A, 1) | ... | col(A, N-1) + surf(B, N-1) | col(B, N-1)
    static int land_steps_interleaved(trm_ctx* c, double dt, int nsteps, int finalize) {
        const bool top0 = c->top_valid, cc0 = c->closure_consistent;
        int rc;
        {   // the state's surface processes for half A
            PartScope scope(c, 0);
            c->top_valid = top0;
            rc = surface(c, c->state, true);
        }
        for (int n = 0; n < nsteps && !rc; ++n) {
            const int fin = (finalize && n == nsteps - 1) ? 1 : 0;
            const bool tops = (n == 0) ? top0 : tops_current(c);     // what a surface evaluation of an UNSTEPPED / stepped half reads
            c->closure_consistent = (n == 0) ? cc0 : true;
            rc = land_launch(c, 0, 1, dt, fin, tops);
            if (rc) break;
            if (n < nsteps - 1) {
                rc = land_launch(c, 1, 0, dt, 0, tops_current(c));   // (half A has just been stepped: its top arrays are current)
            } else {
                PartScope scope(c, 1);
                rc = wave_step(c, dt, fin);
            }
            c->time += dt;
            c->iteration += 1;
        }
        c->closure_consistent = !rc;
        c->tend_valid = finalize != 0;
        c->top_valid = !rc && tops_current(c);
        if (!rc && finalize) rc = surface(c, c->state, true);
        return rc;
    }
    // One fused ForwardEuler step of the columns the launch helpers currently address: update_inputs!, the 0-D surface
    // processes as their own small launch in front of the column kernel (LandModel), and once more after it when finalizing.
    // (+ the 0-D prognostics' step of the coupled vegetation; the per-cell plant_available_water field is materialised with the
    // other per-cell auxiliaries: by the finalizing launch, or every step under TRM_OPT_WRITE_KF_EVERY_STEP)
    // TRM_OPT_SURFACE_IN_LAUNCH: a per-step launch of this context can carry its own surface processes (k_column_land) -- a
    // bare-ground LandModel in fp64 on the branch-free program with the LandModel's boundary wiring, one level per lane, every
    // column in one launch, the top-cell arrays current (the surface workgroups read them).
    static bool surface_in_launch(trm_ctx* c, bool heun = false) {
        if (c->opt_front == 0 || (heun && std::is_same<NF, float>::value)) return false;
        if (!c->params.seb || !richards(c) || coupled(c) || c->Nz > 64 || generic_bcs(c) || c->part >= 0) return false;
        if (c->opt_kernel != TRM_KERNEL_FUSED || !c->opt_bc_signature || bc_signature_of(c) != BCSIG_LAND) return false;
        if (hyd(c) != HYD_BC_LINEAR && hyd(c) != HYD_VG_N2) return false;
        if (!c->top_valid || !tops_current(c)) return false;
        const int d = heun ? DERIVE_NONE : P::template derive_now<true>(c);      // (the Heun program reads T / liq as stored)
        if (std::is_same<NF, float>::value ? !(packed_path(c) && (d == DERIVE_NONE || d == DERIVE_LIQ))         // k_step_pk_land
                                           : !(d == DERIVE_NONE || d == DERIVE_T_LIQ)) return false;          // k_column_land
        if (c->opt_front == 1) return true;
        // The library's rule (2).  What the single launch saves is the FIXED cost of the second launch (~3-4 us); the surface chain
        // itself is still evaluated, and the column waves of the first generation wait for it.  Measured, same box, pair -> one launch
        // (profiles/r05/exp3d_prio_sleep.log, exp4_packed_surface_in_launch.log, exp4b_in_launch_by_size.log): fp64 1 780 columns
        // 9.8 -> 7.1 us, 7 119 (the shard of BASELINE config 4) 11.1 -> 8.6, C4-VG shard 12.3 -> 9.2, 28 476 19.7 -> 19.3, N145
        // (56 951) 30.3 -> 29.6 ... 30.0; fp32 12 696 columns 15.2 -> 10.6, 50 782 29.5 -> 31.2, 203 125 108.6 -> 106.6, C5
        // (812 500) 425.4 -> 426.7, C5-VG 437.1 -> 451.0: a clear win where the step is launch-bound, nothing beyond.
        return c->Nh <= (std::is_same<NF, float>::value ? 32768 : 65536);
    }
    static int fused_step(trm_ctx* c, double dt, int fin) {
        int rc = update_inputs(c, c->state, c->time);
        if (rc) return rc;
        const bool in_launch = surface_in_launch(c);
        if (coupled(c)) rc = surface_veg<true, true>(c, c->state, dt, c->opt_write_kf != 0);
        else if (c->params.seb && !in_launch) rc = surface(c, c->state, true);
        if (!rc && in_launch) {
            rc = std::is_same<NF, float>::value ? PackedLaunch::step_land(c, dt, fin) : FrontLaunch::run(c, dt, fin);
            if (!rc) c->closure_consistent = true;
        } else if (!rc) rc = wave_step(c, dt, fin);
        c->tend_valid = fin != 0;   // only the finalizing launch stores state.tendencies
        c->top_valid = c->params.seb != 0 && !rc && tops_current(c);
        if (!rc && fin && coupled(c)) rc = surface_veg<true, false>(c, c->state, 0.0);
        else if (!rc && fin && c->params.seb) rc = surface(c, c->state, true);
        return rc;
    }
    // TRM_OPT_SINGLE_STEP_PROGRAM: a bare-ground LandModel stepped ONE step per call (its inputs change every step: a coupled
    // atmosphere) takes the resident column program with the surface processes inline -- one launch instead of the
    // k_surface + k_column pair.  At N145 the pair wins by far (the inline surface balance runs on every lane of the column's
    // half-wave: C4 103.9 vs 34.1 us); on a shard of a few thousand columns the step is bound by launch latency and
    // the single launch wins.
    static bool single_step_program(const trm_ctx* c) {
        if (!c->params.seb || c->Nz > 64 || c->opt_single_step == 0) return false;
        if (c->opt_single_step == 1) return true;
        return c->Nh <= TRM_SINGLE_STEP_PROGRAM_MAX_COLUMNS;
    }
    // how many steps ONE launch of trm_step covers for this context: 1 unless the resident multi-step program applies
    static bool program_applies(const trm_ctx* c) {
        const bool fused = c->opt_kernel == TRM_KERNEL_FUSED && (c->Nz <= 64 || deep_columns(c) || wide_columns(c));
        return fused && !generic_bcs(c) && !coupled(c) && c->veg_mode != TRM_VEGETATION_STANDALONE &&
               ((c->Nz <= 64 && series_fit_program(c)) || (deep_columns(c) && !c->params.seb && c->series.empty()));
    }
    static int steps_per_launch_now(trm_ctx* c) {
        return !program_applies(c) ? 1 : (c->opt_steps_per_launch > 0 ? c->opt_steps_per_launch : auto_steps_per_launch(c));
    }
    static int step(trm_ctx* c, double dt, int nsteps, int finalize) {
        if (c->veg_mode == TRM_VEGETATION_STANDALONE) return veg_step(c, dt, nsteps, finalize, false);
        // the fused kernels map one soil level (two for 65 ... 128 levels, branch-free boundary kinds) to one lane; anything
        // deeper takes the reference-order kernels
        // the fused kernels map one soil level to one lane (two for 65 ... 128 levels, four for 129 ... 256); anything deeper takes the
        // reference-order kernels.
        const bool fused = c->opt_kernel == TRM_KERNEL_FUSED && (c->Nz <= 64 || deep_columns(c) || wide_columns(c));
        // Resident-column multi-step program: legal when nothing the host evaluates changes between the steps of a launch --
        // constants, or device-resident time series the program interpolates itself -- and the branch-free boundary kinds apply.
        // (columns of 65 ... 128 levels: contexts without the surface energy balance and without series)
        const bool program_ok = program_applies(c);
        const int spl = steps_per_launch_now(c);
        int n = 0, rc = TRM_OK;
        while (n < nsteps && !rc) {
            int m = std::min(spl, nsteps - n);
            const int fin = (finalize && n + m == nsteps) ? 1 : 0;
            if (!fused) {
                rc = update_inputs(c, c->state, c->time);
                c->top_valid = false;
                c->tend_valid = true;
                if (!rc) rc = unfused_step(c, dt, fin);
                if (!rc) c->closure_consistent = true;   // closure! has just run
            } else if (m > 1 || (program_ok && single_step_program(c))) {
                rc = c->series.empty() ? update_inputs(c, c->state, c->time) : upload_series_rows(c, dt, m);
                if (!rc) rc = deep_columns(c) ? deep_program(c, dt, fin, m) : column_program<PROG_MULTI>(c, dt, fin, m);
                if (!rc) c->closure_consistent = true;
                c->tend_valid = fin != 0;
                c->top_valid = c->params.seb != 0 && !rc && tops_current(c);
                if (!rc && fin && c->params.seb) rc = surface(c, c->state, true);
            } else if (interleave_now(c, nsteps - n)) {
                // every remaining step of the call in one go (the clock is ticked inside)
                m = nsteps - n;
                rc = land_steps_interleaved(c, dt, m, finalize);
                n += m;
                continue;
            } else {
                rc = fused_step(c, dt, fin);
            }
            if (rc) break;
            for (int j = 0; j < m; ++j) c->time += dt;   // tick! per step: the same sequence of sums as per-step calls
            c->iteration += m;
            n += m;
        }
        return rc;
    }

    // ---- Heun (heun.jl:37-71), reference-order kernels on a second copy of the state -----------------
    // copyto!(stage, state) (heun.jl:45) for what the stage's predictor step reads before its own closure! and update_state!
    // overwrite it: every per-column field, and of the per-cell fields the prognostics and their tendencies.  temperature,
    // liquid fraction, pressure head, hydraulic conductivity and plant available water of the stage are outputs of
    // closure!(stage) / compute_auxiliary!(stage) and are never read before that (k_explicit_step reads U, sat, S, Ts, the
    // tendencies and the boundary fluxes only): 4 of 9 per-cell copies instead of all of them.
    // `everything`: the per-cell closure and auxiliary fields too -- the two-call Heun, where the caller's functions may READ the
    // stage's fields before update_state!(stage) has recomputed them and must find what the reference's copyto! left there.
    static int copy_state_to_stage(trm_ctx* c, bool everything = false) {
        for (int f = 0; f < TRM_FIELD_COUNT; ++f) {
            if (!c->state.f[f] || !c->stage.f[f]) continue;
            if (f == TRM_FIELD_VWC_FORCING && c->stage_vwc_own) continue;      // (the caller's stage buffer: refresh_user_stage_buffers)
            const bool needed = everything || !is_3d(f) || f == TRM_FIELD_INTERNAL_ENERGY || f == TRM_FIELD_SATURATION_WATER_ICE ||
                                f == TRM_FIELD_TEND_INTERNAL_ENERGY || f == TRM_FIELD_TEND_SATURATION_WATER_ICE;
            if (needed) TRM_HIP(c, hipMemcpyAsync(c->stage.f[f], c->state.f[f], field_elems(c, f) * sizeof(NF), hipMemcpyDeviceToDevice, c->stream));
        }
        if (everything && c->state.kf_top && c->stage.kf_top)
            TRM_HIP(c, hipMemcpyAsync(c->stage.kf_top, c->state.kf_top, (size_t)c->Nh * sizeof(NF), hipMemcpyDeviceToDevice, c->stream));
        return TRM_OK;
    }
    // Heun in ONE launch (TRM_KERNEL_FUSED, Nz <= 64, branch-free boundary kinds): both stages on the column in registers
    // (k_column<PROG_HEUN>), the stage never touches memory.  The stage's surface energy balance is not evaluated: its
    // fluxes would only enter through compute_z_bcs!, which the reference runs for the state alone (heun.jl:54-69).
    static int heun_step_fused(trm_ctx* c, double dt, int finalize) {
        int rc = update_inputs(c, c->state, c->time);
        if (!rc) rc = update_inputs(c, c->stage, c->time + dt);   // boundary value series at the stage's clock (heun.jl:52)
        const bool in_launch = !rc && surface_in_launch(c, true);      // (k_column_land<..., PROG_HEUN>: the state's surface processes in the launch)
        if (!rc && c->params.seb && !in_launch) rc = surface(c, c->state, true);
        if (!rc) rc = in_launch ? FrontLaunch::run(c, dt, finalize, true) : column_program<PROG_HEUN>(c, dt, finalize, 1);
        if (!rc) c->closure_consistent = true;
        c->tend_valid = finalize != 0;
        c->top_valid = c->params.seb != 0 && !rc && tops_current(c);
        if (!rc && finalize && c->params.seb) rc = surface(c, c->state, true);
        return rc;
    }
    // Heun of the vegetation-coupled LandModel in four launches (heun.jl:37-71 with land_model.jl:79-97).  The soil column's
    // two stages stay in registers (k_column<PROG_HEUN>); what the 0-D processes need AT the stage -- the stage's saturation,
    // liquid fraction, temperature, surface excess water -- is the only part of it that is stored.  The final soil state does not
    // depend on the stage's 0-D evaluation (its boundary fluxes are the state's, heun.jl:64-69), so the order is:
    //   1. k_surface_veg(state): auxiliaries, first-stage 0-D tendencies, PREDICTED 0-D prognostics -> stage arrays
    //   2. k_column<PROG_HEUN>: the soil step; stage soil fields -> stage arrays
    //   3. k_surface_veg(stage): auxiliaries and 0-D tendencies at the stage (inputs evaluated at t + dt)
    //   4. k_heun_average_0d: averaged tendencies, explicit step of canopy water, vegetation carbon, area fraction
    static int heun_step_coupled_fused(trm_ctx* c, double dt, int finalize) {
        int rc = update_inputs(c, c->state, c->time);
        if (!rc) rc = update_inputs(c, c->stage, c->time + dt);
        if (rc) return rc;
        const VegView<NF> vs = veg_view(c, c->state);
        VegView<NF> vg = veg_view(c, c->stage);
        View<NF> sv = cached_view<NF>(c, c->stage);
        // inputs of the stage: its own arrays where a time series feeds them (evaluated at t + dt above), else the state's
        auto fed = [&](int field) {
            for (const auto& sr : c->series) if (!sr.is_bc && sr.field == field) return true;
            return false;
        };
        const View<NF>& v0 = cached_view<NF>(c, c->state);
        if (!fed(TRM_FIELD_AIR_TEMPERATURE)) sv.Tair = v0.Tair;
        if (!fed(TRM_FIELD_AIR_PRESSURE)) sv.pres = v0.pres;
        if (!fed(TRM_FIELD_WINDSPEED)) sv.wind = v0.wind;
        if (!fed(TRM_FIELD_SPECIFIC_HUMIDITY)) sv.qair = v0.qair;
        if (!fed(TRM_FIELD_RAINFALL)) sv.rain = v0.rain;
        if (!fed(TRM_FIELD_SURFACE_SHORTWAVE_DOWN)) sv.swd = v0.swd;
        if (!fed(TRM_FIELD_SURFACE_LONGWAVE_DOWN)) sv.lwd = v0.lwd;
        if (!fed(TRM_FIELD_ALBEDO)) sv.albedo = v0.albedo;
        if (!fed(TRM_FIELD_EMISSIVITY)) sv.emissivity = v0.emissivity;
        vg.Tair = sv.Tair; vg.pres = sv.pres; vg.qair = sv.qair; vg.swd = sv.swd;
        if (!fed(TRM_FIELD_CO2)) vg.CO2 = vs.CO2;
        if (!fed(TRM_FIELD_DAILY_LEAF_RESPIRATION)) vg.daily_Rd = vs.daily_Rd;
        if (!fed(TRM_FIELD_STEM_AREA_INDEX)) vg.SAI = vs.SAI;
        SurfaceVegArgs<NF> a{};
        a.dt = (NF)dt;
        a.richards = richards(c) ? 1 : 0;
        a.from_state = 1;
        a.top_arrays = (c->top_valid && tops_current(c)) ? 1 : 0;
        a.advance = 3;
        a.store_paw = c->opt_write_kf != 0;
        a.st_w_can = vg.w_can; a.st_C_veg = vg.C_veg; a.st_nu = vg.nu; a.st_An = vg.An; a.st_Ts = sv.Ts;
        rc = surface_veg_launch(c, v0, vs, a);
        if (!rc) rc = wide_columns(c) ? WideLaunch<NF>::run(c, PROG_HEUN, false, dt, finalize)
                              : (deep_columns(c) ? DeepLaunch<NF>::run(c, PROG_HEUN, false, dt, finalize, 1) : column_program<PROG_HEUN>(c, dt, finalize, 1));
        if (rc) return rc;
        c->closure_consistent = true;
        c->tend_valid = finalize != 0;
        c->top_valid = tops_current(c);
        SurfaceVegArgs<NF> b{};
        b.dt = (NF)dt;
        b.richards = a.richards;
        b.from_state = 1;
        b.top_arrays = 0;
        b.advance = 2;
        b.store_paw = 0;
        rc = surface_veg_launch(c, sv, vg, b);
        if (rc) return rc;
        rc = Veg<NF>::heun_average_0d(c, vs, vg, dt);
        if (rc) return rc;
        if (finalize) rc = surface_veg<true, false>(c, c->state, 0.0);
        return rc;
    }
    static int heun_step_generic_fused(trm_ctx* c, double dt, int finalize) {
        int rc = update_inputs(c, c->state, c->time);
        if (!rc) rc = update_inputs(c, c->stage, c->time + dt);   // boundary value series at the stage's clock (heun.jl:52)
        if (!rc && c->params.seb) rc = surface(c, c->state, true);
        if (rc) return rc;
        rc = GenericLaunch<NF>::heun(c, dt, finalize);
        if (!rc) c->closure_consistent = true;
        c->tend_valid = finalize != 0;
        c->top_valid = c->params.seb != 0 && !rc && tops_current(c);
        if (!rc && finalize && c->params.seb) rc = surface(c, c->state, true);
        return rc;
    }
    // Stage buffers the caller of the two-call Heun has been handed (trm_stage_bc_device_ptr, the stage's vwc_forcing) hold
    // what it wrote for ITS last stage.  Whenever the library forms a stage on its own they are the state's values again: the
    // stage's predictor runs at the state's clock (heun.jl:47-50).
    static int refresh_user_stage_buffers(trm_ctx* c) {
        for (int a = 0; a < TRM_BCV_COUNT; ++a)
            for (int b = 0; b < 2; ++b)
                if (c->stage_bc_user[a][b] && c->bc_value_stage[a][b] && c->bc_value[a][b])
                    TRM_HIP(c, hipMemcpyAsync(c->bc_value_stage[a][b], c->bc_value[a][b], (size_t)c->Nh * sizeof(NF), hipMemcpyDeviceToDevice, c->stream));
        if (c->stage_vwc_own && c->stage.f[TRM_FIELD_VWC_FORCING])
            TRM_HIP(c, hipMemcpyAsync(c->stage.f[TRM_FIELD_VWC_FORCING], c->state.f[TRM_FIELD_VWC_FORCING], field_elems(c, TRM_FIELD_VWC_FORCING) * sizeof(NF), hipMemcpyDeviceToDevice, c->stream));
        return TRM_OK;
    }
    // heun.jl:41-52: the first half of timestep!(integrator, ::Heun) on the reference-order kernels
    static int heun_predict(trm_ctx* c, double dt, bool copy_everything = false) {
        int rc = update_inputs(c, c->state, c->time);
        if (!rc) rc = update_state(c, c->state, true);
        if (!rc) rc = copy_state_to_stage(c, copy_everything);
        if (!rc) rc = refresh_user_stage_buffers(c);
        if (!rc) rc = update_inputs(c, c->stage, c->time);   // the stage's clock is still t for its predictor step (heun.jl:47-50)
        if (!rc) rc = explicit_step(c, c->stage, dt);
        if (!rc) rc = closure(c, c->stage);
        if (!rc) rc = update_inputs(c, c->stage, c->time + dt);   // the stage's clock has ticked (heun.jl:52)
        return rc;
    }
    // heun.jl:54-71: the second half
    // update_state!(stage) up to and including compute_auxiliary!(stage) (heun.jl:54, state_variables.jl:72-80): what a forcing
    // function evaluated inside compute_tendencies!(stage) finds in `fields`
    static int heun_stage_auxiliary(trm_ctx* c) {
        int rc = reset_tendencies(c, c->stage);
        if (!rc) rc = compute_auxiliary(c, c->stage);
        return rc;
    }
    static int heun_correct(trm_ctx* c, double dt, int finalize, bool stage_auxiliary_done = false) {
        int rc = stage_auxiliary_done ? compute_tendencies(c, c->stage) : update_state(c, c->stage, true);
        if (!rc) rc = average(c, TRM_FIELD_TEND_INTERNAL_ENERGY);
        if (!rc && richards(c)) rc = average(c, TRM_FIELD_TEND_SATURATION_WATER_ICE);
        if (!rc && richards(c)) rc = average(c, TRM_FIELD_TEND_SURFACE_EXCESS_WATER);
        if (coupled(c))
            for (int f : {TRM_FIELD_TEND_CANOPY_WATER, TRM_FIELD_TEND_CARBON_VEGETATION, TRM_FIELD_TEND_VEGETATION_AREA_FRACTION})
                if (!rc) rc = average(c, f);
        if (!rc) rc = explicit_step(c, c->state, dt);
        if (!rc) rc = closure(c, c->state);
        if (!rc && finalize) rc = compute_auxiliary(c, c->state);
        return rc;
    }
    static int heun_step(trm_ctx* c, double dt, int finalize) {
        if (int rr = refresh_user_stage_buffers(c)) return rr;
        if (c->opt_kernel == TRM_KERNEL_FUSED && c->Nz <= 64 && generic_bcs(c) && !coupled(c)) return heun_step_generic_fused(c, dt, finalize);
        if (c->opt_kernel == TRM_KERNEL_FUSED && (c->Nz <= 64 || deep_columns(c) || wide_columns(c)) && !generic_bcs(c) && coupled(c)) return heun_step_coupled_fused(c, dt, finalize);
        if (c->opt_kernel == TRM_KERNEL_FUSED && c->Nz <= 64 && !generic_bcs(c) && !coupled(c)) return heun_step_fused(c, dt, finalize);
        if (c->opt_kernel == TRM_KERNEL_FUSED && (deep_columns(c) || wide_columns(c)) && !coupled(c)) return heun_step_deep(c, dt, finalize);      // (every boundary kind)
        c->top_valid = false;
        c->tend_valid = true;
        c->closure_consistent = true;   // (ends with closure!)
        c->last_program = TRM_PROGRAM_UNFUSED;
        int rc = heun_predict(c, dt);
        if (!rc) rc = heun_correct(c, dt, finalize);
        return rc;
    }
};

#define DISPATCH(c, expr) ((c)->precision == TRM_F64 ? Ops<double>::expr : Ops<float>::expr)

int finish(trm_ctx* c, int rc) {
    if (rc) return rc;
    if (!c->opt_async) TRM_HIP(c, hipStreamSynchronize(c->stream));
    return TRM_OK;
}

// The zero fills run on the CONTEXT stream and are waited for.  A plain hipMemset goes to the null stream, which the context's
// non-blocking stream does not synchronise with: under load (two processes time-slicing one device) a fill could land AFTER a
// later upload / copy on the context stream had written the buffer and wipe it -- seen once as a NaN state in a shared-device
// rehearsal of the N > 1 bench.
// Every field starts (field id mod 32) x 16 640 bytes into its own allocation.  hipMalloc hands out identically aligned blocks, so
// column i of every field of the step (nine streams) would sit at the same offset modulo any power of two -- the same HBM
// channel and bank for all of them at the moment a wave touches them.  65 x 256 B moves each field by an odd number of 256-byte
// interleave units: measured -3.7 ... -4.7 % on the HBM-resident fp64 step (8 x N145: 210.7 -> 200.9 us, 218.1 -> 209.9 on another
// box), -3 % at C5, nothing on the cache-resident ones (profiles/r03/exp15*_skew.log).  TRM_FIELD_SKEW (bytes, a multiple of 256)
// overrides it for experiments.
// Environment switches of experiments and tests are validated once; a value outside its range makes trm_create
// fail with a message instead of silently selecting something else.
bool parse_env_int(const char* name, long lo, long hi, long multiple_of, long& out, std::string& err) {
    const char* e = std::getenv(name);
    if (!e) return false;
    char* end = nullptr;
    const long v = std::strtol(e, &end, 10);
    if (end == e || *end != 0 || v < lo || v > hi || (multiple_of > 1 && v % multiple_of != 0)) {
        err = std::string(name) + "=" + e + ": expected an integer in [" + std::to_string(lo) + ", " + std::to_string(hi) + "]" +
              (multiple_of > 1 ? " that is a multiple of " + std::to_string(multiple_of) : std::string());
        return false;
    }
    out = v;
    return true;
}
struct EnvSwitches {
    long field_skew = 16640, derive_default = -1, debug_placement = 0, handoff_tag_bias = 0;
    std::string error;
    EnvSwitches() {
        long v;
        if (parse_env_int("TRM_FIELD_SKEW", 0, 1 << 20, 256, v, error)) field_skew = v;
        if (error.empty() && parse_env_int("TRM_DERIVE_DEFAULT", 0, 5, 1, v, error)) derive_default = v;
        if (error.empty() && parse_env_int("TRM_DEBUG_PLACEMENT", 0, 1, 1, v, error)) debug_placement = v;   // (prints every field's allocation: profiles/tools/placement_probe.sh)
        if (error.empty() && parse_env_int("TRM_DEBUG_HANDOFF_TAG_BIAS", 0, 1, 1, v, error)) handoff_tag_bias = v;      // (tests: FrontArgs::tag_bias)
        if (error.empty() && parse_env_int("TRM_STAGED_SMALL", 0, 1, 1, v, error)) { /* read by Policy::staged_now */ }
        if (error.empty() && parse_env_int("TRM_SCALAR_INPUTS", 0, 1, 1, v, error)) { /* read by Policy::scalar_inputs_now */ }
    }
};
const EnvSwitches& env_switches() {
    static const EnvSwitches e;
    return e;
}
size_t field_skew_bytes() { return (size_t)env_switches().field_skew; }
int alloc_fields(trm_ctx* c, FieldSet& s) {
    for (int f = 0; f < TRM_FIELD_COUNT; ++f) {
        if (is_lazy_field(f) && c->veg_mode == TRM_VEGETATION_OFF) continue;
        if (s.f[f]) continue;
        size_t bytes = field_elems(c, f) * c->esize;
        const size_t skew = field_skew_bytes() * (size_t)(f % 32);
        TRM_HIP(c, hipMalloc(&s.raw[f], bytes + skew));
        s.f[f] = (char*)s.raw[f] + skew;
        if (env_switches().debug_placement) std::fprintf(stderr, "trm placement: field %d raw %p bytes %zu skew %zu\n", f, s.raw[f], bytes, skew);
        TRM_HIP(c, hipMemsetAsync(s.f[f], 0, bytes, c->stream));
    }
    if (!s.kf_top) {
        TRM_HIP(c, hipMalloc(&s.kf_top, (size_t)c->Nh * c->esize));
        TRM_HIP(c, hipMemsetAsync(s.kf_top, 0, (size_t)c->Nh * c->esize, c->stream));
    }
    TRM_HIP(c, hipStreamSynchronize(c->stream));
    return TRM_OK;
}

template <class NF> int upload_grid(trm_ctx* c, const double* thickness) {
    HostGrid<NF> g;
    g.build(c->Nz, thickness);
    auto up = [&](void** d, const std::vector<NF>& h) -> int {
        TRM_HIP(c, hipMalloc(d, h.size() * sizeof(NF)));
        TRM_HIP(c, hipMemcpyAsync(*d, h.data(), h.size() * sizeof(NF), hipMemcpyHostToDevice, c->stream));
    TRM_HIP(c, hipStreamSynchronize(c->stream));
        return TRM_OK;
    };
    int rc;
    if ((rc = up(&c->d_zC, g.zC))) return rc;
    if ((rc = up(&c->d_zF, g.zF))) return rc;
    if ((rc = up(&c->d_dzc, g.dzc))) return rc;
    if ((rc = up(&c->d_rdzc, g.rdzc))) return rc;
    if ((rc = up(&c->d_rdzf, g.rdzf))) return rc;
    if ((rc = up(&c->d_psiz, g.psiz))) return rc;
    {   // per-level records for the lane = level kernels
        std::vector<NF> lvl((size_t)c->Nz * 8, NF(0));
        for (int k = 0; k < c->Nz; ++k) {
            NF* q = &lvl[(size_t)k * 8];
            q[0] = g.zC[k]; q[1] = g.psiz[k]; q[2] = g.zF[k]; q[3] = g.dzc[k]; q[4] = g.rdzc[k]; q[5] = g.rdzf[k]; q[6] = g.rdzf[k + 1];
        }
        if ((rc = up(&c->d_lvl, lvl))) return rc;
    }
    c->h_zF.assign(g.zF.begin(), g.zF.end());
    c->h_zC.assign(g.zC.begin(), g.zC.end());
    c->h_dzc.assign(g.dzc.begin(), g.dzc.end());
    c->h_dzf.assign(g.dzf.begin(), g.dzf.end());
    c->dzf_bot = g.dzf_bot;
    c->dzf_top = g.dzf_top;
    c->dzc_bot = g.dzc[0];
    c->dzc_top = g.dzc[c->Nz - 1];
    return TRM_OK;
}

template <class NF> int fill_row(trm_ctx* c, int field, double value) {
    std::vector<NF> h((size_t)c->Nh, (NF)value);
    TRM_HIP(c, hipMemcpyAsync(c->state.f[field], h.data(), h.size() * sizeof(NF), hipMemcpyHostToDevice, c->stream));
    TRM_HIP(c, hipStreamSynchronize(c->stream));
    return TRM_OK;
}

// ---- reductions ------------------------------------------------------------------------------------
// minimum / maximum as Julia's Base.minimum / maximum: a NaN anywhere gives NaN
__host__ __device__ inline double nan_min(double a, double b) { return (a != a || b != b) ? (a + b) : (a < b ? a : b); }
__host__ __device__ inline double nan_max(double a, double b) { return (a != a || b != b) ? (a + b) : (a > b ? a : b); }
// element (row r, column i) of a field sits at base[r] + i * stride[r]
template <class NF, int OP> __global__ void k_reduce_rows(const NF* f3, const NF* ftop, long Nh, int Nzp, int Nz, int is3d,
                                                          const NF* weight, double* partial) {
    const int row = blockIdx.y;
    const NF* base = is3d ? ((row == Nz && ftop) ? ftop : f3 + row) : f3;
    const long stride = (is3d && !(row == Nz && ftop)) ? Nzp : 1;
    double acc = (OP == TRM_REDUCE_MIN) ? HUGE_VAL : (OP == TRM_REDUCE_MAX ? -HUGE_VAL : 0.0);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < Nh; i += (long)gridDim.x * blockDim.x) {
        double x = (double)base[i * stride];
        if (OP == TRM_REDUCE_SUM) acc += x;
        else if (OP == TRM_REDUCE_VOLUME_INTEGRAL_Z) acc += x * (double)weight[row];
        else if (OP == TRM_REDUCE_MIN) acc = nan_min(acc, x);
        else if (OP == TRM_REDUCE_MAX) acc = nan_max(acc, x);
        else acc += (x != x) ? 1.0 : 0.0;
    }
    __shared__ double sm[4];
    for (int off = 32; off > 0; off >>= 1) {
        double o = __shfl_down(acc, off, 64);
        if (OP == TRM_REDUCE_MIN) acc = nan_min(acc, o);
        else if (OP == TRM_REDUCE_MAX) acc = nan_max(acc, o);
        else acc += o;
    }
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = sm[0];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) {
            if (OP == TRM_REDUCE_MIN) r = nan_min(r, sm[w]);
            else if (OP == TRM_REDUCE_MAX) r = nan_max(r, sm[w]);
            else r += sm[w];
        }
        partial[(long)row * gridDim.x + blockIdx.x] = r;
    }
}

template <class NF> int reduce_impl(trm_ctx* c, int field, int op, double* out) {
    const long rows = field_rows(c, field);
    const int nblocks = (int)std::min<long>(256, (c->Nh + 255) / 256);
    size_t need = (size_t)rows * nblocks;
    if (need > c->reduce_cap) {
        if (c->d_reduce) TRM_HIP(c, hipFree(c->d_reduce));
        TRM_HIP(c, hipMalloc((void**)&c->d_reduce, need * sizeof(double)));
        c->reduce_cap = need;
    }
    const NF* f = (const NF*)c->state.f[field];
    const NF* ftop = field == TRM_FIELD_HYDRAULIC_CONDUCTIVITY ? (const NF*)c->state.kf_top : nullptr;
    const NF* w = (const NF*)c->d_dzc;
    const int d3 = is_3d(field) ? 1 : 0;
    dim3 grid(nblocks, (unsigned)rows);
#define TRM_REDUCE_LAUNCH(OP) hipLaunchKernelGGL((k_reduce_rows<NF, OP>), grid, dim3(256), 0, c->stream, f, ftop, c->Nh, c->Nzp, c->Nz, d3, w, c->d_reduce)
    switch (op) {
        case TRM_REDUCE_SUM: TRM_REDUCE_LAUNCH(TRM_REDUCE_SUM); break;
        case TRM_REDUCE_MIN: TRM_REDUCE_LAUNCH(TRM_REDUCE_MIN); break;
        case TRM_REDUCE_MAX: TRM_REDUCE_LAUNCH(TRM_REDUCE_MAX); break;
        case TRM_REDUCE_HASNAN: TRM_REDUCE_LAUNCH(TRM_REDUCE_HASNAN); break;
        case TRM_REDUCE_VOLUME_INTEGRAL_Z:
            if (rows != c->Nz) return fail(c, TRM_EINVAL, "VOLUME_INTEGRAL_Z needs a cell-centred 3-D field");
            TRM_REDUCE_LAUNCH(TRM_REDUCE_VOLUME_INTEGRAL_Z);
            break;
        default: return fail(c, TRM_EINVAL, "unknown reduction op");
    }
#undef TRM_REDUCE_LAUNCH
    TRM_HIP(c, hipGetLastError());
    std::vector<double> part(need);
    TRM_HIP(c, hipMemcpyAsync(part.data(), c->d_reduce, need * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    TRM_HIP(c, hipStreamSynchronize(c->stream));
    // fixed-order fold of the block partials: deterministic for a given shard size
    double total = 0.0;
    for (long r = 0; r < rows; ++r) {
        double acc = (op == TRM_REDUCE_MIN) ? HUGE_VAL : (op == TRM_REDUCE_MAX ? -HUGE_VAL : 0.0);
        for (int b = 0; b < nblocks; ++b) {
            double x = part[(size_t)r * nblocks + b];
            if (op == TRM_REDUCE_MIN) acc = nan_min(acc, x);
            else if (op == TRM_REDUCE_MAX) acc = nan_max(acc, x);
            else acc += x;
        }
        if (op == TRM_REDUCE_VOLUME_INTEGRAL_Z) total += acc;
        else if (op == TRM_REDUCE_HASNAN) out[r] = acc > 0.0 ? 1.0 : 0.0;
        else out[r] = acc;
    }
    if (op == TRM_REDUCE_VOLUME_INTEGRAL_Z) out[0] = total;
    return TRM_OK;
}

// Host arrays are the reference's interior layout [rows][Nh] (column fastest); the device keeps 3-D fields
// z-fastest [Nh][Nzp].  The transposition runs on the device (32 x 32 tiles through LDS, both sides coalesced): the host
// array crosses PCIe once, as it is, through a staging buffer the context keeps.
template <class NF, bool TO_DEVICE_LAYOUT>
__global__ void __launch_bounds__(256) k_transpose(const NF* __restrict__ src, NF* __restrict__ dst, long Nh, int Nz, int Nzp) {
    __shared__ NF tile[32][33];
    const long i0 = (long)blockIdx.x * 32;
    const int k0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    if (TO_DEVICE_LAYOUT) {   // src [Nz][Nh] -> dst [Nh][Nzp]; levels Nz..Nzp-1 are padding (0)
        for (int r = ty; r < 32; r += 8) {
            const int k = k0 + r;
            const long i = i0 + tx;
            tile[r][tx] = (k < Nz && i < Nh) ? src[(size_t)k * Nh + i] : NF(0);
        }
        __syncthreads();
        for (int r = ty; r < 32; r += 8) {
            const long i = i0 + r;
            const int k = k0 + tx;
            if (i < Nh && k < Nzp) dst[(size_t)i * Nzp + k] = tile[tx][r];
        }
    } else {                  // src [Nh][Nzp] -> dst [Nz][Nh]
        for (int r = ty; r < 32; r += 8) {
            const long i = i0 + r;
            const int k = k0 + tx;
            tile[r][tx] = (i < Nh && k < Nzp) ? src[(size_t)i * Nzp + k] : NF(0);
        }
        __syncthreads();
        for (int r = ty; r < 32; r += 8) {
            const int k = k0 + r;
            const long i = i0 + tx;
            if (k < Nz && i < Nh) dst[(size_t)k * Nh + i] = tile[tx][r];
        }
    }
}

// Rows [row0, row0 + nrows) of a 3-D field in the host layout [nrows][Nh] (32 x 32 tiles through LDS, as k_transpose): the
// output path of a snapshot writer that wants `ground_temperature` -- one row -- does not move the whole field.
template <class NF>
__global__ void __launch_bounds__(256) k_rows_to_host_layout(const NF* __restrict__ src, NF* __restrict__ dst, long Nh, int Nzp, int row0, int nrows) {
    __shared__ NF tile[32][33];
    const long i0 = (long)blockIdx.x * 32;
    const int r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const long i = i0 + r;
        const int k = row0 + r0 + tx;
        tile[r][tx] = (i < Nh && r0 + tx < nrows && k < Nzp) ? src[(size_t)i * Nzp + k] : NF(0);
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int q = r0 + r;
        const long i = i0 + tx;
        if (q < nrows && i < Nh) dst[(size_t)q * Nh + i] = tile[tx][r];
    }
}
// RingGrids.Field(field, grid; fill_value) (column_ring_grid.jl:102-115): rows [nrows][Nh] of the columns -> [nrows][P] on the
// full ring grid, `fill` outside the mask.  One thread per grid point: coalesced writes, reads in column order.
template <class NF> __global__ void k_scatter_ring(const NF* __restrict__ rows, NF* __restrict__ out, const int32_t* __restrict__ inv, long P, long Nh, NF fill) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const int32_t col = inv[p];
    const size_t r = blockIdx.y;
    out[r * (size_t)P + p] = col >= 0 ? rows[r * (size_t)Nh + col] : fill;
}
// Oceananigans.Field(ring_field, grid) (column_ring_grid.jl:117-149): the masked points of [nrows][P] in ring order
template <class NF> __global__ void k_gather_ring(const NF* __restrict__ full, NF* __restrict__ rows, const int32_t* __restrict__ idx, long P, long Nh) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Nh) return;
    const size_t r = blockIdx.y;
    rows[r * (size_t)Nh + i] = full[r * (size_t)P + idx[i]];
}
int io_buffer(trm_ctx* c, size_t bytes) {
    if (bytes > c->io_cap) {
        if (c->d_io) TRM_HIP(c, hipFree(c->d_io));
        c->d_io = nullptr;
        c->io_cap = 0;
        TRM_HIP(c, hipMalloc(&c->d_io, bytes));
        c->io_cap = bytes;
    }
    return TRM_OK;
}
template <class NF> int upload_impl(trm_ctx* c, int field, const NF* host) {
    const long Nh = c->Nh, rows = field_rows(c, field);
    if (!is_3d(field)) {
        TRM_HIP(c, hipMemcpyAsync(c->state.f[field], host, (size_t)Nh * sizeof(NF), hipMemcpyHostToDevice, c->stream));
        TRM_HIP(c, hipStreamSynchronize(c->stream));
        return TRM_OK;
    }
    int rc = io_buffer(c, (size_t)rows * Nh * sizeof(NF));
    if (rc) return rc;
    TRM_HIP(c, hipMemcpyAsync(c->d_io, host, (size_t)rows * Nh * sizeof(NF), hipMemcpyHostToDevice, c->stream));
    dim3 grid((unsigned)((Nh + 31) / 32), (unsigned)((c->Nzp + 31) / 32));
    hipLaunchKernelGGL((k_transpose<NF, true>), grid, dim3(256), 0, c->stream, (const NF*)c->d_io, (NF*)c->state.f[field], Nh, c->Nz, c->Nzp);
    TRM_HIP(c, hipGetLastError());
    if (rows == c->Nz + 1)  // Face field: the top face lives in its own [Nh] buffer
        TRM_HIP(c, hipMemcpyAsync(c->state.kf_top, (const NF*)c->d_io + (size_t)c->Nz * Nh, (size_t)Nh * sizeof(NF), hipMemcpyDeviceToDevice, c->stream));
    TRM_HIP(c, hipStreamSynchronize(c->stream));
    return TRM_OK;
}
template <class NF> int download_impl(trm_ctx* c, int field, NF* host) {
    const long Nh = c->Nh, rows = field_rows(c, field);
    if (!is_3d(field)) {
        TRM_HIP(c, hipMemcpyAsync(host, c->state.f[field], (size_t)Nh * sizeof(NF), hipMemcpyDeviceToHost, c->stream));
        TRM_HIP(c, hipStreamSynchronize(c->stream));
        return TRM_OK;
    }
    int rc = io_buffer(c, (size_t)rows * Nh * sizeof(NF));
    if (rc) return rc;
    dim3 grid((unsigned)((Nh + 31) / 32), (unsigned)((c->Nzp + 31) / 32));
    hipLaunchKernelGGL((k_transpose<NF, false>), grid, dim3(256), 0, c->stream, (const NF*)c->state.f[field], (NF*)c->d_io, Nh, c->Nz, c->Nzp);
    TRM_HIP(c, hipGetLastError());
    if (rows == c->Nz + 1)
        TRM_HIP(c, hipMemcpyAsync((NF*)c->d_io + (size_t)c->Nz * Nh, c->state.kf_top, (size_t)Nh * sizeof(NF), hipMemcpyDeviceToDevice, c->stream));
    TRM_HIP(c, hipMemcpyAsync(host, c->d_io, (size_t)rows * Nh * sizeof(NF), hipMemcpyDeviceToHost, c->stream));
    TRM_HIP(c, hipStreamSynchronize(c->stream));
    return TRM_OK;
}


// rows [row0, row0 + nrows) of a field into the staging buffer d_io as [nrows][Nh] (on the context stream, not synchronised)
template <class NF> int stage_rows(trm_ctx* c, int field, int row0, int nrows) {
    const long Nh = c->Nh;
    int rc = io_buffer(c, (size_t)nrows * Nh * sizeof(NF));
    if (rc) return rc;
    NF* io = (NF*)c->d_io;
    if (!is_3d(field)) {
        TRM_HIP(c, hipMemcpyAsync(io, c->state.f[field], (size_t)Nh * sizeof(NF), hipMemcpyDeviceToDevice, c->stream));
        return TRM_OK;
    }
    const int cell_rows = std::min(nrows, c->Nz - row0);     // (the Face field's top face lives in its own [Nh] buffer)
    if (cell_rows > 0) {
        dim3 grid((unsigned)((Nh + 31) / 32), (unsigned)((cell_rows + 31) / 32));
        hipLaunchKernelGGL((k_rows_to_host_layout<NF>), grid, dim3(256), 0, c->stream, (const NF*)c->state.f[field], io, Nh, c->Nzp, row0, cell_rows);
        TRM_HIP(c, hipGetLastError());
    }
    if (row0 + nrows == c->Nz + 1)
        TRM_HIP(c, hipMemcpyAsync(io + (size_t)(c->Nz - row0) * Nh, c->state.kf_top, (size_t)Nh * sizeof(NF), hipMemcpyDeviceToDevice, c->stream));
    return TRM_OK;
}
int ring_buffer(trm_ctx* c, size_t bytes) {
    if (bytes > c->ring_cap) {
        if (c->d_ring) TRM_HIP(c, hipFree(c->d_ring));
        c->d_ring = nullptr;
        c->ring_cap = 0;
        TRM_HIP(c, hipMalloc(&c->d_ring, bytes));
        c->ring_cap = bytes;
    }
    return TRM_OK;
}
template <class NF> int scatter_ring_impl(trm_ctx* c, int field, int row0, int nrows, double fill, void* out, bool out_is_device) {
    int rc = stage_rows<NF>(c, field, row0, nrows);
    if (rc) return rc;
    NF* dst = (NF*)out;
    if (!out_is_device) {
        if ((rc = ring_buffer(c, (size_t)nrows * c->ring_points * sizeof(NF)))) return rc;
        dst = (NF*)c->d_ring;
    }
    dim3 grid((unsigned)((c->ring_points + 255) / 256), (unsigned)nrows);
    hipLaunchKernelGGL((k_scatter_ring<NF>), grid, dim3(256), 0, c->stream, (const NF*)c->d_io, dst, c->d_ring_inv, c->ring_points, c->Nh, (NF)fill);
    TRM_HIP(c, hipGetLastError());
    if (!out_is_device) TRM_HIP(c, hipMemcpyAsync(out, dst, (size_t)nrows * c->ring_points * sizeof(NF), hipMemcpyDeviceToHost, c->stream));
    TRM_HIP(c, hipStreamSynchronize(c->stream));
    return TRM_OK;
}
// full-grid host (or device) array [rows][P] -> the field: gather on the device, then the usual layout change
template <class NF> int gather_ring_impl(trm_ctx* c, int field, const void* full, bool full_is_device) {
    const long Nh = c->Nh, rows = field_rows(c, field), P = c->ring_points;
    const NF* src = (const NF*)full;
    int rc;
    if (!full_is_device) {
        if ((rc = ring_buffer(c, (size_t)rows * P * sizeof(NF)))) return rc;
        TRM_HIP(c, hipMemcpyAsync(c->d_ring, full, (size_t)rows * P * sizeof(NF), hipMemcpyHostToDevice, c->stream));
        src = (const NF*)c->d_ring;
    }
    if ((rc = io_buffer(c, (size_t)rows * Nh * sizeof(NF)))) return rc;
    NF* io = (NF*)c->d_io;
    dim3 grid((unsigned)((Nh + 255) / 256), (unsigned)rows);
    hipLaunchKernelGGL((k_gather_ring<NF>), grid, dim3(256), 0, c->stream, src, is_3d(field) ? io : (NF*)c->state.f[field], c->d_ring_idx, P, Nh);
    TRM_HIP(c, hipGetLastError());
    if (is_3d(field)) {
        dim3 tg((unsigned)((Nh + 31) / 32), (unsigned)((c->Nzp + 31) / 32));
        hipLaunchKernelGGL((k_transpose<NF, true>), tg, dim3(256), 0, c->stream, (const NF*)io, (NF*)c->state.f[field], Nh, c->Nz, c->Nzp);
        TRM_HIP(c, hipGetLastError());
        if (rows == c->Nz + 1)
            TRM_HIP(c, hipMemcpyAsync(c->state.kf_top, io + (size_t)c->Nz * Nh, (size_t)Nh * sizeof(NF), hipMemcpyDeviceToDevice, c->stream));
    }
    TRM_HIP(c, hipStreamSynchronize(c->stream));
    return TRM_OK;
}
// every H2D copy of trm_series_append has finished (before a series buffer is freed or reallocated)
int sync_copies(trm_ctx* c) {
    if (c->copy_stream && c->copy_pending) TRM_HIP(c, hipStreamSynchronize(c->copy_stream));
    c->copy_pending = false;
    for (auto& o : c->series) o.pending_from = -1;
    return TRM_OK;
}

// ---- RCCL, opened on first use ------------------------------------------------------------------------
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};
Rccl* rccl() {
    static Rccl r;
    if (r.handle || !r.error.empty()) return &r;
    // TRM_RCCL_LIBRARY (tests): the collective library to open instead -- tests/fake_rccl.c, a stand-in that implements the seven
    // entry points through host memory for n communicators in ONE process, so that the grouped call sequences below run with
    // n > 1 on a one-GPU box (with torch in the process its own librccl.so is already mapped: dlsym on OUR handle still
    // resolves to the library named here)
    if (const char* over = std::getenv("TRM_RCCL_LIBRARY")) {
        r.handle = dlopen(over, RTLD_NOW | RTLD_LOCAL);
    } else {
        for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
    }
    if (!r.handle) {
        r.error = std::string("librccl.so could not be loaded: ") + dlerror();
        return &r;
    }
    auto sym = [&](const char* n) { void* p = dlsym(r.handle, n); if (!p && r.error.empty()) r.error = std::string("librccl.so lacks ") + n; return p; };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    return &r;
}
#define TRM_NCCL(ctx, call)                                                                                   \
    do {                                                                                                      \
        ncclResult_t e__ = (call);                                                                            \
        if (e__ != ncclSuccess) return fail(ctx, TRM_ECOMM, std::string(#call) + ": " + rccl()->GetErrorString(e__)); \
    } while (0)

// one all-reduce of `n` doubles over the context's communicator, on the side stream, after the context stream's work
int comm_allreduce(trm_ctx* c, double* host, int n, ncclRedOp_t op) {
    if (!c->comm) return fail(c, TRM_EINVAL, "no communicator: call trm_comm_init first");
    TRM_HIP(c, hipMemcpyAsync(c->d_comm, host, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->comm_stream));
    TRM_NCCL(c, rccl()->AllReduce(c->d_comm, c->d_comm + n, (size_t)n, ncclDouble, op, c->comm, c->comm_stream));
    TRM_HIP(c, hipMemcpyAsync(host, c->d_comm + n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->comm_stream));
    TRM_HIP(c, hipStreamSynchronize(c->comm_stream));
    return TRM_OK;
}

}  // namespace

// ======================================================================================================
// C ABI
// ======================================================================================================
extern "C" {

// TRM_ENTER: any entry point but the three of the two-call Heun step and the read-only ones.  A stage predicted by
// trm_heun_predict belongs to the state and clock it was predicted from: whatever else runs in between drops it, and a later
// trm_heun_correct fails ("call trm_heun_predict first") instead of averaging tendencies of a stage that describes another state.
#define TRM_ENTER_HEUN(c)                                    \
    if (!(c)) return TRM_EINVAL;                             \
    TRM_HIP(c, hipSetDevice((c)->device));
#define TRM_ENTER(c)                                         \
    TRM_ENTER_HEUN(c)                                        \
    (c)->heun_pending = false;

int trm_abi_version(void) { return TRM_ABI_VERSION; }

int trm_default_params(trm_params* p) {
    if (!p) return TRM_EINVAL;
    std::memset(p, 0, sizeof(*p));
    p->rho_w = 1000.0; p->rho_i = 916.2; p->rho_a = 1.293; p->c_a = 1005.7; p->Lsl = 3.34e5; p->Llg = 2.257e6;
    p->Lsg = 2.834e6; p->g = 9.80665; p->Tref = 273.15; p->sigma = 5.6704e-8; p->kappa_vk = 0.4; p->eps_mw = 0.622;
    p->R_a = 287.058;
    p->k_water = 0.57; p->k_ice = 2.2; p->k_air = 0.025; p->k_mineral = 3.8; p->k_organic = 0.25;
    p->c_water = 4.2e6; p->c_ice = 1.9e6; p->c_air = 0.00125e6; p->c_mineral = 2.0e6; p->c_organic = 2.5e6;
    p->por_mineral = 0.49; p->por_organic = 0.9; p->rho_soc = 0.0; p->rho_org = 1300.0;
    p->K_sat = 1.0e-5; p->theta_res = 0.0; p->bc_psi_s = 0.01; p->bc_lambda = 0.2; p->vg_alpha = 1.0; p->vg_n = 2.0;
    p->impedance = 7.0; p->vwc_forcing = 0.0;
    p->albedo = 0.3; p->emissivity = 0.97; p->kappa_s = 2.0; p->C_h = 1.2e-3; p->min_windspeed = 0.01; p->tau_r = 3600.0;
    p->beta_evap = 1.0;
    p->field_capacity = 0.25;
    p->flow = TRM_FLOW_NOFLOW; p->swrc = TRM_SWRC_BROOKS_COREY; p->unsat_k = TRM_UNSATK_LINEAR; p->seb = 0;
    p->halo_policy = TRM_HALO_REFERENCE_ZERO;
    return TRM_OK;
}

const char* trm_last_error(const trm_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int trm_create(const trm_grid* g, const trm_params* p, trm_ctx** out) {
    if (!g || !p || !out) return fail(nullptr, TRM_EINVAL, "trm_create: null argument");
    if (g->precision != TRM_F64 && g->precision != TRM_F32) return fail(nullptr, TRM_EINVAL, "trm_create: precision");
    if (g->num_layers < 2) return fail(nullptr, TRM_EINVAL, "trm_create: num_layers must be >= 2");
    if (g->num_columns < 1 || !g->thickness) return fail(nullptr, TRM_EINVAL, "trm_create: num_columns / thickness");
    for (int k = 0; k < g->num_layers; ++k)
        if (!(g->thickness[k] > 0)) return fail(nullptr, TRM_EINVAL, "trm_create: layer thickness must be positive");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(nullptr, TRM_EHIP, "trm_create: no HIP device available (this library has no CPU fallback)");
    if (g->device < 0 || g->device >= ndev) return fail(nullptr, TRM_EINVAL, "trm_create: device ordinal out of range");
    {
        long nzp = g->num_layers <= 32 ? 32 : (g->num_layers <= 64 ? 64 : ((g->num_layers + 31) / 32) * 32);
        const double esz = g->precision == TRM_F64 ? 8.0 : 4.0;
        if ((double)g->num_columns * (double)nzp * esz >= 4294967296.0)
            return fail(nullptr, TRM_EINVAL, "trm_create: one field (num_columns * level pitch * word size) must stay below 4 GiB per "
                                             "device (32-bit byte offsets in the step kernel); shard the columns over more contexts");
    }
    if (!env_switches().error.empty()) return fail(nullptr, TRM_EINVAL, "trm_create: " + env_switches().error);
    trm_ctx* c = new trm_ctx();
    c->precision = g->precision;
    c->esize = g->precision == TRM_F64 ? 8 : 4;
    c->Nh = (long)g->num_columns;
    c->Nz = g->num_layers;
    c->Nzp = c->Nz <= 32 ? 32 : (c->Nz <= 64 ? 64 : ((c->Nz + 31) / 32) * 32);
    c->device = g->device;
    c->params = *p;
    c->Az = g->dx > 0 ? g->dx : 1.0 / (double)c->Nh;
    if (c->precision == TRM_F32) c->Az = (double)(float)c->Az;
    // pipeline parts: two blocks of columns, the seam on a multiple of 64 (whole workgroups of every step kernel)
    c->part_n[0] = std::min<long>(c->Nh, ((c->Nh / 2 + 63) / 64) * 64);
    c->part_lo[1] = c->part_n[0];
    c->part_n[1] = c->Nh - c->part_n[0];
    auto bail = [&](int rc) {
        g_create_error = c->err;
        trm_destroy(c);
        return rc;
    };
    int rc = TRM_OK;
    auto hip = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && rc == TRM_OK) { c->err = std::string(what) + ": " + hipGetErrorString(e); rc = TRM_EHIP; }
    };
    hip(hipSetDevice(c->device), "hipSetDevice");
    hip(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking), "hipStreamCreate");
    c->stream = c->own_stream;
    hip(hipEventCreate(&c->ev0), "hipEventCreate");
    hip(hipEventCreate(&c->ev1), "hipEventCreate");
    hip(hipMalloc((void**)&c->d_status, sizeof(uint32_t)), "hipMalloc(status)");
    hip(hipMalloc(&c->d_zero, (size_t)c->Nh * c->esize), "hipMalloc(zero)");
    if (c->params.seb) hip(hipMalloc(&c->d_top3, 3 * (size_t)c->Nh * c->esize), "hipMalloc(top cells)");
    // (tests: TRM_DERIVE_DEFAULT = 1 makes small grids take the instances with the derivation, where the staged outputs and the
    // input paths are compiled in -- the value a context starts with for TRM_OPT_DERIVE_CLOSURE_FIELDS)
    if (env_switches().derive_default >= 0) c->opt_derive = (int)env_switches().derive_default;
    c->debug_handoff_tag_bias = (unsigned)env_switches().handoff_tag_bias;
    if (rc == TRM_OK) hip(hipMemsetAsync(c->d_zero, 0, (size_t)c->Nh * c->esize, c->stream), "hipMemset(zero)");
    if (rc) return bail(rc);
    hip(hipMemsetAsync(c->d_status, 0, sizeof(uint32_t), c->stream), "hipMemset(status)");
    hip(hipStreamSynchronize(c->stream), "hipStreamSynchronize");
    if ((rc = alloc_fields(c, c->state))) return bail(rc);
    rc = c->precision == TRM_F64 ? upload_grid<double>(c, g->thickness) : upload_grid<float>(c, g->thickness);
    if (rc) return bail(rc);
    // input defaults (prescribed_atmosphere.jl:90-92,148,221-223)
    const struct { int f; double v; } defaults[] = {
        {TRM_FIELD_AIR_TEMPERATURE, 10.0}, {TRM_FIELD_AIR_PRESSURE, 101325.0}, {TRM_FIELD_WINDSPEED, 0.1},
        {TRM_FIELD_SPECIFIC_HUMIDITY, 1.0e-3}, {TRM_FIELD_RAINFALL, 0.0}, {TRM_FIELD_SURFACE_SHORTWAVE_DOWN, 300.0},
        {TRM_FIELD_SURFACE_LONGWAVE_DOWN, 50.0}};
    for (auto& d : defaults) {
        rc = c->precision == TRM_F64 ? fill_row<double>(c, d.f, d.v) : fill_row<float>(c, d.f, d.v);
        if (rc) return bail(rc);
    }
    *out = c;
    return TRM_OK;
}

int trm_destroy(trm_ctx* c) {
    if (!c) return TRM_OK;
    (void)hipSetDevice(c->device);
    (void)trm_comm_destroy(c);
    if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    for (int f = 0; f < TRM_FIELD_COUNT; ++f) {
        if (c->state.f[f]) (void)hipFree(c->state.raw[f]);
        if (c->stage.f[f]) (void)hipFree(c->stage.raw[f]);
        if (c->saved.f[f]) (void)hipFree(c->saved.raw[f]);
    }
    if (c->d_gran) (void)hipFree(c->d_gran);
    if (c->state.kf_top) (void)hipFree(c->state.kf_top);
    if (c->stage.kf_top) (void)hipFree(c->stage.kf_top);
    if (c->saved.kf_top) (void)hipFree(c->saved.kf_top);
    for (int a = 0; a < TRM_BCV_COUNT; ++a)
        for (int b = 0; b < 2; ++b)
        {
            if (c->bc_value[a][b]) (void)hipFree(c->bc_value[a][b]);
            if (c->bc_value_stage[a][b]) (void)hipFree(c->bc_value_stage[a][b]);
        }
    for (auto& sr : c->series)
        if (sr.d_values) (void)hipFree(sr.d_values);
    for (void* q : {c->d_zC, c->d_zF, c->d_dzc, c->d_rdzc, c->d_rdzf, c->d_psiz, c->d_lvl, c->d_rootf, c->d_zero, c->d_top3, (void*)c->d_status, (void*)c->d_reduce, c->d_io, c->d_series_table, c->d_series_rows})
        if (q) (void)hipFree(q);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    if (c->copy_done) (void)hipEventDestroy(c->copy_done);
    if (c->copy_order) (void)hipEventDestroy(c->copy_order);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    for (auto& st : c->row_stage) {
        if (st.pending) (void)hipEventSynchronize(st.done);
        if (st.done) (void)hipEventDestroy(st.done);
        if (st.h) (void)hipHostFree(st.h);
    }
    for (void* q : {(void*)c->d_ring_inv, (void*)c->d_ring_idx, c->d_ring})
        if (q) (void)hipFree(q);

    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    if (c->args && c->args_free) c->args_free(c->args);
    delete c;
    return TRM_OK;
}

int trm_field_rows(const trm_ctx* c, int field, int64_t* rows) {
    if (!c || !rows || !valid_field(field)) return TRM_EINVAL;
    *rows = field_rows(c, field);
    return TRM_OK;
}

int trm_get_grid(const trm_ctx* c, double* z_faces, double* z_centers, double* dz_center, double* dz_face) {
    if (!c) return TRM_EINVAL;
    if (z_faces) std::memcpy(z_faces, c->h_zF.data(), sizeof(double) * (c->Nz + 1));
    if (z_centers) std::memcpy(z_centers, c->h_zC.data(), sizeof(double) * c->Nz);
    if (dz_center) std::memcpy(dz_center, c->h_dzc.data(), sizeof(double) * c->Nz);
    if (dz_face) std::memcpy(dz_face, c->h_dzf.data(), sizeof(double) * (c->Nz + 1));
    return TRM_OK;
}

int trm_upload(trm_ctx* c, int field, const void* host) {
    if (!c || !host || !valid_field(field)) return fail(c, TRM_EINVAL, "trm_upload: bad argument");
    if (!c->state.f[field]) return fail(c, TRM_EINVAL, "trm_upload: the field exists only after trm_set_vegetation");
    if (field == TRM_FIELD_ROOT_FRACTION)
        return fail(c, TRM_EINVAL, "trm_upload: root_fraction is a static function of the root distribution parameters "
                                   "(root_distribution.jl:45-63): set them with trm_set_vegetation");
    TRM_HIP(c, hipSetDevice(c->device));
    c->heun_pending = false;
    int rc = c->precision == TRM_F64 ? upload_impl<double>(c, field, (const double*)host) : upload_impl<float>(c, field, (const float*)host);
    if (field <= TRM_FIELD_PRESSURE_HEAD || field == TRM_FIELD_WATER_TABLE) c->closure_consistent = false;   // (U, sat, T, liq, psi, water table)
    if (!rc && field == TRM_FIELD_VWC_FORCING) {
        c->opt_vwc_field = 1;
        c->args_valid = false;
    }
    c->top_valid = false;
    if (!rc && (field == TRM_FIELD_TEND_INTERNAL_ENERGY || field == TRM_FIELD_TEND_SATURATION_WATER_ICE || field == TRM_FIELD_TEND_SURFACE_EXCESS_WATER))
        c->tend_valid = true;   // (a caller that sets one tendency field owns all of them)
    return rc;
}

static bool is_tendency(int f) {
    return f == TRM_FIELD_TEND_INTERNAL_ENERGY || f == TRM_FIELD_TEND_SATURATION_WATER_ICE || f == TRM_FIELD_TEND_SURFACE_EXCESS_WATER;
}
static const char* kStaleTendencies =
    "the tendency fields are not materialised by a fused step that does not finalize (finalize = 0): call trm_step / "
    "trm_step_heun with finalize = 1, or trm_update_state(ctx, 1), before reading them";

int trm_download(trm_ctx* c, int field, void* host) {
    if (!c || !host || !valid_field(field)) return fail(c, TRM_EINVAL, "trm_download: bad argument");
    if (is_tendency(field) && !c->tend_valid) return fail(c, TRM_ESTALE, kStaleTendencies);
    if (!c->state.f[field]) return fail(c, TRM_EINVAL, "trm_download: the field exists only after trm_set_vegetation");
    TRM_HIP(c, hipSetDevice(c->device));
    return c->precision == TRM_F64 ? download_impl<double>(c, field, (double*)host) : download_impl<float>(c, field, (float*)host);
}

int trm_field_device_ptr(trm_ctx* c, int field, void** dev, int64_t* pitch_elems) {
    if (!c || !dev || !valid_field(field)) return fail(c, TRM_EINVAL, "trm_field_device_ptr: bad argument");
    if (is_tendency(field) && !c->tend_valid) return fail(c, TRM_ESTALE, kStaleTendencies);
    *dev = c->state.f[field];
    if (pitch_elems) *pitch_elems = is_3d(field) ? c->Nzp : 1;
    // the caller may write the state behind the library's back from now on: stop trusting the top-cell copies
    if (field == TRM_FIELD_TEMPERATURE || field == TRM_FIELD_SATURATION_WATER_ICE || field == TRM_FIELD_LIQUID_WATER_FRACTION)
        c->top_escaped = true;
    if (field <= TRM_FIELD_PRESSURE_HEAD || field == TRM_FIELD_WATER_TABLE) c->closure_escaped = true;   // U, sat, T, liq, psi, the water table may change behind the library's back
    c->top_valid = false;
    return TRM_OK;
}

int trm_bc_device_ptr(trm_ctx* c, int var, int side, void** dev) {
    if (!c || !dev || var < 0 || var >= TRM_BCV_COUNT || (side != TRM_TOP && side != TRM_BOTTOM)) return fail(c, TRM_EINVAL, "trm_bc_device_ptr: bad argument");
    if (c->bc_kind[var][side] == TRM_BC_NOFLUX || !c->bc_value[var][side]) return fail(c, TRM_EINVAL, "trm_bc_device_ptr: the condition carries no values (set it with trm_set_bc first)");
    for (const auto& sr : c->series)
        if (sr.is_bc && sr.var == var && sr.side == side) return fail(c, TRM_EINVAL, "trm_bc_device_ptr: the boundary values are evaluated from a time series every step");
    *dev = c->bc_value[var][side];
    c->bc_zero_gradient[var][side] = false;      // (the caller may write the buffer from now on)
    return TRM_OK;
}

int trm_set_bc(trm_ctx* c, int var, int side, int kind, const void* values, double scalar) {
    if (!c || var < 0 || var >= TRM_BCV_COUNT || (side != TRM_TOP && side != TRM_BOTTOM) || kind < 0 || kind > TRM_BC_GRADIENT)
        return fail(c, TRM_EINVAL, "trm_set_bc: bad argument");
    TRM_HIP(c, hipSetDevice(c->device));
    TRM_HIP(c, hipStreamSynchronize(c->stream));
    if (int rs = sync_copies(c)) return rs;
    // a constant replaces an earlier time series of the same boundary value (which would otherwise overwrite it at the next step)
    for (size_t n = 0; n < c->series.size(); ++n) {
        auto& o = c->series[n];
        if (o.is_bc && o.var == var && o.side == side) {
            if (o.d_values) (void)hipFree(o.d_values);
            c->series.erase(c->series.begin() + (long)n);
            break;
        }
    }
    c->bc_kind[var][side] = kind;
    c->bc_zero_gradient[var][side] = false;
    c->args_valid = false;
    c->heun_pending = false;
    if (kind == TRM_BC_NOFLUX) return TRM_OK;
    size_t bytes = (size_t)c->Nh * c->esize;
    if (!c->bc_value[var][side]) TRM_HIP(c, hipMalloc(&c->bc_value[var][side], bytes));
    std::vector<unsigned char> h(bytes, 0);
    if (values) {
        std::memcpy(h.data(), values, (size_t)c->Nh * c->esize);
    } else if (c->precision == TRM_F64) {
        double* d = (double*)h.data();
        for (long i = 0; i < c->Nh; ++i) d[i] = scalar;
    } else {
        float* d = (float*)h.data();
        for (long i = 0; i < c->Nh; ++i) d[i] = (float)scalar;
    }
    TRM_HIP(c, hipMemcpyAsync(c->bc_value[var][side], h.data(), bytes, hipMemcpyHostToDevice, c->stream));
    TRM_HIP(c, hipStreamSynchronize(c->stream));
    if (kind == TRM_BC_GRADIENT) {   // every value +0 (all bits clear): see trm_ctx::bc_zero_gradient
        bool zero = true;
        for (size_t n = 0; n < bytes && zero; ++n) zero = h[n] == 0;
        c->bc_zero_gradient[var][side] = zero;
    }
    return TRM_OK;
}

int trm_set_forcing(trm_ctx* c, int input_field, const void* per_column) {
    if (!c || !is_input_field(input_field))
        return fail(c, TRM_EINVAL, "trm_set_forcing: not an input field");
    return trm_upload(c, input_field, per_column);
}

int trm_set_forcing_device(trm_ctx* c, int input_field, const void* dev_per_column) {
    TRM_ENTER(c);
    if (!is_input_field(input_field) || !dev_per_column) return fail(c, TRM_EINVAL, "trm_set_forcing_device: not an input field / null pointer");
    if (!c->state.f[input_field]) return fail(c, TRM_EINVAL, "trm_set_forcing_device: the field exists only after trm_set_vegetation");
    for (const auto& sr : c->series)
        if (!sr.is_bc && sr.field == input_field) return fail(c, TRM_EINVAL, "trm_set_forcing_device: the input is evaluated from a time series every step (trm_clear_series / trm_set_forcing first)");
    // stream-ordered: behind the steps already enqueued, in front of the next one; the host does not wait
    TRM_HIP(c, hipMemcpyAsync(c->state.f[input_field], dev_per_column, (size_t)c->Nh * c->esize, hipMemcpyDeviceToDevice, c->stream));
    return TRM_OK;
}

namespace {
int add_series(trm_ctx* c, trm_ctx::Series&& sr, int nt, const double* times, const void* values, const char* who) {
    if (nt < 1 || !times || !values) return fail(c, TRM_EINVAL, std::string(who) + ": nt >= 1, times and values are required");
    for (int n = 1; n < nt; ++n)
        if (!(times[n] > times[n - 1])) return fail(c, TRM_EINVAL, std::string(who) + ": times must be strictly increasing");
    if (sr.indexing < TRM_TIME_LINEAR || sr.indexing > TRM_TIME_RASTER) return fail(c, TRM_EINVAL, std::string(who) + ": bad time_indexing");
    TRM_HIP(c, hipSetDevice(c->device));
    TRM_HIP(c, hipStreamSynchronize(c->stream));
    if (int rs = sync_copies(c)) return rs;
    // replace an earlier series with the same target
    for (size_t n = 0; n < c->series.size(); ++n) {
        auto& o = c->series[n];
        if (o.is_bc == sr.is_bc && (sr.is_bc ? (o.var == sr.var && o.side == sr.side) : o.field == sr.field)) {
            if (o.d_values) (void)hipFree(o.d_values);
            c->series.erase(c->series.begin() + (long)n);
            break;
        }
    }
    sr.times.assign(times, times + nt);
    sr.cap = nt;
    sr.head = 0;
    size_t bytes = (size_t)nt * (size_t)c->Nh * c->esize;
    TRM_HIP(c, hipMalloc(&sr.d_values, bytes));
    TRM_HIP(c, hipMemcpyAsync(sr.d_values, values, bytes, hipMemcpyHostToDevice, c->stream));
    TRM_HIP(c, hipStreamSynchronize(c->stream));
    c->series.push_back(std::move(sr));
    return TRM_OK;
}
}  // namespace

int trm_set_forcing_series(trm_ctx* c, int input_field, int nt, const double* times, const void* values, int time_indexing) {
    if (!c || !is_input_field(input_field))
        return fail(c, TRM_EINVAL, "trm_set_forcing_series: not an input field");
    trm_ctx::Series sr;
    sr.is_bc = false;
    sr.field = input_field;
    sr.indexing = time_indexing;
    return add_series(c, std::move(sr), nt, times, values, "trm_set_forcing_series");
}

int trm_set_bc_series(trm_ctx* c, int var, int side, int kind, int nt, const double* times, const void* values, int time_indexing) {
    if (!c || var < 0 || var >= TRM_BCV_COUNT || (side != TRM_TOP && side != TRM_BOTTOM) || kind <= TRM_BC_NOFLUX || kind > TRM_BC_GRADIENT)
        return fail(c, TRM_EINVAL, "trm_set_bc_series: bad argument");
    trm_ctx::Series sr;
    sr.is_bc = true;
    sr.var = var;
    sr.side = side;
    sr.indexing = time_indexing;
    int rc = add_series(c, std::move(sr), nt, times, values, "trm_set_bc_series");
    if (rc) return rc;
    c->bc_kind[var][side] = kind;
    c->bc_zero_gradient[var][side] = false;      // (the values come from the series from now on)
    c->args_valid = false;
    if (!c->bc_value[var][side]) {
        TRM_HIP(c, hipMalloc(&c->bc_value[var][side], (size_t)c->Nh * c->esize));
        TRM_HIP(c, hipMemsetAsync(c->bc_value[var][side], 0, (size_t)c->Nh * c->esize, c->stream));
        TRM_HIP(c, hipStreamSynchronize(c->stream));
    }
    return TRM_OK;
}

int trm_clear_series(trm_ctx* c) {
    if (!c) return TRM_EINVAL;
    TRM_HIP(c, hipSetDevice(c->device));
    TRM_HIP(c, hipStreamSynchronize(c->stream));
    if (int rs = sync_copies(c)) return rs;
    for (auto& sr : c->series)
        if (sr.d_values) (void)hipFree(sr.d_values);
    c->series.clear();
    return TRM_OK;
}

// ---- windowed time series ("device-side double-buffered forcing slabs") -------------------------------------
namespace {
trm_ctx::Series* find_series(trm_ctx* c, int is_bc, int id, int side) {
    for (auto& sr : c->series)
        if (sr.is_bc == (is_bc != 0) && (is_bc ? (sr.var == id && sr.side == side) : sr.field == id)) return &sr;
    return nullptr;
}
}  // namespace

int trm_series_append(trm_ctx* c, int is_bc, int id, int side, int nt, const double* times, const void* values) {
    TRM_ENTER(c);
    trm_ctx::Series* sr = find_series(c, is_bc, id, side);
    if (!sr) return fail(c, TRM_EINVAL, "trm_series_append: no such series (create it with trm_set_forcing_series / trm_set_bc_series first)");
    if (nt < 1 || !times || !values) return fail(c, TRM_EINVAL, "trm_series_append: nt >= 1, times and values are required");
    if (sr->indexing == TRM_TIME_CYCLICAL) return fail(c, TRM_EINVAL, "trm_series_append: a cyclical series is periodic over its whole record and cannot be windowed");
    if (!(times[0] > sr->times.back())) return fail(c, TRM_EINVAL, "trm_series_append: times must continue the series (strictly increasing)");
    for (int n = 1; n < nt; ++n)
        if (!(times[n] > times[n - 1])) return fail(c, TRM_EINVAL, "trm_series_append: times must be strictly increasing");
    const size_t row = (size_t)c->Nh * c->esize;
    const long held = (long)sr->times.size();
    if (!c->copy_stream) {
        TRM_HIP(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        TRM_HIP(c, hipEventCreateWithFlags(&c->copy_done, hipEventDisableTiming));
    }
    if (held + nt > sr->cap) {
        // no free slots left (nothing was trimmed): the ring grows; the levels held are laid out from slot 0 again
        TRM_HIP(c, hipStreamSynchronize(c->stream));
        if (int rs = sync_copies(c)) return rs;
        const long cap = held + nt;
        void* grown = nullptr;
        TRM_HIP(c, hipMalloc(&grown, (size_t)cap * row));
        for (long n = 0; n < held; ++n)
            TRM_HIP(c, hipMemcpyAsync((char*)grown + (size_t)n * row, (char*)sr->d_values + sr->slot((int)n) * row, row, hipMemcpyDeviceToDevice, c->stream));
        TRM_HIP(c, hipStreamSynchronize(c->stream));
        TRM_HIP(c, hipFree(sr->d_values));
        sr->d_values = grown;
        sr->cap = cap;
        sr->head = 0;
    }
    // host -> pinned staging (the caller's array is borrowed for the duration of the call only) -> device, on the side stream:
    // the copy runs under whatever the context stream is executing; the next step that evaluates the series waits for it
    if (c->copy_pending) TRM_HIP(c, hipEventSynchronize(c->copy_done));   // the staging buffer is free again
    if ((size_t)nt * row > c->h_stage_cap) {
        if (c->h_stage) TRM_HIP(c, hipHostFree(c->h_stage));
        c->h_stage = nullptr;
        c->h_stage_cap = 0;
        TRM_HIP(c, hipHostMalloc(&c->h_stage, (size_t)nt * row, hipHostMallocDefault));
        c->h_stage_cap = (size_t)nt * row;
    }
    std::memcpy(c->h_stage, values, (size_t)nt * row);
    // the slots written are free ones: never used, or released by trm_series_trim_before -- whose event marks the context
    // stream's work that may still have been reading them; steps enqueued after the trim cannot see them
    if (c->order_recorded) TRM_HIP(c, hipStreamWaitEvent(c->copy_stream, c->copy_order, 0));
    for (int n = 0; n < nt; ++n)
        TRM_HIP(c, hipMemcpyAsync((char*)sr->d_values + sr->slot((int)held + n) * row, (char*)c->h_stage + (size_t)n * row, row, hipMemcpyHostToDevice, c->copy_stream));
    TRM_HIP(c, hipEventRecord(c->copy_done, c->copy_stream));
    c->copy_pending = true;
    if (sr->pending_from < 0) sr->pending_from = held;
    sr->windowed = true;
    sr->times.insert(sr->times.end(), times, times + nt);
    return TRM_OK;
}

int trm_series_window(trm_ctx* c, int is_bc, int id, int side, int levels) {
    TRM_ENTER(c);
    trm_ctx::Series* sr = find_series(c, is_bc, id, side);
    if (!sr) return fail(c, TRM_EINVAL, "trm_series_window: no such series (create it with trm_set_forcing_series / trm_set_bc_series first)");
    if (sr->indexing == TRM_TIME_CYCLICAL) return fail(c, TRM_EINVAL, "trm_series_window: a cyclical series is periodic over its whole record and cannot be windowed");
    if (levels < 0) return fail(c, TRM_EINVAL, "trm_series_window: levels < 0");
    sr->windowed = true;
    const long held = (long)sr->times.size();
    if (levels > sr->cap) {      // reserve: the levels held are laid out from slot 0 of the larger ring
        const size_t row = (size_t)c->Nh * c->esize;
        TRM_HIP(c, hipStreamSynchronize(c->stream));
        if (int rs = sync_copies(c)) return rs;
        void* grown = nullptr;
        TRM_HIP(c, hipMalloc(&grown, (size_t)levels * row));
        for (long n = 0; n < held; ++n)
            TRM_HIP(c, hipMemcpyAsync((char*)grown + (size_t)n * row, (char*)sr->d_values + sr->slot((int)n) * row, row, hipMemcpyDeviceToDevice, c->stream));
        TRM_HIP(c, hipStreamSynchronize(c->stream));
        TRM_HIP(c, hipFree(sr->d_values));
        sr->d_values = grown;
        sr->cap = levels;
        sr->head = 0;
    }
    return TRM_OK;
}

int trm_series_trim_before(trm_ctx* c, double t) {
    if (!c) return TRM_EINVAL;
    bool released = false;
    for (auto& sr : c->series) {
        // only series that stream through a window (levels have been appended): a fully resident series that merely lives in
        // the same context keeps its whole record, whatever the clock does later
        if (sr.indexing == TRM_TIME_CYCLICAL || !sr.windowed) continue;
        // keep the node at or before t (the lower bracket of every later evaluation) and at least two levels
        long drop = 0;
        while ((long)sr.times.size() - drop > 2 && sr.times[(size_t)drop + 1] <= t) ++drop;
        if (drop > 0) {
            sr.times.erase(sr.times.begin(), sr.times.begin() + drop);
            sr.head = (sr.head + drop) % sr.cap;
            if (sr.pending_from >= 0) sr.pending_from = std::max<long>(0, sr.pending_from - drop);
            sr.trimmed = true;
            sr.trimmed_before = sr.times.front();
            released = true;
        }
    }
    if (released) {
        if (hipSetDevice(c->device) != hipSuccess) return fail(c, TRM_EHIP, "trm_series_trim_before: hipSetDevice");
        if (!c->copy_order) TRM_HIP(c, hipEventCreateWithFlags(&c->copy_order, hipEventDisableTiming));
        TRM_HIP(c, hipEventRecord(c->copy_order, c->stream));
        c->order_recorded = true;
    }
    return TRM_OK;
}

int trm_series_info(const trm_ctx* c, int is_bc, int id, int side, int64_t* levels_held, int64_t* capacity, double* t_first, double* t_last) {
    if (!c) return TRM_EINVAL;
    const trm_ctx::Series* sr = find_series(const_cast<trm_ctx*>(c), is_bc, id, side);
    if (!sr) return TRM_EINVAL;
    if (levels_held) *levels_held = (int64_t)sr->times.size();
    if (capacity) *capacity = sr->cap;
    if (t_first) *t_first = sr->times.front();
    if (t_last) *t_last = sr->times.back();
    return TRM_OK;
}

// ---- reset!(state) + reset!(clock) (state_variables.jl:102-120, model_integrator.jl:96-99) -----------------------------------
int trm_reset(trm_ctx* c) {
    TRM_ENTER(c);
    TRM_HIP(c, hipStreamSynchronize(c->stream));
    for (int f = 0; f < TRM_FIELD_COUNT; ++f) {
        // prognostic, auxiliary and tendency fields go to zero; inputs keep their values (they are re-initialised from their
        // sources), and so do the static root fractions and the user's per-cell forcing
        if (!c->state.f[f] || is_input_field(f) || f == TRM_FIELD_ROOT_FRACTION || f == TRM_FIELD_VWC_FORCING) continue;
        TRM_HIP(c, hipMemsetAsync(c->state.f[f], 0, field_elems(c, f) * c->esize, c->stream));
    }
    TRM_HIP(c, hipMemsetAsync(c->state.kf_top, 0, (size_t)c->Nh * c->esize, c->stream));
    if (c->d_top3) TRM_HIP(c, hipMemsetAsync(c->d_top3, 0, 3 * (size_t)c->Nh * c->esize, c->stream));
    TRM_HIP(c, hipMemsetAsync(c->d_status, 0, sizeof(uint32_t), c->stream));
    c->time = 0.0;
    c->iteration = 0;
    c->top_valid = false;
    c->tend_valid = true;
    c->closure_consistent = false;
    return finish(c, TRM_OK);
}

// ---- output: rows of a field, and the full ring grid (column_ring_grid.jl:102-149) -----------------------------------------
int trm_download_rows(trm_ctx* c, int field, int row0, int nrows, void* host) {
    TRM_ENTER_HEUN(c);
    if (!host || !valid_field(field) || !c->state.f[field]) return fail(c, TRM_EINVAL, "trm_download_rows: bad argument");
    if (row0 < 0 || nrows < 1 || row0 + nrows > field_rows(c, field)) return fail(c, TRM_EINVAL, "trm_download_rows: rows out of range");
    if (is_tendency(field) && !c->tend_valid) return fail(c, TRM_ESTALE, kStaleTendencies);
    int rc = c->precision == TRM_F64 ? stage_rows<double>(c, field, row0, nrows) : stage_rows<float>(c, field, row0, nrows);
    if (rc) return rc;
    TRM_HIP(c, hipMemcpyAsync(host, c->d_io, (size_t)nrows * c->Nh * c->esize, hipMemcpyDeviceToHost, c->stream));
    TRM_HIP(c, hipStreamSynchronize(c->stream));
    return TRM_OK;
}

int trm_set_ring_grid(trm_ctx* c, int64_t num_points, const int64_t* mask_index) {
    TRM_ENTER_HEUN(c);
    if (num_points < c->Nh || num_points >= ((int64_t)1 << 31) || !mask_index) return fail(c, TRM_EINVAL, "trm_set_ring_grid: bad argument");
    std::vector<int32_t> inv((size_t)num_points, -1), idx((size_t)c->Nh);
    for (long i = 0; i < c->Nh; ++i) {
        const int64_t p = mask_index[i];
        if (p < 0 || p >= num_points || inv[(size_t)p] >= 0 || (i > 0 && p <= mask_index[i - 1]))
            return fail(c, TRM_EINVAL, "trm_set_ring_grid: mask_index must be strictly increasing grid point indices, one per column");
        inv[(size_t)p] = (int32_t)i;
        idx[(size_t)i] = (int32_t)p;
    }
    TRM_HIP(c, hipStreamSynchronize(c->stream));
    for (void* q : {(void*)c->d_ring_inv, (void*)c->d_ring_idx})
        if (q) (void)hipFree(q);
    c->d_ring_inv = c->d_ring_idx = nullptr;
    TRM_HIP(c, hipMalloc((void**)&c->d_ring_inv, inv.size() * sizeof(int32_t)));
    TRM_HIP(c, hipMalloc((void**)&c->d_ring_idx, idx.size() * sizeof(int32_t)));
    TRM_HIP(c, hipMemcpyAsync(c->d_ring_inv, inv.data(), inv.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    TRM_HIP(c, hipStreamSynchronize(c->stream));
    TRM_HIP(c, hipMemcpyAsync(c->d_ring_idx, idx.data(), idx.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    TRM_HIP(c, hipStreamSynchronize(c->stream));
    c->ring_points = (long)num_points;
    return TRM_OK;
}

static int ring_args_ok(trm_ctx* c, int field, int row0, int nrows, const void* buf, const char* who) {
    if (!c->ring_points) return fail(c, TRM_EINVAL, std::string(who) + ": call trm_set_ring_grid first");
    if (!buf || !valid_field(field) || !c->state.f[field]) return fail(c, TRM_EINVAL, std::string(who) + ": bad argument");
    if (row0 < 0 || nrows < 1 || row0 + nrows > field_rows(c, field)) return fail(c, TRM_EINVAL, std::string(who) + ": rows out of range");
    if (is_tendency(field) && !c->tend_valid) return fail(c, TRM_ESTALE, kStaleTendencies);
    return TRM_OK;
}
int trm_download_ring(trm_ctx* c, int field, int row0, int nrows, double fill, void* host) {
    TRM_ENTER_HEUN(c);
    if (int rc = ring_args_ok(c, field, row0, nrows, host, "trm_download_ring")) return rc;
    return c->precision == TRM_F64 ? scatter_ring_impl<double>(c, field, row0, nrows, fill, host, false) : scatter_ring_impl<float>(c, field, row0, nrows, fill, host, false);
}
int trm_scatter_ring_device(trm_ctx* c, int field, int row0, int nrows, double fill, void* dev_out) {
    TRM_ENTER_HEUN(c);
    if (int rc = ring_args_ok(c, field, row0, nrows, dev_out, "trm_scatter_ring_device")) return rc;
    return c->precision == TRM_F64 ? scatter_ring_impl<double>(c, field, row0, nrows, fill, dev_out, true) : scatter_ring_impl<float>(c, field, row0, nrows, fill, dev_out, true);
}
static int gather_ring(trm_ctx* c, int field, const void* full, bool device, const char* who) {
    if (int rc = ring_args_ok(c, field, 0, (int)field_rows(c, field), full, who)) return rc;
    if (field == TRM_FIELD_ROOT_FRACTION) return fail(c, TRM_EINVAL, std::string(who) + ": root_fraction is derived from the root distribution parameters");
    int rc = c->precision == TRM_F64 ? gather_ring_impl<double>(c, field, full, device) : gather_ring_impl<float>(c, field, full, device);
    if (field <= TRM_FIELD_PRESSURE_HEAD || field == TRM_FIELD_WATER_TABLE) c->closure_consistent = false;
    if (!rc && field == TRM_FIELD_VWC_FORCING) { c->opt_vwc_field = 1; c->args_valid = false; }
    c->top_valid = false;
    if (!rc && is_tendency(field)) c->tend_valid = true;
    return rc;
}
int trm_upload_ring(trm_ctx* c, int field, const void* host_full) {
    TRM_ENTER(c);
    return gather_ring(c, field, host_full, false, "trm_upload_ring");
}
int trm_gather_ring_device(trm_ctx* c, int field, const void* dev_full) {
    TRM_ENTER(c);
    return gather_ring(c, field, dev_full, true, "trm_gather_ring_device");
}

int trm_initialize(trm_ctx* c) {
    TRM_ENTER(c);
    c->top_valid = false;
    c->closure_consistent = false;   // temperature is the user's, internal_energy follows from it
    return finish(c, DISPATCH(c, initialize(c)));
}
int trm_update_inputs(trm_ctx* c) {
    TRM_ENTER(c);
    return finish(c, DISPATCH(c, update_inputs(c, c->state, c->time)));
}
int trm_update_state(trm_ctx* c, int compute_tendencies) {
    TRM_ENTER(c);
    c->tend_valid = true;
    if (c->veg_mode == TRM_VEGETATION_STANDALONE) {
        int rc = DISPATCH(c, update_inputs(c, c->state, c->time));
        if (!rc) rc = compute_tendencies ? DISPATCH(c, template veg_launch<VEG_UPDATE>(c, 0.0, 1, 0)) : DISPATCH(c, template veg_launch<VEG_AUX>(c, 0.0, 1, 0));
        return finish(c, rc);
    }
    int rc = DISPATCH(c, update_inputs(c, c->state, c->time));
    if (rc) return rc;
    return finish(c, DISPATCH(c, update_state(c, c->state, compute_tendencies != 0)));
}
int trm_compute_auxiliary(trm_ctx* c) {
    TRM_ENTER(c);
    if (c->veg_mode == TRM_VEGETATION_STANDALONE) return finish(c, DISPATCH(c, template veg_launch<VEG_AUX>(c, 0.0, 1, 0)));
    return finish(c, DISPATCH(c, compute_auxiliary(c, c->state)));
}
int trm_compute_tendencies(trm_ctx* c) {
    TRM_ENTER(c);
    if (c->veg_mode == TRM_VEGETATION_STANDALONE) return finish(c, DISPATCH(c, template veg_launch<VEG_TEND>(c, 0.0, 1, 0)));
    return finish(c, DISPATCH(c, compute_tendencies(c, c->state)));
}
int trm_reset_tendencies(trm_ctx* c) {
    TRM_ENTER(c);
    c->tend_valid = true;
    return finish(c, DISPATCH(c, reset_tendencies(c, c->state)));
}
int trm_explicit_step(trm_ctx* c, double dt) {
    TRM_ENTER(c);
    if (c->veg_mode == TRM_VEGETATION_STANDALONE) return finish(c, DISPATCH(c, template veg_launch<VEG_EXPLICIT>(c, dt, 1, 0)));
    c->top_valid = false;
    c->closure_consistent = false;
    return finish(c, DISPATCH(c, explicit_step(c, c->state, dt)));
}
int trm_closure(trm_ctx* c) {
    TRM_ENTER(c);
    c->top_valid = false;
    c->closure_consistent = true;
    return finish(c, DISPATCH(c, closure(c, c->state)));
}
int trm_invclosure(trm_ctx* c) {
    TRM_ENTER(c);
    c->top_valid = false;
    c->closure_consistent = false;
    return finish(c, DISPATCH(c, invclosure(c, c->state)));
}

int trm_step(trm_ctx* c, double dt, int nsteps, int finalize) {
    TRM_ENTER(c);
    if (nsteps < 0) return fail(c, TRM_EINVAL, "trm_step: nsteps < 0");
    return finish(c, DISPATCH(c, step(c, dt, nsteps, finalize)));
}

namespace {
// The timed entry points return as soon as the stop event has completed: the host POLLS it (a blocking wait is woken by an
// interrupt some 10-20 us later, which a caller's wall clock around a short run of steps would see).
int wait_polling(trm_ctx* c, hipEvent_t ev) {
    for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) {
            (void)hipGetLastError();   // (hipErrorNotReady of the polls is not an error: do not leave it for the next launch check)
            return TRM_OK;
        }
        if (e != hipErrorNotReady) TRM_HIP(c, e);
    }
}
}  // namespace

int trm_step_timed(trm_ctx* c, double dt, int nsteps, int finalize, float* ms) {
    TRM_ENTER(c);
    if (nsteps < 0 || !ms) return fail(c, TRM_EINVAL, "trm_step_timed: bad argument");
    TRM_HIP(c, hipEventRecord(c->ev0, c->stream));
    int rc = DISPATCH(c, step(c, dt, nsteps, finalize));
    if (rc) return rc;
    TRM_HIP(c, hipEventRecord(c->ev1, c->stream));
    if (int rw = wait_polling(c, c->ev1)) return rw;
    TRM_HIP(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
    return TRM_OK;
}

int trm_step_heun(trm_ctx* c, double dt, int nsteps, int finalize) {
    TRM_ENTER(c);
    if (nsteps < 0) return fail(c, TRM_EINVAL, "trm_step_heun: nsteps < 0");
    if (c->veg_mode == TRM_VEGETATION_STANDALONE) return finish(c, DISPATCH(c, veg_step(c, dt, nsteps, finalize, true)));
    const bool generic = c->precision == TRM_F64 ? Ops<double>::generic_bcs(c) : Ops<float>::generic_bcs(c);
    // (the one-launch programs keep the stage in registers; the coupled vegetation stores part of it, the reference-order
    // kernels all of it -- with the generic boundary kinds AND the coupled vegetation they are what runs)
    const bool fused_heun = c->opt_kernel == TRM_KERNEL_FUSED && c->Nz <= 64 && c->veg_mode != TRM_VEGETATION_COUPLED;
    (void)generic;
    if (!fused_heun) {   // the reference-order kernels work on a second copy of the state (fields enabled since the last call included)
        int rc = alloc_fields(c, c->stage);
        if (rc) return rc;
        if (!c->has_stage) c->args_valid = false;
        c->has_stage = true;
    }
    for (int n = 0; n < nsteps; ++n) {
        int fin = (finalize && n == nsteps - 1) ? 1 : 0;
        int rc = DISPATCH(c, heun_step(c, dt, fin));
        if (rc) return rc;
        c->time += dt;
        c->iteration += 1;
    }
    return finish(c, TRM_OK);
}

int trm_step_heun_timed(trm_ctx* c, double dt, int nsteps, int finalize, float* ms) {
    TRM_ENTER(c);
    if (nsteps < 0 || !ms) return fail(c, TRM_EINVAL, "trm_step_heun_timed: bad argument");
    const int async = c->opt_async;
    c->opt_async = 1;                       // (no synchronisation between the two event records)
    TRM_HIP(c, hipEventRecord(c->ev0, c->stream));
    const int rc = trm_step_heun(c, dt, nsteps, finalize);
    c->opt_async = async;
    if (rc) return rc;
    TRM_HIP(c, hipEventRecord(c->ev1, c->stream));
    if (int rw = wait_polling(c, c->ev1)) return rw;
    TRM_HIP(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
    return TRM_OK;
}

// ---- Heun in two calls: the caller evaluates state-dependent forcings / boundary values at the stage in between ------------
namespace {
int ensure_stage(trm_ctx* c) {
    int rc = alloc_fields(c, c->stage);
    if (rc) return rc;
    if (!c->has_stage) c->args_valid = false;
    c->has_stage = true;
    return TRM_OK;
}
}  // namespace

int trm_heun_predict(trm_ctx* c, double dt) {
    TRM_ENTER_HEUN(c);
    if (c->veg_mode == TRM_VEGETATION_STANDALONE) return fail(c, TRM_EUNSUPPORTED, "trm_heun_predict: the standalone VegetationModel has no state-dependent callbacks; use trm_step_heun");
    if (int rc = ensure_stage(c)) return rc;
    c->top_valid = false;
    c->tend_valid = true;
    c->closure_consistent = false;     // (the state is untouched so far; the flag is set again by trm_heun_correct)
    const int rc = DISPATCH(c, heun_predict(c, dt, true));
    if (rc) return rc;
    c->heun_pending = true;
    c->heun_stage_aux = false;
    c->heun_dt = dt;
    return finish(c, TRM_OK);
}

int trm_heun_stage_auxiliary(trm_ctx* c) {
    TRM_ENTER_HEUN(c);
    if (!c->heun_pending) return fail(c, TRM_EINVAL, "trm_heun_stage_auxiliary: call trm_heun_predict first");
    if (c->heun_stage_aux) return TRM_OK;
    const int rc = DISPATCH(c, heun_stage_auxiliary(c));
    if (rc) return rc;
    c->heun_stage_aux = true;
    return finish(c, TRM_OK);
}

int trm_heun_correct(trm_ctx* c, double dt, int finalize) {
    TRM_ENTER_HEUN(c);
    if (!c->heun_pending) return fail(c, TRM_EINVAL, "trm_heun_correct: call trm_heun_predict first");
    if (dt != c->heun_dt) return fail(c, TRM_EINVAL, "trm_heun_correct: dt differs from the dt of trm_heun_predict");
    c->heun_pending = false;
    const int rc = DISPATCH(c, heun_correct(c, dt, finalize, c->heun_stage_aux));
    c->heun_stage_aux = false;
    if (rc) return rc;
    c->top_valid = false;
    c->tend_valid = true;
    c->closure_consistent = true;      // (ends with closure!)
    c->time += dt;
    c->iteration += 1;
    return finish(c, TRM_OK);
}

int trm_stage_field_device_ptr(trm_ctx* c, int field, void** dev, int64_t* pitch_elems) {
    TRM_ENTER_HEUN(c);
    if (!dev || !valid_field(field)) return fail(c, TRM_EINVAL, "trm_stage_field_device_ptr: bad argument");
    if (!c->state.f[field]) return fail(c, TRM_EINVAL, "trm_stage_field_device_ptr: the field exists only after trm_set_vegetation");
    if (int rc = ensure_stage(c)) return rc;
    if (field == TRM_FIELD_VWC_FORCING && !c->stage_vwc_own) {
        if (!c->opt_vwc_field) return fail(c, TRM_EINVAL, "trm_stage_field_device_ptr: upload TRM_FIELD_VWC_FORCING (or set TRM_OPT_VWC_FORCING_FIELD) first");
        TRM_HIP(c, hipMemcpyAsync(c->stage.f[field], c->state.f[field], field_elems(c, field) * c->esize, hipMemcpyDeviceToDevice, c->stream));
        c->stage_vwc_own = true;
        c->args_valid = false;
    }
    *dev = c->stage.f[field];
    if (pitch_elems) *pitch_elems = is_3d(field) ? c->Nzp : 1;
    return TRM_OK;
}

int trm_stage_bc_device_ptr(trm_ctx* c, int var, int side, void** dev) {
    TRM_ENTER_HEUN(c);
    if (!dev || var < 0 || var >= TRM_BCV_COUNT || (side != TRM_TOP && side != TRM_BOTTOM)) return fail(c, TRM_EINVAL, "trm_stage_bc_device_ptr: bad argument");
    if (c->bc_kind[var][side] == TRM_BC_NOFLUX || !c->bc_value[var][side]) return fail(c, TRM_EINVAL, "trm_stage_bc_device_ptr: the condition carries no values (set it with trm_set_bc first)");
    for (const auto& sr : c->series)
        if (sr.is_bc && sr.var == var && sr.side == side) return fail(c, TRM_EINVAL, "trm_stage_bc_device_ptr: the boundary values are evaluated from a time series at both stages");
    c->bc_zero_gradient[var][side] = false;      // (the stage's values become the caller's)
    const size_t bytes = (size_t)c->Nh * c->esize;
    if (!c->bc_value_stage[var][side]) {
        TRM_HIP(c, hipMalloc(&c->bc_value_stage[var][side], bytes));
        c->args_valid = false;
    }
    if (!c->stage_bc_user[var][side]) {
        TRM_HIP(c, hipMemcpyAsync(c->bc_value_stage[var][side], c->bc_value[var][side], bytes, hipMemcpyDeviceToDevice, c->stream));
        c->stage_bc_user[var][side] = true;
    }
    *dev = c->bc_value_stage[var][side];
    return TRM_OK;
}

int trm_save_state(trm_ctx* c) {
    TRM_ENTER_HEUN(c);
    {   // (allocates what is missing: everything the first time, the vegetation fields once they exist)
        int rc = alloc_fields(c, c->saved);
        if (rc) return rc;
        c->has_saved = true;
    }
    for (int f = 0; f < TRM_FIELD_COUNT; ++f)
        if (c->state.f[f] && c->saved.f[f]) TRM_HIP(c, hipMemcpyAsync(c->saved.f[f], c->state.f[f], field_elems(c, f) * c->esize, hipMemcpyDeviceToDevice, c->stream));
    TRM_HIP(c, hipMemcpyAsync(c->saved.kf_top, c->state.kf_top, (size_t)c->Nh * c->esize, hipMemcpyDeviceToDevice, c->stream));
    TRM_HIP(c, hipMemcpyAsync(&c->saved_status, c->d_status, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    TRM_HIP(c, hipStreamSynchronize(c->stream));
    c->saved_time = c->time;
    c->saved_iteration = c->iteration;
    c->saved_tend_valid = c->tend_valid;
    c->saved_closure_consistent = c->closure_consistent;
    return TRM_OK;
}
int trm_restore_state(trm_ctx* c) {
    TRM_ENTER(c);
    if (!c->has_saved) return fail(c, TRM_EINVAL, "trm_restore_state: nothing was saved");
    for (int f = 0; f < TRM_FIELD_COUNT; ++f)
        if (c->state.f[f] && c->saved.f[f]) TRM_HIP(c, hipMemcpyAsync(c->state.f[f], c->saved.f[f], field_elems(c, f) * c->esize, hipMemcpyDeviceToDevice, c->stream));
    TRM_HIP(c, hipMemcpyAsync(c->state.kf_top, c->saved.kf_top, (size_t)c->Nh * c->esize, hipMemcpyDeviceToDevice, c->stream));
    TRM_HIP(c, hipMemcpyAsync(c->d_status, &c->saved_status, sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    c->time = c->saved_time;
    c->iteration = c->saved_iteration;
    c->tend_valid = c->saved_tend_valid;
    c->closure_consistent = c->saved_closure_consistent;
    c->top_valid = false;
    return finish(c, TRM_OK);
}

int trm_clock(const trm_ctx* c, double* time, int64_t* iteration) {
    if (!c) return TRM_EINVAL;
    if (time) *time = c->time;
    if (iteration) *iteration = c->iteration;
    return TRM_OK;
}
int trm_set_clock(trm_ctx* c, double time, int64_t iteration) {
    if (!c) return TRM_EINVAL;
    c->heun_pending = false;
    c->time = time;
    c->iteration = iteration;
    return TRM_OK;
}

int trm_reduce(trm_ctx* c, int field, int op, double* out) {
    TRM_ENTER_HEUN(c);
    if (!out || !valid_field(field)) return fail(c, TRM_EINVAL, "trm_reduce: bad argument");
    if (is_tendency(field) && !c->tend_valid) return fail(c, TRM_ESTALE, kStaleTendencies);
    return c->precision == TRM_F64 ? reduce_impl<double>(c, field, op, out) : reduce_impl<float>(c, field, op, out);
}

// ---- vegetation ---------------------------------------------------------------------------
int trm_default_vegetation_params(trm_vegetation_params* p) {
    if (!p) return TRM_EINVAL;
    // photosynthesis.jl:17-68
    p->tau25 = 2600.0; p->Kc25 = 30.0; p->Ko25 = 3.0e4; p->q10_tau = 0.57; p->q10_Kc = 2.1; p->q10_Ko = 1.2; p->alpha_leaf = 0.17;
    p->alpha_a = 0.5; p->alpha_C3 = 0.08; p->cq = 4.6e-6; p->k_ext = 0.5; p->T_CO2_high = 42.0; p->T_CO2_low = -4.0;
    p->T_photos_high = 30.0; p->T_photos_low = 15.0; p->theta_r = 0.7;
    p->g1 = 2.3; p->g_min = 0.5;                                              // stomatal_conductance.jl:16-24
    p->cn_sapwood = 330.0; p->cn_root = 29.0; p->aws = 10.0;                  // autotrophic_respiration.jl:14-23
    p->SLA = 10.0; p->awl = 2.0; p->LAI_min = 1.0; p->LAI_max = 6.0; p->gamma_L = 0.3; p->gamma_R = 0.3; p->gamma_S = 0.05;   // carbon_dynamics.jl:18-43
    p->nu_seed = 0.001; p->gamma_v_min = 0.002;                               // vegetation_dynamics.jl:15-22
    p->root_a = 7.0; p->root_b = 2.0;                                         // root_distribution.jl:23-29
    p->wilting_point = 0.05; p->field_capacity = 0.25;                        // soil_hydraulic_properties.jl:74-80
    p->C_mass = 12.0;                                                         // physical_constants.jl:50
    p->alpha_int = 0.2; p->canopy_k_ext = 0.5; p->w_can_max = 2.0e-4; p->tau_w = 86400.0;   // canopy_interception.jl:37-49
    p->C_can = 0.006;                                                         // canopy_evapotranspiration.jl:33-40
    return TRM_OK;
}

int trm_set_vegetation(trm_ctx* c, const trm_vegetation_params* p, int mode) {
    TRM_ENTER(c);
    if (!p || mode < TRM_VEGETATION_OFF || mode > TRM_VEGETATION_COUPLED) return fail(c, TRM_EINVAL, "trm_set_vegetation: bad argument");
    if (mode == TRM_VEGETATION_COUPLED && !c->params.seb)
        return fail(c, TRM_EINVAL, "trm_set_vegetation: TRM_VEGETATION_COUPLED needs a LandModel context (surface_energy_balance = 1)");
    TRM_HIP(c, hipStreamSynchronize(c->stream));
    c->veg_params = *p;
    const bool first = c->veg_mode == TRM_VEGETATION_OFF && mode != TRM_VEGETATION_OFF && !c->state.f[TRM_FIELD_ROOT_FRACTION];
    c->veg_mode = mode;
    c->args_valid = false;
    if (mode == TRM_VEGETATION_OFF) return TRM_OK;
    int rc = alloc_fields(c, c->state);          // the 3-D vegetation fields
    if (rc) return rc;
    if (first) {
        // input defaults: CO2 380 ppm (prescribed_atmosphere.jl:14), soil moisture limiting factor 1 (photosynthesis.jl:76),
        // ground temperature 10 degC (autotrophic_respiration.jl:38)
        const struct { int f; double v; } defaults[] = {{TRM_FIELD_CO2, 380.0}, {TRM_FIELD_SOIL_MOISTURE_LIMITING_FACTOR, 1.0},
                                                       {TRM_FIELD_VEGETATION_GROUND_TEMPERATURE, 10.0}};
        for (auto& d : defaults) {
            rc = c->precision == TRM_F64 ? fill_row<double>(c, d.f, d.v) : fill_row<float>(c, d.f, d.v);
            if (rc) return rc;
        }
    }
    return c->precision == TRM_F64 ? Ops<double>::upload_root_fraction(c) : Ops<float>::upload_root_fraction(c);
}

int trm_compute_plant_available_water(trm_ctx* c) {
    TRM_ENTER(c);
    if (c->veg_mode == TRM_VEGETATION_OFF) return fail(c, TRM_EINVAL, "trm_compute_plant_available_water: call trm_set_vegetation first");
    return finish(c, DISPATCH(c, plant_available_water(c)));
}

// ---- multi-device diagnostics --------------------------------------------------------
int trm_comm_unique_id(void* id128) {
    if (!id128) return TRM_EINVAL;
    Rccl* r = rccl();
    if (!r->error.empty()) return fail(nullptr, TRM_ECOMM, r->error);
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    ncclResult_t e = r->GetUniqueId(&id);
    if (e != ncclSuccess) return fail(nullptr, TRM_ECOMM, std::string("ncclGetUniqueId: ") + r->GetErrorString(e));
    std::memcpy(id128, &id, sizeof(id));
    return TRM_OK;
}

namespace {
// communicators of one ncclUniqueId form one group: the grouped calls of trm_*_all refuse a list that mixes groups (RCCL would
// wait in ncclGroupEnd for ranks nobody posts)
uint64_t unique_id_hash(const ncclUniqueId& id) {
    uint64_t h = 1469598103934665603ull;
    for (size_t n = 0; n < sizeof(id); ++n) h = (h ^ (unsigned char)id.internal[n]) * 1099511628211ull;
    return h ? h : 1;
}
// the side stream and the send | receive buffer of a context's collectives
int comm_buffers(trm_ctx* c) {
    TRM_HIP(c, hipSetDevice(c->device));
    TRM_HIP(c, hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
    TRM_HIP(c, hipMalloc((void**)&c->d_comm, (size_t)(4 * (c->Nz + 1) + 16) * sizeof(double)));
    return TRM_OK;
}
}  // namespace

int trm_comm_init(trm_ctx* c, int rank, int world, const void* id128) {
    TRM_ENTER_HEUN(c);
    if (!id128 || world < 1 || rank < 0 || rank >= world) return fail(c, TRM_EINVAL, "trm_comm_init: bad argument");
    if (c->comm) return fail(c, TRM_EINVAL, "trm_comm_init: the context already has a communicator");
    Rccl* r = rccl();
    if (!r->error.empty()) return fail(c, TRM_ECOMM, r->error);
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    TRM_NCCL(c, r->CommInitRank(&c->comm, world, id, rank));
    c->comm_rank = rank;
    c->comm_world = world;
    c->comm_group = unique_id_hash(id);
    if (int rc = comm_buffers(c)) {
        (void)trm_comm_destroy(c);
        return rc;
    }
    return TRM_OK;
}

int trm_comm_destroy(trm_ctx* c) {
    if (!c || !c->comm) return TRM_OK;
    (void)hipSetDevice(c->device);
    if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream);
    (void)rccl()->CommDestroy(c->comm);
    c->comm = nullptr;
    if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
    c->comm_stream = nullptr;
    if (c->d_comm) (void)hipFree(c->d_comm);
    c->d_comm = nullptr;
    c->comm_world = 1;
    c->comm_rank = 0;
    c->comm_group = 0;
    return TRM_OK;
}

int trm_comm_info(const trm_ctx* c, int* rank, int* world) {
    if (!c) return TRM_EINVAL;
    if (rank) *rank = c->comm_rank;
    if (world) *world = c->comm ? c->comm_world : 0;
    return TRM_OK;
}

namespace {
// What travels for one reduction: the local per-row results, for MIN / MAX next to a NaN flag that travels with the same
// operator -- a NaN on any rank must reach every rank (Base.minimum / maximum), which RCCL's min / max do not promise.
int reduce_rows(const trm_ctx* c, int field, int op) { return op == TRM_REDUCE_VOLUME_INTEGRAL_Z ? 1 : (int)field_rows(c, field); }
int pack_reduce(int op, int rows, const double* local, std::vector<double>& buf) {
    if (op == TRM_REDUCE_MIN || op == TRM_REDUCE_MAX) {
        const double sentinel = op == TRM_REDUCE_MIN ? HUGE_VAL : -HUGE_VAL, yes = op == TRM_REDUCE_MIN ? -1.0 : 1.0;
        buf.assign((size_t)2 * rows, 0.0);
        for (int r = 0; r < rows; ++r) {
            const bool nan = local[r] != local[r];
            buf[r] = nan ? sentinel : local[r];
            buf[rows + r] = nan ? yes : 0.0;
        }
        return 2 * rows;
    }
    buf.assign(local, local + rows);
    return rows;
}
ncclRedOp_t reduce_operator(int op) { return op == TRM_REDUCE_MIN ? ncclMin : ((op == TRM_REDUCE_MAX || op == TRM_REDUCE_HASNAN) ? ncclMax : ncclSum); }
void unpack_reduce(int op, int rows, const std::vector<double>& buf, double* out) {
    if (op == TRM_REDUCE_MIN || op == TRM_REDUCE_MAX) {
        for (int r = 0; r < rows; ++r) out[r] = buf[rows + r] != 0.0 ? std::nan("") : buf[r];
        return;
    }
    for (int r = 0; r < rows; ++r) out[r] = buf[r];
}
// the all-reduce of the packed form on the HOST, ranks folded in order (one process: no wire needed)
void fold_packed(int op, int n, std::vector<double>& acc, const std::vector<double>& x) {
    for (int j = 0; j < n; ++j) {
        if (op == TRM_REDUCE_MIN) acc[j] = std::min(acc[j], x[j]);
        else if (op == TRM_REDUCE_MAX || op == TRM_REDUCE_HASNAN) acc[j] = std::max(acc[j], x[j]);
        else acc[j] += x[j];
    }
}
// 1: every context holds rank i of n of ONE communicator group (the grouped all-reduce applies); 0: none has a communicator (host
// fold); -1: anything else -- communicators of different groups, a subset of a group, ranks out of order: refused, because the
// grouped call would post an incomplete set of ranks and RCCL would wait for the others forever
int comm_state(trm_ctx** ctxs, int n) {
    int with = 0;
    for (int i = 0; i < n; ++i) with += ctxs[i]->comm != nullptr;
    if (with == 0) return 0;
    if (with != n) return -1;
    for (int i = 0; i < n; ++i)
        if (ctxs[i]->comm_world != n || ctxs[i]->comm_rank != i || ctxs[i]->comm_group != ctxs[0]->comm_group) return -1;
    return 1;
}
const char* kMixedCommunicators = ": the contexts hold communicators that do not form ONE group in rank order (all of trm_comm_init_all's "
                                  "contexts, in its order) -- a grouped collective over them would wait for ranks nobody posts";
// n all-reduces of `count` doubles, one per context, issued from ONE thread inside a group (RCCL would otherwise block in the
// first one waiting for ranks this thread has not reached yet)
int grouped_allreduce(trm_ctx** ctxs, int n, std::vector<std::vector<double>>& bufs, int count, ncclRedOp_t op) {
    Rccl* r = rccl();
    for (int i = 0; i < n; ++i) {
        trm_ctx* c = ctxs[i];
        TRM_HIP(c, hipSetDevice(c->device));
        TRM_HIP(c, hipMemcpyAsync(c->d_comm, bufs[i].data(), (size_t)count * sizeof(double), hipMemcpyHostToDevice, c->comm_stream));
    }
    TRM_NCCL(ctxs[0], r->GroupStart());
    for (int i = 0; i < n; ++i) {
        trm_ctx* c = ctxs[i];
        const ncclResult_t e = r->AllReduce(c->d_comm, c->d_comm + count, (size_t)count, ncclDouble, op, c->comm, c->comm_stream);
        if (e != ncclSuccess) {
            (void)r->GroupEnd();
            return fail(c, TRM_ECOMM, std::string("ncclAllReduce: ") + r->GetErrorString(e));
        }
    }
    TRM_NCCL(ctxs[0], r->GroupEnd());
    for (int i = 0; i < n; ++i) {
        trm_ctx* c = ctxs[i];
        TRM_HIP(c, hipSetDevice(c->device));
        TRM_HIP(c, hipMemcpyAsync(bufs[i].data(), c->d_comm + count, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, c->comm_stream));
        TRM_HIP(c, hipStreamSynchronize(c->comm_stream));
    }
    return TRM_OK;
}
int check_ctx_list(trm_ctx** ctxs, int n, const char* who) {
    if (!ctxs || n < 1) return fail(nullptr, TRM_EINVAL, std::string(who) + ": need at least one context");
    for (int i = 0; i < n; ++i)
        if (!ctxs[i]) return fail(nullptr, TRM_EINVAL, std::string(who) + ": null context in the list");
    for (int i = 1; i < n; ++i)
        if (ctxs[i]->Nz != ctxs[0]->Nz) return fail(ctxs[0], TRM_EINVAL, std::string(who) + ": the contexts are shards of one grid: same number of levels expected");
    // (a caller that holds only the list reads the first non-empty message: none may be left over from an earlier call)
    for (int i = 0; i < n; ++i) ctxs[i]->err.clear();
    return TRM_OK;
}
}  // namespace

int trm_reduce_global(trm_ctx* c, int field, int op, double* out) {
    int rc = trm_reduce(c, field, op, out);     // this device's columns (synchronous)
    if (rc) return rc;
    if (!c->comm) return fail(c, TRM_EINVAL, "trm_reduce_global: call trm_comm_init first");
    const int rows = reduce_rows(c, field, op);
    std::vector<double> buf;
    const int count = pack_reduce(op, rows, out, buf);
    rc = comm_allreduce(c, buf.data(), count, reduce_operator(op));
    if (rc) return rc;
    unpack_reduce(op, rows, buf, out);
    return TRM_OK;
}

int trm_status_global(trm_ctx* c, uint32_t* flags) {
    uint32_t local = 0;
    int rc = trm_status(c, &local);
    if (rc) return rc;
    if (!flags) return TRM_EINVAL;
    if (!c->comm) return fail(c, TRM_EINVAL, "trm_status_global: call trm_comm_init first");
    double bits[8];
    for (int b = 0; b < 8; ++b) bits[b] = (double)((local >> b) & 1u);
    rc = comm_allreduce(c, bits, 8, ncclMax);    // OR of the flag bits
    if (rc) return rc;
    *flags = 0;
    for (int b = 0; b < 8; ++b) *flags |= bits[b] != 0.0 ? (1u << b) : 0u;
    return TRM_OK;
}

// ---- one host thread, n contexts ("1 process x 8 HIP devices") ---------------------------------------------
int trm_comm_init_all(trm_ctx** ctxs, int n) {
    if (int rc = check_ctx_list(ctxs, n, "trm_comm_init_all")) return rc;
    for (int i = 0; i < n; ++i) {
        if (ctxs[i]->comm) return fail(ctxs[i], TRM_EINVAL, "trm_comm_init_all: the context already has a communicator");
        for (int j = 0; j < i; ++j)
            if (ctxs[j]->device == ctxs[i]->device && !std::getenv("TRM_RCCL_ALLOW_SHARED_DEVICE"))      // (tests: with tests/fake_rccl.c)
                return fail(ctxs[i], TRM_EINVAL, "trm_comm_init_all: two contexts on one device (RCCL takes one rank per device; without communicators "
                                                 "trm_reduce_global_all / trm_status_global_all combine the shards on the host)");
    }
    Rccl* r = rccl();
    if (!r->error.empty()) return fail(ctxs[0], TRM_ECOMM, r->error);
    ncclUniqueId id;
    TRM_NCCL(ctxs[0], r->GetUniqueId(&id));
    TRM_NCCL(ctxs[0], r->GroupStart());
    for (int i = 0; i < n; ++i) {
        trm_ctx* c = ctxs[i];
        ncclResult_t e = ncclSuccess;
        if (hipSetDevice(c->device) != hipSuccess) e = ncclUnhandledCudaError;
        if (e == ncclSuccess) e = r->CommInitRank(&c->comm, n, id, i);
        if (e != ncclSuccess) {
            (void)r->GroupEnd();
            // (the communicators handed out so far were never completed by a ncclGroupEnd that succeeded: nothing to destroy)
            for (int j = 0; j <= i; ++j) ctxs[j]->comm = nullptr;
            return fail(c, TRM_ECOMM, std::string("ncclCommInitRank (grouped): ") + r->GetErrorString(e));
        }
    }
    {
        const ncclResult_t e = r->GroupEnd();
        if (e != ncclSuccess) {
            for (int j = 0; j < n; ++j) ctxs[j]->comm = nullptr;
            return fail(ctxs[0], TRM_ECOMM, std::string("ncclGroupEnd: ") + r->GetErrorString(e));
        }
    }
    for (int i = 0; i < n; ++i) {
        trm_ctx* c = ctxs[i];
        c->comm_rank = i;
        c->comm_world = n;
        c->comm_group = unique_id_hash(id);
    }
    for (int i = 0; i < n; ++i) {
        trm_ctx* c = ctxs[i];
        if (int rc = comm_buffers(c)) {
            // a context without its stream / buffer must not keep a communicator (the grouped all-reduce would write through a
            // null pointer): the whole group goes, the failing context's message is the one reported
            const std::string msg = c->err;
            for (int j = 0; j < n; ++j) (void)trm_comm_destroy(ctxs[j]);
            return fail(ctxs[0], rc, "trm_comm_init_all: context " + std::to_string(i) + ": " + msg);
        }
    }
    return TRM_OK;
}

namespace {
int step_all(trm_ctx** ctxs, int n, const char* who, int (*step)(trm_ctx*, double, int, int), double dt, int nsteps, int finalize) {
    if (int rc = check_ctx_list(ctxs, n, who)) return rc;
    // every context is stepped without waiting for it (each on its own stream, each device running its shard), then ONE wait
    // per context -- unless the caller drives them asynchronously anyway
    std::vector<int> async((size_t)n);
    int rc = TRM_OK;
    // The steps are dealt to the devices launch by launch -- every context receives the steps of ONE of its launches (1, or up to
    // 50 with the resident program) before the next context is visited -- so that all devices start at once instead of device
    // d waiting for the host to have enqueued all nsteps launches of devices 0 .. d - 1.  Same results as one call per context
    // (a call of m steps equals m calls of one step, bit for bit).
    // A context that launches per step (steps_per_launch = 1, or a model the resident program does not take: generic boundary
    // kinds, the coupled vegetation, Heun) would receive 50 launches before the next device is visited: the chunk is the smallest
    // number of steps ONE launch of any context covers.
    int chunk = 50;
    const bool heun = step == trm_step_heun;
    for (int i = 0; i < n; ++i) {
        async[(size_t)i] = ctxs[i]->opt_async;
        ctxs[i]->opt_async = 1;
        chunk = std::min(chunk, heun ? 1 : (ctxs[i]->precision == TRM_F64 ? Ops<double>::steps_per_launch_now(ctxs[i]) : Ops<float>::steps_per_launch_now(ctxs[i])));
    }
    int failed = -1;
    for (int done = 0; done < nsteps && !rc; done += chunk) {
        const int m = std::min(chunk, nsteps - done);
        const int fin = (finalize && done + m == nsteps) ? 1 : 0;
        for (int i = 0; i < n && !rc; ++i) {
            rc = step(ctxs[i], dt, m, fin);
            if (rc) failed = i;
        }
    }
    if (nsteps == 0 && finalize)
        for (int i = 0; i < n && !rc; ++i) {
            rc = step(ctxs[i], dt, 0, finalize);
            if (rc) failed = i;
        }
    for (int i = 0; i < n; ++i) ctxs[i]->opt_async = async[(size_t)i];
    for (int i = 0; i < n; ++i) {
        if (async[(size_t)i] && failed < 0) continue;
        const int rs = trm_synchronize(ctxs[i]);      // (also after a failure: nothing is left running behind the caller's back)
        if (!rc) { rc = rs; if (rs) failed = i; }
    }
    if (failed >= 0) {
        // the contexts stand at different clocks now: say which one failed and where every one stands (its message moves to
        // the first context, where a caller that holds only the list looks)
        std::string msg = std::string(who) + ": context " + std::to_string(failed) + " failed: " + ctxs[failed]->err + " [iterations:";
        for (int i = 0; i < n; ++i) msg += " " + std::to_string((long long)ctxs[i]->iteration);
        msg += "]";
        for (int i = 0; i < n; ++i) ctxs[i]->err = msg;
    }
    return rc;
}
}  // namespace
int trm_step_all(trm_ctx** ctxs, int n, double dt, int nsteps, int finalize) {
    return step_all(ctxs, n, "trm_step_all", trm_step, dt, nsteps, finalize);
}
int trm_step_heun_all(trm_ctx** ctxs, int n, double dt, int nsteps, int finalize) {
    return step_all(ctxs, n, "trm_step_heun_all", trm_step_heun, dt, nsteps, finalize);
}
int trm_synchronize_all(trm_ctx** ctxs, int n) {
    if (int rc = check_ctx_list(ctxs, n, "trm_synchronize_all")) return rc;
    int rc = TRM_OK;
    for (int i = 0; i < n; ++i) {
        const int rs = trm_synchronize(ctxs[i]);
        if (!rc) rc = rs;
    }
    return rc;
}

int trm_reduce_global_all(trm_ctx** ctxs, int n, int field, int op, double* out) {
    if (int rc = check_ctx_list(ctxs, n, "trm_reduce_global_all")) return rc;
    if (!out || !valid_field(field)) return fail(ctxs[0], TRM_EINVAL, "trm_reduce_global_all: bad argument");
    const int rows = reduce_rows(ctxs[0], field, op);
    std::vector<std::vector<double>> bufs((size_t)n);
    std::vector<double> local((size_t)field_rows(ctxs[0], field) + 1);
    int count = 0;
    for (int i = 0; i < n; ++i) {
        int rc = trm_reduce(ctxs[i], field, op, local.data());       // this device's columns
        if (rc) return rc;
        count = pack_reduce(op, rows, local.data(), bufs[(size_t)i]);
    }
    const int cs = comm_state(ctxs, n);
    if (cs < 0) return fail(ctxs[0], TRM_EINVAL, std::string("trm_reduce_global_all") + kMixedCommunicators);
    if (n > 1 && cs == 1) {
        if (int rc = grouped_allreduce(ctxs, n, bufs, count, reduce_operator(op))) return rc;
    } else {
        for (int i = 1; i < n; ++i) fold_packed(op, count, bufs[0], bufs[(size_t)i]);
    }
    unpack_reduce(op, rows, bufs[0], out);
    return TRM_OK;
}

int trm_status_global_all(trm_ctx** ctxs, int n, uint32_t* flags) {
    if (int rc = check_ctx_list(ctxs, n, "trm_status_global_all")) return rc;
    if (!flags) return TRM_EINVAL;
    std::vector<std::vector<double>> bufs((size_t)n, std::vector<double>(8));
    for (int i = 0; i < n; ++i) {
        uint32_t local = 0;
        int rc = trm_status(ctxs[i], &local);
        if (rc) return rc;
        for (int b = 0; b < 8; ++b) bufs[(size_t)i][(size_t)b] = (double)((local >> b) & 1u);
    }
    const int cs = comm_state(ctxs, n);
    if (cs < 0) return fail(ctxs[0], TRM_EINVAL, std::string("trm_status_global_all") + kMixedCommunicators);
    if (n > 1 && cs == 1) {
        if (int rc = grouped_allreduce(ctxs, n, bufs, 8, ncclMax)) return rc;
    } else {
        for (int i = 1; i < n; ++i) fold_packed(TRM_REDUCE_MAX, 8, bufs[0], bufs[(size_t)i]);
    }
    *flags = 0;
    for (int b = 0; b < 8; ++b) *flags |= bufs[0][(size_t)b] != 0.0 ? (1u << b) : 0u;
    return TRM_OK;
}

int trm_status(trm_ctx* c, uint32_t* flags) {
    TRM_ENTER_HEUN(c);
    if (!flags) return TRM_EINVAL;
    TRM_HIP(c, hipMemcpyAsync(flags, c->d_status, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    TRM_HIP(c, hipStreamSynchronize(c->stream));
    return TRM_OK;
}

int trm_set_status(trm_ctx* c, uint32_t flags) {
    TRM_ENTER(c);
    TRM_HIP(c, hipMemcpyAsync(c->d_status, &flags, sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    TRM_HIP(c, hipStreamSynchronize(c->stream));      // (`flags` lives on this call's stack)
    return TRM_OK;
}

int trm_set_option(trm_ctx* c, int option, int value) {
    if (!c) return TRM_EINVAL;
    c->args_valid = false;
    c->heun_pending = false;
    switch (option) {
        case TRM_OPT_ASYNC: c->opt_async = value != 0; return TRM_OK;
        case TRM_OPT_STEP_KERNEL:
            if (value != TRM_KERNEL_FUSED && value != TRM_KERNEL_UNFUSED) break;
            c->opt_kernel = value;
            return TRM_OK;
        case TRM_OPT_WRITE_KF_EVERY_STEP: c->opt_write_kf = value != 0; return TRM_OK;
        case TRM_OPT_VWC_FORCING_FIELD: c->opt_vwc_field = value != 0; return TRM_OK;
        case TRM_OPT_PACKED_F32: c->opt_packed = value != 0; return TRM_OK;
        case TRM_OPT_DERIVE_CLOSURE_FIELDS:
            if (value < 0 || value > 5) break;
            c->opt_derive = value;
            return TRM_OK;
        case TRM_OPT_STEPS_PER_LAUNCH:
            if (value < 0 || value > 100000) break;
            c->opt_steps_per_launch = value;
            return TRM_OK;
        case TRM_OPT_PIPELINE_PARTS:
            if (value < 0 || value > 2) break;
            c->opt_pipeline = value;
            return TRM_OK;
        case TRM_OPT_SINGLE_STEP_PROGRAM:
            if (value < 0 || value > 2) break;
            c->opt_single_step = value;
            return TRM_OK;
        case TRM_OPT_BC_SIGNATURE: c->opt_bc_signature = value != 0; return TRM_OK;
        case TRM_OPT_ZERO_GRADIENT_FAST: c->opt_zero_gradient_fast = value != 0; c->args_valid = false; return TRM_OK;
        case TRM_OPT_SURFACE_IN_LAUNCH:
            if (value < 0 || value > 2) break;
            c->opt_front = value;
            return TRM_OK;
        default: break;
    }
    return fail(c, TRM_EINVAL, "trm_set_option: unknown option or value");
}
int trm_get_option(const trm_ctx* c, int option, int* value) {
    if (!c || !value) return TRM_EINVAL;
    switch (option) {
        case TRM_OPT_ASYNC: *value = c->opt_async; return TRM_OK;
        case TRM_OPT_STEP_KERNEL: *value = c->opt_kernel; return TRM_OK;
        case TRM_OPT_WRITE_KF_EVERY_STEP: *value = c->opt_write_kf; return TRM_OK;
        case TRM_OPT_VWC_FORCING_FIELD: *value = c->opt_vwc_field; return TRM_OK;
        case TRM_OPT_PACKED_F32: *value = c->opt_packed; return TRM_OK;
        case TRM_OPT_DERIVE_CLOSURE_FIELDS: *value = c->opt_derive; return TRM_OK;
        case TRM_OPT_STEPS_PER_LAUNCH: *value = c->opt_steps_per_launch; return TRM_OK;
        case TRM_OPT_PIPELINE_PARTS: *value = c->opt_pipeline; return TRM_OK;
        case TRM_OPT_SINGLE_STEP_PROGRAM: *value = c->opt_single_step; return TRM_OK;
        case TRM_OPT_BC_SIGNATURE: *value = c->opt_bc_signature; return TRM_OK;
        case TRM_OPT_ZERO_GRADIENT_FAST: *value = c->opt_zero_gradient_fast; return TRM_OK;
        case TRM_OPT_SURFACE_IN_LAUNCH: *value = c->opt_front; return TRM_OK;
        case TRM_INFO_LAST_PROGRAM: *value = c->last_program; return TRM_OK;
        case TRM_INFO_GENERIC_BOUNDARY_KERNELS: *value = (c->precision == TRM_F64 ? trmh::Policy<double>::generic_bcs(c) : trmh::Policy<float>::generic_bcs(c)) ? 1 : 0; return TRM_OK;
        case TRM_INFO_BC_SIGNATURE: *value = trmh::bc_signature_of(c); return TRM_OK;
        case TRM_INFO_TOP_ARRAYS_CURRENT: *value = c->top_valid ? 1 : 0; return TRM_OK;
        case TRM_INFO_CLOSURE_CONSISTENT: *value = (c->closure_consistent && !c->closure_escaped) ? 1 : 0; return TRM_OK;
        default: return TRM_EINVAL;
    }
}

int trm_set_stream(trm_ctx* c, void* hip_stream) {
    TRM_ENTER(c);
    TRM_HIP(c, hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return TRM_OK;
}
int trm_synchronize(trm_ctx* c) {
    TRM_ENTER_HEUN(c);
    TRM_HIP(c, hipStreamSynchronize(c->stream));
    return TRM_OK;
}

}  // extern "C"
