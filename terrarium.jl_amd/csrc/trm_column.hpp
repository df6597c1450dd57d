// trm_column.hpp -- the register-resident column program (lane = soil level).
//
// A soil column lives in the registers of LPC = 32 / 64 lanes for the whole launch.  The three building blocks --
//   column_tendencies   compute_auxiliary! + compute_tendencies! of the state held in registers (reference a4-a6)
//   column_advance      compute_z_bcs! + explicit_step! + hydrology closure (repair, water table)  (a7, a9)
//   column_closure      energy closure + saturation_to_pressure!                                   (a8, a9)
// are composed into three programs by k_column:
//   PROG_EULER   one ForwardEuler step                                 (forward_euler.jl:19-31)
//   PROG_HEUN    one Heun step, both stages in registers: the stage never touches memory       (heun.jl:37-71)
//   PROG_MULTI   `nsteps` ForwardEuler steps on the resident column, fields written once per launch (temporal
//                blocking of run!'s loop, model_integrator.jl:72-88; legal because columns are independent)
// Every operation is the one the per-step kernels perform, in the same order: results are bit-identical to them
// (tests/test_gpu_column_programs.py).
//
// DERIVE: temperature and liquid fraction of the incoming state are re-derived from (U, sat) by the energy closure
// instead of being read -- legal only when the stored (T, liq) ARE the closure of the stored (U, sat), which the host
// tracks (trm_ctx::closure_consistent): 2 of the 5 field reads disappear.  The kernel is bound by the bytes it moves
// (profiles/tools/microbench/memfloor.hip: every access pattern reaches the same floor), so bytes are what counts.
//
// Boundary conditions: the branch-free kinds only (Value on temperature, Flux on the prognostics, LandModel wiring);
// anything else takes k_step_wave (trm_kernels.hpp).
#pragma once
#include "trm_kernels.hpp"

namespace trm {

enum { PROG_EULER = 0, PROG_HEUN = 1, PROG_MULTI = 2 };
// (diagnostic builds of the in-launch surface processes, wrong results by construction: 1 no surface chain, 2 no granule poll,
// 4 the granules taken as valid whatever their tags)
#ifndef TRM_FRONT_DIAG
#define TRM_FRONT_DIAG 0
#endif
#ifndef TRM_FRONT_POLL_SLEEP
#define TRM_FRONT_POLL_SLEEP 4       // s_sleep between two polls of the granules (x 64 clocks)
#endif
#ifndef TRM_FRONT_PRIO
#define TRM_FRONT_PRIO 3             // s_setprio of the surface waves
#endif

template <class NF> struct Cell { NF U, sat, T, liq, psi; };
// what one lane knows about its place in the column
struct LaneInfo { int lane, k; bool is_bot, is_top, act; unsigned long long m_act, m_bot, m_top; };   // m_*: the wave's ballots of act, is_bot, is_top
// boundary inputs of this lane's column (every lane of a column holds the same values)
template <class NF> struct ColumnBC {
    NF bTb, bTt;            // temperature boundary values (used when the Value condition is set)
    NF flux_U, flux_S;      // compute_z_bcs! term of this lane's cell (flux * Az / V, signed; 0 in the interior)
    // whether any flux condition feeds flux_U / flux_S (wave-uniform): without one the term is +0 in every lane and the
    // tendencies, which are never -0.0 (`0 + ...`), are left alone instead of receiving `+ 0`
    bool has_U = true, has_S = true;
};
template <class NF> struct Tendency { NF gU, gS, Kf_lo, Kc; };

// Energy closure (soil_energy_closures.jl:99-159) for a whole wave: same results as energy_closure(), but the
// liquid-fraction divide U / (-L_theta + eps) is skipped when no lane needs its value.  For a frozen cell
// (U < -L_theta) the reference's `false * (1 - x)` is a zero carrying the sign of 1 - x; with L_theta > eps the
// quotient x of two distinct floats of the same sign and |U| > |denominator| rounds to >= 1 + 2^-52, so 1 - x < 0
// and the result is -0.0 without dividing.  Cells in phase change (and L_theta <= eps, sat below 1.4e-24) take the
// divide -- decided per wave by one ballot.
// a lane needs the divide when it is in phase change, NaN, or its L_theta vanishes: !thawed && !(frozen && L_theta > eps) --
// combined on the scalar unit from three ballots (the lane-wise boolean expression costs 8 vector instructions)
TRM_DEV bool no_lane_divides(bool thawed, bool frozen, bool latent_resolved) {
    const unsigned long long need = ~wave_ballot(thawed) & ~(wave_ballot(frozen) & wave_ballot(latent_resolved)) & wave_ballot(true);
    return need == 0ull;
}
template <class NF> TRM_DEV NF liquid_fraction_wave(const DevParams<NF>& p, NF U, NF sat) {
    const NF Lth = p.L * sat * p.por;
    const NF nLth = -Lth;
    const bool thawed = U >= NF(0), frozen = U < nLth;
    if (no_lane_divides(thawed, frozen, Lth > Limits<NF>::eps())) return thawed ? NF(1) : NF(-0.0);
    return thawed ? NF(1) : boolmul(U >= nLth, NF(1) - safediv(U, nLth));
}
// CHECK: 0 = no composition check (the caller ignores it: a state whose bounds were flagged by the launch that produced it),
// 1 = the full check of volumetric_fractions (soil_volume.jl:26-28), 2 = the saturation has been through the repair of this
// step and is in [0, 1] or NaN by construction (adjust_saturation_profile!: every cell above 1 / below 0 is levelled, the top
// overflows, the bottom is clamped), so `0 <= sat <= 1` is `sat == sat`.  In the common path the liquid fraction is 1 or -0.0:
// inside the bounds; the path that divides checks it.  Same flags as the full check, 1 compare instead of 4.
// Returns the volumetric fractions it formed (the tendencies that follow need the same ones).
template <class NF, int CHECK = 1> TRM_DEV Frac<NF> energy_closure_wave(const DevParams<NF>& p, NF U, NF sat, NF& liq, NF& T, uint32_t& viol) {
    const NF Lth = p.L * sat * p.por;
    const NF nLth = -Lth;
    const bool thawed = U >= NF(0), frozen = U < nLth;
    bool ok;
    if (no_lane_divides(thawed, frozen, Lth > Limits<NF>::eps())) {
        liq = thawed ? NF(1) : NF(-0.0);
        ok = (CHECK == 2 && TRM_CUT_CHECK) ? sat == sat : (TRM_CUT_CHECK ? (NF(0) <= sat && sat <= NF(1)) : ((NF(0) <= sat && sat <= NF(1)) && (NF(0) <= liq && liq <= NF(1))));
    } else {
        TRM_PHASE("rare+ phase-change divide");
        liq = thawed ? NF(1) : boolmul(U >= nLth, NF(1) - safediv(U, nLth));
        ok = (NF(0) <= sat && sat <= NF(1)) && (NF(0) <= liq && liq <= NF(1));
        TRM_PHASE("rare-");
    }
    if (CHECK != 0) viol |= ok ? 0u : 2u;
    const Frac<NF> f = fractions_unchecked(p, sat, liq);
    const NF C = heat_capacity(p, f);
    const NF num = frozen ? (U + Lth) : U;
    const NF quo = div_nr(num, C);
    T = (frozen || thawed) ? quo : NF(0);
    return f;
}

// compute_auxiliary! + compute_tendencies! of the column in registers, WITHOUT the compute_z_bcs! terms.
// `pre`: the volumetric fractions of (c.sat, c.liq) when the closure that produced the cell has just formed them (same operands,
// same operations: same bits), else null.
template <class NF, bool RICHARDS, int HYD, int LPC>
TRM_DEV Tendency<NF> column_tendencies(const View<NF>& v, const DevParams<NF>& p, const LevelGeom<NF>& L, const LaneInfo& ln,
                                       const Cell<NF>& c, NF bTb, NF bTt, bool need_kc, uint32_t& viol, const Frac<NF>* pre = nullptr) {
    const bool is_bot = ln.is_bot, is_top = ln.is_top;
    // (composition bounds of an incoming state were flagged by the launch / program step that produced it)
    const Frac<NF> f = (pre && TRM_CUT_FRAC) ? *pre : fractions_unchecked(p, c.sat, c.liq);
    const NF kap = conductivity(p, f);
    const NF Kc = need_kc ? conductivity_hydraulic<NF, HYD, false>(p, c.liq, f) : NF(0);
    // neighbours by DPP shifts (executed by all lanes, never inside a divergent select)
    const NF T_sh = shfl_up1<NF, LPC>(c.T), kap_sh = shfl_up1<NF, LPC>(kap);
    // temperature halos: every condition that is not set costs one wave-uniform branch, every condition that is set is
    // computed by all lanes and kept by the edge lane
    NF T_ext_b = c.T, T_ext_t = c.T;
    // (div_const_nsz: a zero gradient's sign reaches T_ext only when T itself is a zero, then the face flux q_T as a signed zero,
    // and `0 + (-(dq * rdz))` below gives +0 whatever its sign -- or a non-zero neighbour flux absorbs it)
    if (v.bc.kind[2][0] == 1) T_ext_b = c.T + div_const_nsz(c.T - bTb, v.g.hdzf_bot, v.g.rhdzf_bot) * (-v.g.dzf_bot);
    if (v.bc.kind[2][1] == 1) T_ext_t = c.T + div_const_nsz(bTt - c.T, v.g.hdzf_top, v.g.rhdzf_top) * v.g.dzf_top;
    const NF T_m = is_bot ? T_ext_b : T_sh;
    const NF T_h = T_ext_t;
    // liquid fraction / saturation / pressure head carry the default condition (halo = edge cell): the halo cell's
    // conductivity is the edge cell's, bit for bit -- except under NoFlow with the reference's never-filled saturation
    // halo (SURVEY C-1), where the halo cell is dry
    NF kap_halo = kap;
    if (!RICHARDS && p.halo_policy != 1) kap_halo = conductivity(p, fractions(p, NF(0), c.liq, viol));
    const NF kap_m = is_bot ? kap_halo : kap_sh;
    const NF kap_h = kap_halo;
    // heat: every lane forms its lower face, the top lane also the boundary face (soil_energy.jl:112-149)
    const NF qT_lo = -(NF(0.5) * (kap + kap_m)) * ((c.T - T_m) * L.rdzf_lo);
    const NF qT_sh = shfl_dn1<NF, LPC>(qT_lo);
    const NF qT_hi = is_top ? -(NF(0.5) * (kap_h + kap)) * ((T_h - c.T) * L.rdzf_hi) : qT_sh;
    Tendency<NF> t;
    t.gU = NF(0) + (-((qT_hi - qT_lo) * L.rdzc));
    t.gS = NF(0);
    t.Kf_lo = NF(0);
    t.Kc = Kc;
    if (need_kc) {   // face conductivities (soil_hydrology.jl:145-163)
        const NF Kc_m = shfl_up1<NF, LPC>(Kc);
        const NF Kmin = jl_min(Kc, Kc_m);
        t.Kf_lo = (is_bot || is_top) ? Kc : Kmin;
    }
    if (RICHARDS) {  // Darcy fluxes (soil_hydrology_rre.jl:95-131)
        const NF Kf_lo = t.Kf_lo;
        const NF Kf_up = shfl_up1<NF, LPC>(Kf_lo), Kf_dn = shfl_dn1<NF, LPC>(Kf_lo), psi_sh = shfl_up1<NF, LPC>(c.psi);
        // The halo face below the bottom cell (never written: 0) would be selected by a NEGATIVE head gradient only; with the
        // default condition the bottom lane's gradient is (psi - psi) * rdz = +0 or NaN, never negative: its value is never
        // taken, and the select that put the 0 there is gone (the lane receives its neighbour column's top face instead).
        const NF Kf_m = Kf_up;
        const NF Kf_p = is_top ? Kc : Kf_dn;      // face Nz repeats the top cell's value
        const NF psi_m = is_bot ? c.psi : psi_sh;
        const NF g_lo = (c.psi - psi_m) * L.rdzf_lo;
        const NF Ks_lo = upwind_conductivity(g_lo, Kf_m, Kf_lo, Kf_p);
        const NF qW_lo = -Ks_lo * g_lo;
        const NF qW_sh = shfl_dn1<NF, LPC>(qW_lo);
        // boundary face above the top cell, default condition: the halo cell repeats psi, so the head difference is +0
        // (NaN for a non-finite psi), never negative: K* = min(K, K_halo_face = 0)
        const NF zero_or_nan = c.psi - c.psi;
        const NF qW_t = -jl_min(Kc, NF(0)) * zero_or_nan;
        const NF qW_hi = is_top ? qW_t : qW_sh;
        const NF dtheta = -((qW_hi - qW_lo) * L.rdzc) + NF(0) + p.vwc_forcing;
        t.gS = NF(0) + div_const_nsz(dtheta, p.por, p.rpor);       // (`0 + q`: +0 for a zero quotient of either sign)
    }
    return t;
}

// compute_z_bcs! + explicit_step! + the hydrology closure's repair and water table for the column in registers.
// (gU, gS): tendencies WITHOUT the boundary flux terms (they are added here).  Returns the new (U, sat) in `n`, the
// water table in z0 and, in the top lane, the column's overflow into surface_excess_water (0 elsewhere).
template <class NF, bool RICHARDS, int LPC>
TRM_DEV NF column_advance(const View<NF>& v, const LevelGeom<NF>& L, const LaneInfo& ln, int Nz, const ColumnBC<NF>& bc,
                          NF U0, NF sat0, NF& gU, NF& gS, NF dt, Cell<NF>& n, NF& z0, bool& bad) {
    if (bc.has_U) gU += bc.flux_U;
    n.U = U0 + gU * dt;
    if (!RICHARDS) bad = bad || (ln.act && is_nan(n.U));
    n.sat = sat0;
    z0 = NF(0);
    NF over = NF(0);
    if (RICHARDS) {
        if (bc.has_S) gS += bc.flux_S;
        NF snew = sat0 + gS * dt;
        bad = bad || (ln.act && __builtin_isunordered(n.U, snew));   // either one NaN: one compare
        over = repair_saturation<NF, LPC>(v, snew, ln.k, Nz, ln.m_act, ln.is_bot, ln.is_top, L, ln.m_bot, ln.m_top);
        z0 = water_table<NF, LPC>(snew, ln.m_act, ln.lane, L);
        n.sat = snew;
    }
    return over;
}

// (n.sat comes out of column_advance: under Richards it has been through the repair)
template <class NF, bool RICHARDS, int HYD>
TRM_DEV Frac<NF> column_closure(const DevParams<NF>& p, const LevelGeom<NF>& L, NF z0, Cell<NF>& n, uint32_t& viol) {
    const Frac<NF> f = energy_closure_wave<NF, RICHARDS ? 2 : 1>(p, n.U, n.sat, n.liq, n.T, viol);
    n.psi = RICHARDS ? pressure_head<NF, HYD, true>(p, n.sat, L.zC, L.psiz, z0) : NF(0);
    return f;
}

// LandModel inside the program (PROG_MULTI): the 0-D surface processes of the column, evaluated by its top lane from
// the registers (land_model.jl:79-88) -- what k_surface<FROM_STATE> does in front of a per-step launch.
template <class NF> struct SurfaceRegs {
    SebIn<NF> in;
    SebOut<NF> out;   // out.Ts is the prognostic skin temperature
};

// Time series inside the multi-step program: update_inputs! (input_sources.jl:162-168) for every step of the launch.
// What a per-step launch reads from an array that k_interp_series has just filled, the program interpolates itself:
// the host lays out, per step and series, the bracketing nodes and the fraction (the clock is the host's), the kernel
// reads the two node values of its column and applies the same formula.
enum { SLOT_T_BOT = 0, SLOT_T_TOP, SLOT_FU_BOT, SLOT_FU_TOP, SLOT_FS_BOT, SLOT_FS_TOP, SLOT_TAIR, SLOT_PRES, SLOT_WIND, SLOT_QAIR,
       SLOT_RAIN, SLOT_SWD, SLOT_LWD, SLOT_ALBEDO, SLOT_EMISSIVITY, SLOT_COUNT };
template <class NF> struct SeriesTable {
    const NF* base[SLOT_COUNT];   // [nt][Nh] node values of the series feeding the slot, or null
    NF* dst[SLOT_COUNT];          // where update_inputs! leaves the evaluated values (written back after the last step)
    int row_of[SLOT_COUNT];       // column of the per-step rows that belongs to the slot
    int raster[SLOT_COUNT];       // the Raster source's formula instead of FieldTimeSeries' (k_interp_series)
};
struct SeriesRow { long long n1, n2; double f, g; };   // element offsets n * Nh of the two nodes, fraction (eps, dt for rasters)
template <class NF> TRM_DEV NF series_value(const SeriesTable<NF>* tb, const SeriesRow* rows, int slot, int ii) {
    const SeriesRow r = rows[tb->row_of[slot]];
    const NF* base = tb->base[slot];
    const NF x1 = base[r.n1 + ii];
    if (r.n1 == r.n2) return x1;
    const NF x2 = base[r.n2 + ii];
    if (tb->raster[slot]) return (NF)((double)x1 + r.f * (double)(NF)(x2 - x1) / r.g);
    return (NF)((double)x2 * r.f + (double)x1 * (1.0 - r.f));
}

// the six granules of one column as the scalar path delivers them (see FrontArgs)
struct FrontGranules {
    unsigned long long w[FRONT_GRANULES];
    TRM_DEV void load(const unsigned long long* gran, unsigned byte_off_uniform) {     // (adjacent scalar loads: the backend widens them)
        for (int n = 0; n < FRONT_GRANULES; ++n) w[n] = sld_off<unsigned long long>(gran, byte_off_uniform + (unsigned)n * 8u);
    }
    // (integer arithmetic on purpose: a chain of `&&` over the six compares was compiled into vector selects, shifts and
    //  v_readfirstlane -- ~25 vector and ~40 scalar instructions per wave for what is 12 s_xor / s_or)
    TRM_DEV unsigned mismatch(unsigned epoch) const {
        unsigned bad = 0;
        for (int n = 0; n < FRONT_GRANULES; ++n) bad |= (unsigned)(w[n] >> 32) ^ epoch;
        asm volatile("" : "+s"(bad));      // (or else `bad == 0` is folded back into a conjunction of compares, and those into vector selects)
        return bad;
    }
    TRM_DEV double value(int q) const { return __builtin_bit_cast(double, (w[q + 1] << 32) | (w[q] & 0xffffffffull)); }
};
template <class NF> struct ColumnArgs {
    NF dt;
    int finalize, write_kf, nsteps;
    // Heun: the stage's temperature boundary values (a series evaluated at t + dt), else the state's
    const NF *bcT_bot_stage, *bcT_top_stage;
    // multi-step program with time series: the slot table and [nsteps][nseries] rows
    const SeriesTable<NF>* series;
    const SeriesRow* series_rows;
    int nseries;
    // Heun of the vegetation-coupled LandModel: the stage's saturation, liquid fraction, temperature ([Nh][Nzp]) and surface
    // excess water ([Nh]) are stored for the 0-D processes that are evaluated AT the stage (null otherwise)
    NF *stage_sat, *stage_liq, *stage_T, *stage_S;
};



#ifndef TRM_PICK_EXEC
#define TRM_PICK_EXEC 1
#endif
#ifndef TRM_COLUMN_WAVES_EULER
#define TRM_COLUMN_WAVES_EULER 1
#endif
// The program itself, as a device function: `k_column` below is its launch; `k_land_euler` runs it beside the surface processes
// of OTHER columns in one launch.  Its kernel must pass (View, DevParams, ColumnArgs) as its first three arguments: the program
// re-reads them from the kernarg segment at those offsets (kernarg_reload).  `block`: index of the 256-thread workgroup among
// those that run the program.
// STAGED / SCALAR_IN: how the per-column outputs leave and the per-column inputs arrive (see below) -- compile-time: as
// wave-uniform run-time branches they cost the field loads their back-to-back issue (profiles/r03/exp28: 8 x N145 +7 %).
// FRONT (LandModel, BCSIG_LAND): ground heat flux, infiltration and the new skin temperature come from the surface workgroups of THIS
// launch (surface_front, trm_kernels.hpp) as granules; the kernel's fourth argument is then a FrontArgs.
template <class NF, bool RICHARDS, int HYD, int LPC, int DERIVE, int PROG, bool SEB_INLINE, bool SERIES = false, bool STAGED = false, bool SCALAR_IN = true, int BCSIG = BCSIG_RUNTIME, bool FRONT = false>
TRM_DEV void column_program(const View<NF>& v_arg, const DevParams<NF>& p_arg, const ColumnArgs<NF>& a, unsigned block) {
    // (kernarg layout: the arguments in order, each at its natural alignment)
    constexpr unsigned off_p = round_up_to((unsigned)sizeof(View<NF>), (unsigned)alignof(DevParams<NF>));
    constexpr unsigned off_args = round_up_to(off_p + (unsigned)sizeof(DevParams<NF>), (unsigned)alignof(ColumnArgs<NF>));
    constexpr unsigned off_front = round_up_to(off_args + (unsigned)sizeof(ColumnArgs<NF>), (unsigned)alignof(FrontArgs));
    static_assert(!FRONT || (PROG != PROG_MULTI && BCSIG == BCSIG_LAND && RICHARDS && !SEB_INLINE && sizeof(NF) == 8), "the in-launch surface processes feed the per-step fp64 LandModel programs");
    const View<NF>& v = v_arg;
    const DevParams<NF>& p = p_arg;
    static_assert(!SEB_INLINE || PROG == PROG_MULTI, "the in-kernel surface energy balance belongs to the multi-step program");
    static_assert(!SERIES || PROG == PROG_MULTI, "in-kernel time series belong to the multi-step program");
    constexpr int CPW = 64 / LPC;
    TRM_PHASE("addressing");
    LaneInfo ln;
    ln.lane = threadIdx.x & 63;
#if TRM_CUT_MASKS
    // (the wave index and everything that follows from it alone lives on the scalar unit)
    const int wave = __builtin_amdgcn_readfirstlane((int)((block * (unsigned)TRM_STEP_BLOCK + threadIdx.x) >> 6));
    ln.k = ln.lane % LPC;
    const int sub = ln.lane / LPC;
    const int Nz = v.Nz, Nh = (int)v.Nh;
    // which lanes hold the bottom / top cell, a real cell, the wave's second column: wave-uniform masks, no lane-wise compare
    const bool upper = CPW == 2 && lane_in(0xffffffff00000000ull);
    ln.m_bot = level_lanes<LPC>(0);
    ln.m_top = level_lanes<LPC>(Nz - 1);
    ln.is_bot = lane_in(ln.m_bot);
    ln.is_top = lane_in(ln.m_top);
    const NF dt = a.dt;
    const int finalize = a.finalize, write_kf = a.write_kf;
    const bool need_kc = RICHARDS || write_kf;

    const int col_first = wave * CPW;                              // (uniform) first column of the wave
    const int i = col_first + sub;
    const unsigned long long m_col = CPW == 1 ? (col_first < Nh ? ~0ull : 0ull)
                                              : ((col_first < Nh ? 0x00000000ffffffffull : 0ull) | (col_first + 1 < Nh ? 0xffffffff00000000ull : 0ull));
    ln.m_act = m_col & levels_below<LPC>(Nz);
    ln.act = lane_in(ln.m_act);
    const int ii = i < Nh ? i : Nh - 1;
#else      // (A/B builds: round 3's lane-wise form)
    const int wave = (int)((block * (unsigned)TRM_STEP_BLOCK + threadIdx.x) >> 6);
    ln.k = ln.lane % LPC;
    const int sub = ln.lane / LPC;
    const int Nz = v.Nz, Nh = (int)v.Nh;
    const bool upper = sub != 0;
    ln.is_bot = ln.k == 0;
    ln.is_top = ln.k == Nz - 1;
    ln.m_bot = wave_ballot(ln.is_bot);
    ln.m_top = wave_ballot(ln.is_top);
    const NF dt = a.dt;
    const int finalize = a.finalize, write_kf = a.write_kf;
    const bool need_kc = RICHARDS || write_kf;
    const int i = wave * CPW + sub;
    const bool colok = i < Nh;
    ln.act = colok && ln.k < Nz;
    ln.m_act = wave_ballot(colok) & wave_ballot(ln.k < Nz);
    const int ii = colok ? i : Nh - 1;
    const int col_first = __builtin_amdgcn_readfirstlane(wave * CPW);
#endif
    const unsigned ib0 = (unsigned)ii * (unsigned)sizeof(NF);
    const unsigned cb0 = ((unsigned)ii * (unsigned)v.Nzp + (unsigned)(ln.k < Nz ? ln.k : Nz - 1)) * (unsigned)sizeof(NF);
    uint32_t viol = 0;
    bool bad = false;

    // ---- the column comes in: 5 (3 with DERIVE) coalesced reads -------------------------------------------------------
    TRM_PHASE("loads+derive");
    Cell<NF> c;
    c.U = ldg(v.U, cb0);
    c.sat = ldg(v.sat, cb0);
    static_assert(DERIVE == DERIVE_NONE || DERIVE == DERIVE_T_LIQ, "the column program reads T / liq or derives both (the liquid fraction alone and the "
                  "pressure head as well were measured and lost: EXPERIMENTS.md; their instances were removed in round 5)");
    constexpr bool DERIVE_TL = DERIVE == DERIVE_T_LIQ;
    c.psi = RICHARDS ? ldg(v.psi, cb0) : NF(0);
    if (!DERIVE_TL) { c.T = ldg(v.T, cb0); c.liq = ldg(v.liq, cb0); }
    const LevelGeom<NF> L = level_geom(v, ln.k);      // (behind the field loads: see level_geom)
    // (BCSIG >= 0: the launcher has matched the context's kinds against the signature -- constants from here on)
    constexpr bool SIG = BCSIG >= 0;
    const bool seb = SIG ? (BCSIG & BCSIG_LAND) != 0 : p.seb != 0;
    const bool vTb = SIG ? (BCSIG & BCSIG_T_BOT) != 0 : v.bc.kind[2][0] == 1, vTt = SIG ? (BCSIG & BCSIG_T_TOP) != 0 : v.bc.kind[2][1] == 1;
    const bool bU = SIG ? (BCSIG & BCSIG_FU_BOT) != 0 : v.bc.kind[0][0] == 2, bS = RICHARDS && (SIG ? (BCSIG & BCSIG_FS_BOT) != 0 : v.bc.kind[1][0] == 2);
    const bool tU = !SEB_INLINE && (seb || (SIG ? (BCSIG & BCSIG_FU_TOP) != 0 : v.bc.kind[0][1] == 2));
    const bool tS = RICHARDS && !SEB_INLINE && (seb || (SIG ? (BCSIG & BCSIG_FS_TOP) != 0 : v.bc.kind[1][1] == 2));
    // The per-column inputs (boundary values, LandModel's ground heat flux / infiltration, the 0-D fields) come through the
    // scalar memory path (sld): one s_load per column of the wave, selected per half-wave -- where the state is cache-resident
    // (SCALAR_IN, chosen per launch; profiles/r03/exp26: C4 34.1 -> 32.7 us, vegetation-coupled 48.2 -> 46.0, C3 25.3 -> 24.8, but
    // 8 x N145 200.8 -> 208.0: from HBM the scalar cache's 64-byte lines for 16 useful bytes cost more than the vector path).
    const unsigned jo0 = (unsigned)(col_first < Nh ? col_first : Nh - 1) * (unsigned)sizeof(NF), jo1 = (unsigned)(col_first + 1 < Nh ? col_first + 1 : Nh - 1) * (unsigned)sizeof(NF);
    // (a request and its use are two steps: on the scalar path the select between the wave's two columns is a VECTOR instruction that
    // needs both values -- formed where the value is requested, inside the branch of a condition that is set, it puts a wait for
    // the scalar load right behind it, one exposed trip to L2 per condition)
    struct ColVal { NF x0, x1; };
    // (with the kinds read at run time every request sits in a branch and eleven pending pairs would not fit the scalar registers:
    // that instance resolves each value where it requests it, as round 3 did)
    constexpr bool DEFER = SCALAR_IN && CPW == 2 && BCSIG >= 0;
    auto col_req = [&](const NF* ptr) -> ColVal {
        if (!SCALAR_IN) return ColVal{ldg(ptr, ib0), NF(0)};
        const NF x0 = sld_off<NF>(ptr, jo0);
        if (CPW == 1) return ColVal{x0, NF(0)};
        const NF x1 = sld_off<NF>(ptr, jo1);
        if (DEFER) return ColVal{x0, x1};
        return ColVal{upper ? x1 : x0, NF(0)};
    };
    // The select between the wave's two columns, both values wave-uniform: `upper ? x1 : x0` is two v_mov (scalar -> vector, the
    // select takes one scalar operand: its mask) and a v_cndmask per 32-bit half.  Writing x0 to every lane and then x1 to the upper
    // half-wave under an execution mask is two v_mov per half (TRM_PICK_EXEC 0: the select, A/B).
    auto pick_column = [&](NF x0, NF x1) -> NF {
#if TRM_PICK_EXEC
        if (__builtin_constant_p(x0 == x1) && x0 == x1) return x0;      // (a condition that is not set: both halves the same constant)
        if constexpr (sizeof(NF) == 8) {
            const unsigned long long b0 = __builtin_bit_cast(unsigned long long, x0), b1 = __builtin_bit_cast(unsigned long long, x1);
            unsigned lo = (unsigned)b0, hi = (unsigned)(b0 >> 32);
            const unsigned lo1 = (unsigned)b1, hi1 = (unsigned)(b1 >> 32);
            unsigned long long save;
            asm("s_mov_b64 %[save], exec\n\ts_and_b64 exec, %[save], %[m]\n\tv_mov_b32 %[lo], %[slo]\n\tv_mov_b32 %[hi], %[shi]\n\ts_mov_b64 exec, %[save]"
                         : [lo] "+v"(lo), [hi] "+v"(hi), [save] "=&s"(save)
                         : [slo] "s"(lo1), [shi] "s"(hi1), [m] "s"(0xffffffff00000000ull)
                         : "scc");
            return __builtin_bit_cast(NF, ((unsigned long long)hi << 32) | lo);
        } else
#endif
        return upper ? x1 : x0;
    };
    auto col_get = [&](const ColVal& q) -> NF { return DEFER ? pick_column(q.x0, q.x1) : q.x0; };
    // ALL of them are requested HERE, behind the field loads and in front of the derivation: the derivation waits for U and sat and
    // branches (the phase-change divide), and a load issued behind it starts its trip to memory only then -- a second full memory
    // latency in every wave's life (round 4: in the round-3 order the boundary values of the HBM-resident step were requested ~110
    // instructions after the fields).  surface_excess_water and the skin temperature likewise: vector memory retires in order, loads and
    // stores through the one counter, so a load issued behind the field stores would hold the wave until its stores were acknowledged.
    const ColVal none{NF(0), NF(0)};
    ColVal q_Tb = none, q_Tt = none, q_Ub = none, q_Sb = none, q_Ut = none, q_St = none, q_Tb2 = none, q_Tt2 = none, q_S = none, q_Ts = none;
    NF S_stage_out = NF(0);
    FrontGranules fg0{}, fg1{};
    unsigned front_epoch = 0;
    auto request_inputs = [&] {
        if (vTb) q_Tb = col_req(bcval(v, 2, 0));
        if (vTt) q_Tt = col_req(bcval(v, 2, 1));
        if (bU) q_Ub = col_req(bcval(v, 0, 0));
        if (bS) q_Sb = col_req(bcval(v, 1, 0));
        // LandModel wires ground_heat_flux / -infiltration (land_model.jl:56-61), produced by k_surface just before this launch
        if (tU && !FRONT) q_Ut = col_req(seb ? v.ghf : bcval(v, 0, 1));
        if (tS && !FRONT) q_St = col_req(seb ? v.infil : bcval(v, 1, 1));
        if (PROG == PROG_HEUN) {      // the stage's temperature boundary values, taken at t + dt (heun.jl:52-59)
            if (vTb) q_Tb2 = col_req(a.bcT_bot_stage);
            if (vTt) q_Tt2 = col_req(a.bcT_top_stage);
        }
        if (PROG != PROG_MULTI) {
            if (RICHARDS) q_S = col_req(v.S);
            if (seb && !FRONT) q_Ts = col_req(v.Ts);
        }
        if constexpr (FRONT) {   // the granules of the wave's columns through the scalar path: valid if the surface workgroups have published them
            const FrontArgs& fa = kernarg_reload<FrontArgs>(off_front);
            fg0.load(fa.gran, jo0 * (unsigned)(FRONT_GRANULES * sizeof(unsigned long long) / sizeof(NF)));
            if (CPW == 2) fg1.load(fa.gran, jo1 * (unsigned)(FRONT_GRANULES * sizeof(unsigned long long) / sizeof(NF)));
            front_epoch = fa.epoch;
        }
    };
    // ... on the VECTOR path only.  Scalar loads return out of order, so every use of one waits for ALL of them (lgkmcnt(0)): requested
    // early, the per-column values -- from L2 or beyond -- hold up the parameter reloads of the derivation, which hit the scalar cache
    // (profiles/r04/exp8: early / late on one box, 8 x N145 [vector path] 0.981 / 0.997 of round 3's time, C3 [scalar path] 0.960 / 0.929).
    constexpr bool EARLY = TRM_EARLY_INPUTS && !SCALAR_IN;
    if (EARLY) request_inputs();
    Frac<NF> f_in{};          // the incoming cell's volumetric fractions, when the derivation has formed them
    if (DERIVE_TL) {
        uint32_t viol_in = 0;
        f_in = energy_closure_wave<NF, 0>(kernarg_reload<DevParams<NF>>(off_p), c.U, c.sat, c.liq, c.T, viol_in);   // (its scalars die right here)
    }
    // ---- boundary inputs of the column --------------------------------------------------------------------------------
    TRM_PHASE_FENCE("inputs", c.U, c.sat, c.psi, c.T, c.liq);
    if (!EARLY) request_inputs();
    const NF in_Tb = col_get(q_Tb), in_Tt = col_get(q_Tt), in_Ub = col_get(q_Ub), in_Sb = col_get(q_Sb);
    NF in_Ut = col_get(q_Ut), in_St = col_get(q_St);
    const NF in_Tb2 = col_get(q_Tb2), in_Tt2 = col_get(q_Tt2);
    NF S_in = col_get(q_S), Ts_in = col_get(q_Ts);
    NF front_Ut = NF(0), front_St = NF(0);
    bool front_ready = true;
    if constexpr (FRONT) {
        front_ready = (TRM_FRONT_DIAG & 4) || (fg0.mismatch(front_epoch) | (CPW == 1 ? 0u : fg1.mismatch(front_epoch))) == 0u;      // (wave-uniform, on the scalar unit)
        if (CPW == 2) {
            front_Ut = pick_column(fg0.value(FRONT_GHF), fg1.value(FRONT_GHF)); front_St = pick_column(fg0.value(FRONT_INFIL), fg1.value(FRONT_INFIL));
            Ts_in = pick_column(fg0.value(FRONT_TS), fg1.value(FRONT_TS));
        } else {
            front_Ut = fg0.value(FRONT_GHF); front_St = fg0.value(FRONT_INFIL); Ts_in = fg0.value(FRONT_TS);
        }
    }
    ColumnBC<NF> bc;
    if (FRONT) { in_Ut = front_Ut; in_St = front_St; }
    bc.bTb = in_Tb;
    bc.bTt = in_Tt;
    {   // flux conditions: a term for the edge lane of every condition that is SET (wave-uniform branches), nothing otherwise.
        // (flux_term_*_nsz: the term is added to a tendency that is never -0.0, so the sign of a zero term is immaterial)
        // Top terms enter with a minus sign.
        NF fU = NF(0), fS = NF(0);
#if TRM_CUT_FLUX
        if (bU) { const NF e = flux_term_bottom_nsz(in_Ub, v.g); fU = ln.is_bot ? e : fU; }
        if (bS) { const NF e = flux_term_bottom_nsz(in_Sb, v.g); fS = ln.is_bot ? e : fS; }
        if (tU) { const NF e = -flux_term_top_nsz(in_Ut, v.g); fU = ln.is_top ? e : fU; }
        if (tS) { const NF e = -flux_term_top_nsz(seb ? -in_St : in_St, v.g); fS = ln.is_top ? e : fS; }
#endif
#if !TRM_CUT_FLUX
        {   // both edge terms always formed, two selects per variable: the form measured faster (no select inside the branches,
            // no branch around the `+ flux` of column_advance)
            NF eU_b = NF(0), eU_t = NF(0), eS_b = NF(0), eS_t = NF(0);
            if (bU) eU_b = flux_term_bottom_nsz(in_Ub, v.g);
            if (bS) eS_b = flux_term_bottom_nsz(in_Sb, v.g);
            if (tU) eU_t = -flux_term_top_nsz(in_Ut, v.g);
            if (tS) eS_t = -flux_term_top_nsz(seb ? -in_St : in_St, v.g);
            // (a side whose condition is known at compile time not to be set needs no select of its own: the bottom and the top cell are
            //  different lanes, Nz >= 2)
            const NF tU_term = ln.is_top ? eU_t : NF(0), tS_term = ln.is_top ? eS_t : NF(0);
            fU = (BCSIG >= 0 && !bU) ? tU_term : (ln.is_bot ? eU_b : tU_term);
            fS = (BCSIG >= 0 && !bS) ? tS_term : (ln.is_bot ? eS_b : tS_term);
        }
#endif
        bc.flux_U = fU;
        bc.flux_S = fS;
        // (the multi-step program may receive its terms later: from a series, from the inline surface energy balance)
        bc.has_U = !TRM_CUT_FLUX || PROG == PROG_MULTI || bU || tU;
        bc.has_S = !TRM_CUT_FLUX || PROG == PROG_MULTI || bS || tS;
    }
    NF S = NF(0);
    SurfaceRegs<NF> sf;
    if (PROG == PROG_MULTI) {
        if (RICHARDS) S = ldg(v.S, ib0);
        if (SEB_INLINE) {
            sf.in = SebIn<NF>{ldg(v.Tair, ib0), ldg(v.pres, ib0), ldg(v.wind, ib0), ldg(v.qair, ib0), ldg(v.rain, ib0), ldg(v.swd, ib0), ldg(v.lwd, ib0), NF(0), NF(0), NF(0)};
            seb_radiation_inputs(p, v.albedo, v.emissivity, ib0, sf.in);
            sf.out.Ts = ldg(v.Ts, ib0);
        }
    }

    // FRONT: the values are first needed by the explicit step (of the stage under Heun); if the scalar read at the top came before the
    // surface workgroups had published, the wave polls its granules here
    auto front_wait = [&] {
        if constexpr (FRONT && !(TRM_FRONT_DIAG & 2)) if (!front_ready) {
            // the scalar read came before the surface workgroups had published: poll the granules (vector loads at agent scope,
            // every lane k < 6 of a column its granule k), here, where the values are first needed
            TRM_PHASE("rare+ granule poll");
            const FrontArgs& fa = kernarg_reload<FrontArgs>(off_front);
            const unsigned long long* gp = fa.gran + (size_t)ii * FRONT_GRANULES + (ln.k < FRONT_GRANULES ? ln.k : FRONT_GRANULES - 1);
            unsigned long long g = 0;
            bool ok = false;
            for (int spin = 0; spin < FRONT_SPIN_LIMIT && !ok; ++spin) {
                g = ld_agent(gp);
                ok = wave_ballot((unsigned)(g >> 32) != fa.epoch) == 0ull;
                if (!ok) __builtin_amdgcn_s_sleep(TRM_FRONT_POLL_SLEEP);
            }
            const int w = (int)(unsigned)g;                 // the granule's payload: one 32-bit half of a value
            auto value = [&](int q, int base) {
                const unsigned lo = (unsigned)__builtin_amdgcn_readlane(w, base + q), hi = (unsigned)__builtin_amdgcn_readlane(w, base + q + 1);
                return __builtin_bit_cast(NF, ((unsigned long long)hi << 32) | lo);
            };
            NF ut = value(FRONT_GHF, 0), st = value(FRONT_INFIL, 0), ts = value(FRONT_TS, 0);
            if (CPW == 2) {
                const NF ut1 = value(FRONT_GHF, LPC), st1 = value(FRONT_INFIL, LPC), ts1 = value(FRONT_TS, LPC);
                ut = pick_column(ut, ut1); st = pick_column(st, st1); ts = pick_column(ts, ts1);
            }
            if (!ok) {   // gave up: no hang, the columns of the wave are flagged and NaN
                viol |= 4u;
                ut = st = ts = __builtin_nan("");
            }
            Ts_in = ts;
            const NF eU_t = -flux_term_top_nsz(ut, v.g), eS_t = -flux_term_top_nsz(-st, v.g);
            bc.flux_U = ln.is_top ? eU_t : NF(0);
            bc.flux_S = ln.is_top ? eS_t : NF(0);
            TRM_PHASE("rare-");
        }
    };
    Cell<NF> n = c;          // the state after the program's last step
    Tendency<NF> t{};        // tendencies of the last evaluation at the STATE (hydraulic_conductivity comes from here)
    NF gU_out = NF(0), gS_out = NF(0), GS_out = NF(0), z0 = NF(0);

    NF over = NF(0), over_stage = NF(0);   // (top lane) overflow of the column into surface_excess_water
    Frac<NF> f_new{};                      // volumetric fractions of the new state (its closure forms them)
    if (PROG == PROG_HEUN) {
        // stage 1: tendencies at the state, Euler predictor (with the state's boundary fluxes) and its closures
        t = column_tendencies<NF, RICHARDS, HYD, LPC>(v, p, L, ln, c, bc.bTb, bc.bTt, need_kc, viol, DERIVE_TL ? &f_in : nullptr);
        const NF G1U = t.gU, G1S = t.gS;
        NF gU = G1U, gS = G1S, z0s;
        Cell<NF> s;
        front_wait();
        over_stage = column_advance<NF, RICHARDS, LPC>(v, L, ln, Nz, bc, c.U, c.sat, gU, gS, dt, s, z0s, bad);
        const Frac<NF> f_stage = column_closure<NF, RICHARDS, HYD>(kernarg_reload<DevParams<NF>>(off_p), L, z0s, s, viol);
        if (a.stage_T && ln.act) {   // (wave-uniform: the stage leaves the registers only for the coupled vegetation)
            const unsigned cb = block_local(cb0);
            stg(a.stage_sat, cb, s.sat);
            stg(a.stage_liq, cb, s.liq);
            stg(a.stage_T, cb, s.T);
        }
        // stage 2: tendencies at the stage, its temperature boundary values taken at t + dt (heun.jl:52-59)
        const NF bTb2 = in_Tb2, bTt2 = in_Tt2;
        uint32_t viol_stage = 0;
        const Tendency<NF> t2 = column_tendencies<NF, RICHARDS, HYD, LPC>(kernarg_reload<View<NF>>(0), kernarg_reload<DevParams<NF>>(off_p), L, ln, s, bTb2, bTt2, RICHARDS, viol_stage, &f_stage);
        viol |= viol_stage;
        // average_tendencies! (heun.jl:27-35), then the step of the STATE with its own boundary fluxes
        gU = (G1U + t2.gU) / NF(2);
        gS = RICHARDS ? (G1S + t2.gS) / NF(2) : NF(0);
        over = column_advance<NF, RICHARDS, LPC>(v, L, ln, Nz, bc, c.U, c.sat, gU, gS, dt, n, z0, bad);
        f_new = column_closure<NF, RICHARDS, HYD>(kernarg_reload<DevParams<NF>>(off_p), L, z0, n, viol);
        gU_out = gU; gS_out = gS;
    } else {
        const int nsteps = PROG == PROG_MULTI ? a.nsteps : 1;
        for (int step = 0; step < nsteps; ++step) {
            // (the multi-step loop reads its kernel arguments afresh every iteration, see kernarg_reload)
            const View<NF>& v = PROG == PROG_MULTI ? kernarg_reload<View<NF>>(0) : v_arg;
            const DevParams<NF>& p = PROG == PROG_MULTI ? kernarg_reload<DevParams<NF>>(off_p) : p_arg;
            if (PROG == PROG_MULTI && step > 0) c = n;
            if (SERIES) {
                // update_inputs!(state, clock) of this step, for the slots a series feeds (wave-uniform branches)
                const SeriesTable<NF>* tb = a.series;
                const SeriesRow* rows = a.series_rows + (size_t)step * (size_t)a.nseries;
                if (tb->base[SLOT_T_BOT]) bc.bTb = series_value(tb, rows, SLOT_T_BOT, ii);
                if (tb->base[SLOT_T_TOP]) bc.bTt = series_value(tb, rows, SLOT_T_TOP, ii);
                if (tb->base[SLOT_FU_BOT]) { const NF e = flux_term_bottom(series_value(tb, rows, SLOT_FU_BOT, ii), v.g); if (ln.is_bot) bc.flux_U = e; }
                if (RICHARDS && tb->base[SLOT_FS_BOT]) { const NF e = flux_term_bottom(series_value(tb, rows, SLOT_FS_BOT, ii), v.g); if (ln.is_bot) bc.flux_S = e; }
                if (!SEB_INLINE) {
                    if (tb->base[SLOT_FU_TOP]) { const NF e = -flux_term_top(series_value(tb, rows, SLOT_FU_TOP, ii), v.g); if (ln.is_top) bc.flux_U = e; }
                    if (RICHARDS && tb->base[SLOT_FS_TOP]) { const NF e = -flux_term_top(series_value(tb, rows, SLOT_FS_TOP, ii), v.g); if (ln.is_top) bc.flux_S = e; }
                } else {
                    if (tb->base[SLOT_TAIR]) sf.in.Tair = series_value(tb, rows, SLOT_TAIR, ii);
                    if (tb->base[SLOT_PRES]) sf.in.pres = series_value(tb, rows, SLOT_PRES, ii);
                    if (tb->base[SLOT_WIND]) sf.in.wind = series_value(tb, rows, SLOT_WIND, ii);
                    if (tb->base[SLOT_QAIR]) sf.in.qair = series_value(tb, rows, SLOT_QAIR, ii);
                    if (tb->base[SLOT_RAIN]) sf.in.rain = series_value(tb, rows, SLOT_RAIN, ii);
                    if (tb->base[SLOT_SWD]) sf.in.swd = series_value(tb, rows, SLOT_SWD, ii);
                    if (tb->base[SLOT_LWD]) sf.in.lwd = series_value(tb, rows, SLOT_LWD, ii);
                    if (p.prescribed_albedo && (tb->base[SLOT_ALBEDO] || tb->base[SLOT_EMISSIVITY])) {
                        const NF al = tb->base[SLOT_ALBEDO] ? series_value(tb, rows, SLOT_ALBEDO, ii) : sf.in.albedo;
                        if (tb->base[SLOT_ALBEDO]) sf.in.albedo = al;
                        if (tb->base[SLOT_EMISSIVITY]) {
                            const NF e = series_value(tb, rows, SLOT_EMISSIVITY, ii);
                            sf.in.eps_sigma = e * p.sigma;
                            sf.in.one_minus_emissivity = NF(1) - e;
                        }
                    }
                }
                if (step == nsteps - 1 && ln.act && ln.is_top) {
                    // the input fields / boundary value arrays keep what the last step evaluated, as after update_inputs!
                    for (int s = 0; s < SLOT_COUNT; ++s)
                        if (tb->base[s]) tb->dst[s][ii] = series_value(tb, rows, s, ii);
                }
            }
            if (SEB_INLINE) {
                // compute_auxiliary! of the surface processes from the top cell in registers (k_surface<FROM_STATE>)
                uint32_t viol_s = 0;
                (void)viol_s;
                const NF Kf_top = conductivity_hydraulic<NF, HYD, false>(p, c.liq, step > 0 ? f_new : fractions_unchecked(p, c.sat, c.liq));
                const NF Ts_in = sf.out.Ts;
                surface_processes(p, sf.in, Ts_in, c.T, c.sat, c.liq, Kf_top, S, RICHARDS, v.g.dzc_top, sf.out);
                if (ln.is_top) {
                    bc.flux_U = -flux_term_top(sf.out.ghf, v.g);
                    if (RICHARDS) bc.flux_S = -flux_term_top(-sf.out.infil, v.g);
                }
                if (step == nsteps - 1 && ln.act && ln.is_top) {
                    // the diagnostics of the surface processes as the last step saw them (what k_surface leaves behind);
                    // stored here so that they do not occupy registers across the rest of the step
                    const SebOut<NF>& o = sf.out;
                    const unsigned ib = block_local(ib0);
                    stg(v.ghf, ib, o.ghf); stg(v.swu, ib, o.swu); stg(v.lwu, ib, o.lwu); stg(v.rnet, ib, o.rnet);
                    stg(v.Hs, ib, o.Hs); stg(v.Hl, ib, o.Hl); stg(v.evap, ib, o.evap); stg(v.infil, ib, o.infil); stg(v.runoff, ib, o.runoff);
                }
                sf.out.Ts = sf.out.Ts + NF(0) * dt;   // zero-tendency prognostic skin_temperature
            }
            TRM_PHASE_FENCE("tendencies", c.U, c.sat, c.psi, c.T, c.liq, bc.bTb, bc.bTt, bc.flux_U, bc.flux_S, S_in, Ts_in);
            // (the cell's fractions: from the derivation at entry, from the previous step's closure inside the multi-step loop)
            const Frac<NF>* pre = (PROG == PROG_MULTI && step > 0) ? &f_new : (DERIVE_TL ? &f_in : nullptr);
            t = column_tendencies<NF, RICHARDS, HYD, LPC>(v, p, L, ln, c, bc.bTb, bc.bTt, need_kc, viol, pre);
            NF gU = t.gU, gS = t.gS;
            TRM_PHASE_FENCE("advance", gU, gS, t.Kf_lo, t.Kc);
            front_wait();
            over = column_advance<NF, RICHARDS, LPC>(v, L, ln, Nz, bc, c.U, c.sat, gU, gS, dt, n, z0, bad);
            if (PROG == PROG_MULTI && RICHARDS) {   // surface_excess_water carried in the top lane's register
                GS_out = NF(0) + jl_min(NF(0), S);
                S = (S + GS_out * dt) + over;
            }
            // (second half of the step: parameters fetched afresh instead of being kept in SGPRs across the first half --
            // the kernel is short of scalar registers, and what does not fit is parked in VGPR lanes at a VALU move each)
            TRM_PHASE_FENCE("closure", n.U, n.sat, z0, over, gU, gS);
            f_new = column_closure<NF, RICHARDS, HYD>(kernarg_reload<DevParams<NF>>(off_p), L, z0, n, viol);
            gU_out = gU; gS_out = gS;
        }
    }
    TRM_PHASE_FENCE("outputs", n.U, n.sat, n.T, n.liq, n.psi);

    // ---- hydraulic_conductivity of the state: K(state the last tendencies saw), K(new state) when finalizing -----------
    NF Kf_out = t.Kf_lo, Kf_out_top = t.Kc;
    if (finalize && write_kf) {
        const DevParams<NF>& p = kernarg_reload<DevParams<NF>>(off_p);
        const NF Kc_new = conductivity_hydraulic<NF, HYD, false>(p, n.liq, f_new);     // (the closure has checked this composition)
        const NF Kc_new_m = shfl_up1<NF, LPC>(Kc_new);
        const NF Kmin_new = jl_min(Kc_new, Kc_new_m);
        Kf_out = (ln.is_bot || ln.is_top) ? Kc_new : Kmin_new;
        Kf_out_top = Kc_new;
    }
    // surface_excess_water and the skin temperature after the step, formed BEFORE the first store is issued: every loaded value
    // has been consumed by then, and nothing is waited for behind the stores (see S_in above)
    NF Ts_new = NF(0);
    if (PROG != PROG_MULTI) {
        if (RICHARDS) {
            // tendency min(0, S) once per column (SURVEY C-3), Euler / Heun update, overflow
            S = S_in;
            GS_out = NF(0) + jl_min(NF(0), S);
            if (PROG == PROG_HEUN) {
                S_stage_out = (S + GS_out * dt) + over_stage;
                GS_out = (GS_out + (NF(0) + jl_min(NF(0), S_stage_out))) / NF(2);
            }
            S = (S + GS_out * dt) + over;
        }
        if (seb) Ts_new = Ts_in + NF(0) * dt;   // zero-tendency prognostic skin_temperature
        asm volatile("" : "+v"(S), "+v"(GS_out), "+v"(Ts_new), "+v"(S_stage_out));
    }
    // ---- the column goes out: 6 coalesced stores ---------------------------------------------------------------------------
    TRM_PHASE_FENCE("stores", Kf_out, Kf_out_top, S, GS_out, Ts_new);
    if (ln.act) {
        const View<NF>& v = kernarg_reload<View<NF>>(0);
        // every base pointer the store phase may need in ONE batch of scalar loads: fetched where they are used -- inside the
        // finalize / write_kf / top-lane branches -- each is a scalar load and a wait of its own in front of its store
        NF* const pGU = v.G_U; NF* const pGS3 = v.G_sat; NF* const pKf = v.Kf; NF* const pKft = v.Kf_top; NF* const pS = v.S; NF* const pwt = v.wt;
        asm volatile("" : : "s"(pGU), "s"(pGS3), "s"(pKf), "s"(pKft), "s"(pS), "s"(pwt));
        // (block_local in EVERY block that stores: an offset defined in another block has been widened to 64 bits there)
        const unsigned cb = block_local(cb0), ib = block_local(ib0);
        stg(v.U, cb, n.U);
        stg(v.T, cb, n.T);
        stg(v.liq, cb, n.liq);
        if (RICHARDS) { stg(v.sat, cb, n.sat); stg(v.psi, cb, n.psi); }
        if (finalize) {   // state.tendencies as the reference leaves them after its last step
            stg(pGU, block_local(cb0), gU_out);
            if (RICHARDS) stg(pGS3, block_local(cb0), gS_out);
        }
        if (write_kf) stg(pKf, block_local(cb0), Kf_out);
        if (ln.is_top && RICHARDS && PROG == PROG_HEUN && a.stage_S) stg(a.stage_S, ib, S_stage_out);
        if (ln.is_top && STAGED) {
            // The per-column outputs (up to eight 8-byte values: top face of K, surface excess water, water table, its tendency,
            // the top cell for the next surface energy balance, the skin temperature) go to the workgroup's staging table: one
            // wave writes them below, 64 contiguous bytes per array and workgroup in a single instruction, instead of eight
            // stores of two active lanes each -- eight partially written cache lines per wave (profiles/r03/exp17, 19, 20).
            const int cib = (int)(threadIdx.x >> 6) * CPW + sub;      // column within the workgroup
            constexpr int cpb = (TRM_STEP_BLOCK / 64) * CPW;
            NF* st = small_stage<NF>();
            if (write_kf) st[SMALL_KF_TOP * cpb + cib] = Kf_out_top;
            if (RICHARDS) {
                st[SMALL_S * cpb + cib] = S;
                st[SMALL_WT * cpb + cib] = z0;
                if (finalize) st[SMALL_G_S * cpb + cib] = GS_out;
            }
            if (seb) {   // the next surface energy balance reads these
                st[SMALL_TOP_T * cpb + cib] = n.T;
                st[SMALL_TOP_SAT * cpb + cib] = n.sat;
                st[SMALL_TOP_LIQ * cpb + cib] = n.liq;
                st[SMALL_TS * cpb + cib] = SEB_INLINE ? sf.out.Ts : Ts_new;
            }
        } else if (ln.is_top) {
            if (write_kf) stg(pKft, block_local(ib0), Kf_out_top);
            if (RICHARDS) {
                const unsigned ib = block_local(ib0);
                stg(pS, ib, S);
                stg(pwt, ib, z0);
                if (finalize) stg(v.G_S, block_local(ib0), GS_out);
            }
            if (seb) {   // the next surface energy balance reads these
                const unsigned ib = block_local(ib0);
                stg(v.top_T, ib, n.T);
                stg(v.top_sat, ib, n.sat);
                stg(v.top_liq, ib, n.liq);
                stg(v.Ts, ib, SEB_INLINE ? sf.out.Ts : Ts_new);
            }
        }
        viol |= bad ? 1u : 0u;
    }
    if (STAGED) {
        const unsigned enabled = (write_kf ? 1u << SMALL_KF_TOP : 0u) | (RICHARDS ? (1u << SMALL_S) | (1u << SMALL_WT) : 0u) |
                                 ((RICHARDS && finalize) ? 1u << SMALL_G_S : 0u) |
                                 (seb ? (1u << SMALL_TOP_T) | (1u << SMALL_TOP_SAT) | (1u << SMALL_TOP_LIQ) | (1u << SMALL_TS) : 0u);
        store_small_outputs<NF, (TRM_STEP_BLOCK / 64) * CPW>(enabled, block, Nh);
    }
    // (only real cells report: the clamped copies that tail lanes carry are not repaired and may be out of bounds)
    if (viol && ln.act) atomicOr(v_arg.status, viol);
}

template <class NF, bool RICHARDS, int HYD, int LPC, int DERIVE, int PROG, bool SEB_INLINE, bool SERIES = false, bool STAGED = false, bool SCALAR_IN = true, int BCSIG = BCSIG_RUNTIME>
__global__ void __launch_bounds__(TRM_STEP_BLOCK)
    __attribute__((amdgpu_waves_per_eu(PROG == PROG_EULER ? (HYD == HYD_BC_LINEAR ? TRM_COLUMN_WAVES_EULER : 5) : (PROG == PROG_HEUN ? 5 : (HYD == HYD_BC_LINEAR ? 4 : 3)), 8)))
#ifdef TRM_COLUMN_NUM_SGPR
    __attribute__((amdgpu_num_sgpr(TRM_COLUMN_NUM_SGPR)))
#endif
    k_column(View<NF> v_arg, DevParams<NF> p_arg, ColumnArgs<NF> a) {
    column_program<NF, RICHARDS, HYD, LPC, DERIVE, PROG, SEB_INLINE, SERIES, STAGED, SCALAR_IN, BCSIG>(v_arg, p_arg, a, xcd_block<TRM_XCD_REMAP != 0>(blockIdx.x, gridDim.x));
}

// ---- LandModel, ONE launch per step (TRM_OPT_SURFACE_IN_LAUNCH): the first workgroups evaluate the 0-D surface processes of 256
// columns each (surface_front), the others run the ForwardEuler column program, which receives ground heat flux, infiltration and
// the new skin temperature from them through granules.  Same operations per column as the k_surface / k_column pair: bit-identical.
#ifndef TRM_LAND_WAVES
#define TRM_LAND_WAVES 7
#endif
template <class NF, bool RICHARDS, int HYD, int LPC, int DERIVE, bool STAGED, bool SCALAR_IN, int PROG = PROG_EULER>
__global__ void __launch_bounds__(TRM_STEP_BLOCK)
    __attribute__((amdgpu_waves_per_eu(PROG == PROG_HEUN ? 5 : TRM_LAND_WAVES, 8)))
    k_column_land(View<NF> v_arg, DevParams<NF> p_arg, ColumnArgs<NF> a, FrontArgs fa) {
    // (FrontArgs::chain_blocks, formed here from the column count: the head of the View is what a column wave needs first anyway, so the
    //  branch costs it no scalar round trip of its own)
    const int chain_blocks = (int)((v_arg.Nh + (TRM_STEP_BLOCK - 1)) / TRM_STEP_BLOCK);
    if ((int)blockIdx.x < chain_blocks) {
        // (the surface waves share their SIMDs with up to seven column waves, all of which will wait for them: they issue first)
        __builtin_amdgcn_s_setprio(TRM_FRONT_PRIO);
        const long i = (long)blockIdx.x * TRM_STEP_BLOCK + threadIdx.x;
        // (the surface workgroups fetch their arguments HERE, through a pointer the optimiser cannot see through: read as `v_arg.x`
        //  their scalar loads were hoisted in front of the branch, where the COLUMN waves paid for them -- ten v_writelane per wave
        //  to park them in a vector register, from the disassembly)
        constexpr unsigned off_p = round_up_to((unsigned)sizeof(View<NF>), (unsigned)alignof(DevParams<NF>));
        constexpr unsigned off_args = round_up_to(off_p + (unsigned)sizeof(DevParams<NF>), (unsigned)alignof(ColumnArgs<NF>));
        constexpr unsigned off_front = round_up_to(off_args + (unsigned)sizeof(ColumnArgs<NF>), (unsigned)alignof(FrontArgs));
        const View<NF>& v = kernarg_reload<View<NF>>(0);
        if (i - (long)(threadIdx.x & 63u) < v.Nh && !(TRM_FRONT_DIAG & 1))      // (wave-uniform)
            surface_front<NF, RICHARDS, HYD>(v, kernarg_reload<DevParams<NF>>(off_p), kernarg_reload<FrontArgs>(off_front), i);
        return;
    }
    column_program<NF, RICHARDS, HYD, LPC, DERIVE, PROG, false, false, STAGED, SCALAR_IN, BCSIG_LAND, true>(v_arg, p_arg, a, blockIdx.x - (unsigned)chain_blocks);
}

// ---- LandModel, one launch per half step: the soil columns of ONE half of the context and the 0-D surface processes of the
// OTHER half (land_model.jl:79-88: evaporation, runoff, the surface energy balance twice -- k_surface).  The surface launch
// is a latency-bound chain of ~450 dependent fp64 instructions on one wave per SIMD (5.4 us at N145, 16 % of the step); its
// result is needed by the columns' explicit step, and the columns' new top cells by the next surface evaluation -- a strict
// chain per column, but columns are independent: with the columns dealt to two halves A and B the launches
//     surf(A, 0) | col(A, 0) + surf(B, 0) | col(B, 0) + surf(A, 1) | col(A, 1) + surf(B, 1) | ...
// keep that chain for every column while each surface evaluation runs UNDER the column program of the other half.  The
// surface workgroups come first in the grid so that they are dispatched at once.  Same operations per column as the
// k_surface / k_column pair: bit-identical results.
template <class NF, bool RICHARDS, int HYD, int LPC, int DERIVE, bool TOP_ARRAYS>
__global__ void __launch_bounds__(TRM_STEP_BLOCK)
    __attribute__((amdgpu_waves_per_eu(HYD == HYD_BC_LINEAR ? TRM_COLUMN_WAVES_EULER : 5, 8)))
    k_land_euler(View<NF> v_col, DevParams<NF> p_arg, ColumnArgs<NF> a, View<NF> v_surf, int surface_blocks) {
    if ((int)blockIdx.x < surface_blocks) {
        const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
        if (i < v_surf.Nh) surface_program<NF, RICHARDS, HYD, true, TOP_ARRAYS>(v_surf, p_arg, i);
        return;
    }
    column_program<NF, RICHARDS, HYD, LPC, DERIVE, PROG_EULER, false, false>(v_col, p_arg, a, blockIdx.x - (unsigned)surface_blocks);
}

// ---- Heun with every boundary kind -------------------------------------------------------------------------------------
// compute_auxiliary! + compute_tendencies! of the column in registers with the generic boundary handling of k_step_wave
// (Value / Gradient on temperature, liquid fraction, saturation, pressure head; per-cell vwc_forcing), WITHOUT the
// compute_z_bcs! terms.  `vb` supplies the boundary kinds and values: the state's view, or the stage's (values at t + dt).
template <class NF, bool RICHARDS, int HYD, int LPC>
TRM_DEV Tendency<NF> column_tendencies_generic(const View<NF>& vb, const DevParams<NF>& p, const LevelGeom<NF>& L, const LaneInfo& ln,
                                               const Cell<NF>& c, int ii, unsigned cb, bool need_kc, uint32_t& viol) {
    const bool is_bot = ln.is_bot, is_top = ln.is_top;
    uint32_t viol_old = 0;
    const Frac<NF> f = fractions(p, c.sat, c.liq, viol_old);
    const NF kap = conductivity(p, f);
    const NF Kc = need_kc ? conductivity_hydraulic<NF, HYD, false>(p, c.liq, f) : NF(0);
    const NF T_sh = shfl_up1<NF, LPC>(c.T), kap_sh = shfl_up1<NF, LPC>(kap);
    NF T_m = T_sh, kap_m = kap_sh, T_h = NF(0), kap_h = NF(0), psi_hb = NF(0), psi_ht = NF(0);
    const bool same_bot = vb.bc.kind[3][0] != 1 && vb.bc.kind[3][0] != 3 &&
                          (RICHARDS ? (vb.bc.kind[1][0] != 1 && vb.bc.kind[1][0] != 3) : p.halo_policy == 1);
    const bool same_top = vb.bc.kind[3][1] != 1 && vb.bc.kind[3][1] != 3 &&
                          (RICHARDS ? (vb.bc.kind[1][1] != 1 && vb.bc.kind[1][1] != 3) : p.halo_policy == 1);
    if (is_bot) {
        T_m = halo_bottom(vb.bc.kind[2][0], bcval(vb, 2, 0), ii, c.T, vb.g);
        kap_m = kap;
        if (!same_bot) {
            const NF lh = halo_bottom(vb.bc.kind[3][0], bcval(vb, 3, 0), ii, c.liq, vb.g);
            const NF sh = sat_halo<NF, RICHARDS>(vb, p, 0, ii, c.sat);
            kap_m = conductivity(p, fractions(p, sh, lh, viol));
        }
        if (RICHARDS) psi_hb = halo_bottom(vb.bc.kind[4][0], bcval(vb, 4, 0), ii, c.psi, vb.g);
    }
    if (is_top) {
        T_h = halo_top(vb.bc.kind[2][1], bcval(vb, 2, 1), ii, c.T, vb.g);
        kap_h = kap;
        if (!same_top) {
            const NF lh = halo_top(vb.bc.kind[3][1], bcval(vb, 3, 1), ii, c.liq, vb.g);
            const NF sh = sat_halo<NF, RICHARDS>(vb, p, 1, ii, c.sat);
            kap_h = conductivity(p, fractions(p, sh, lh, viol));
        }
        if (RICHARDS) psi_ht = halo_top(vb.bc.kind[4][1], bcval(vb, 4, 1), ii, c.psi, vb.g);
    }
    const NF qT_lo = -(NF(0.5) * (kap + kap_m)) * ((c.T - T_m) * L.rdzf_lo);
    const NF qT_sh = shfl_dn1<NF, LPC>(qT_lo);
    const NF qT_hi = is_top ? -(NF(0.5) * (kap_h + kap)) * ((T_h - c.T) * L.rdzf_hi) : qT_sh;
    Tendency<NF> t;
    t.gU = NF(0) + (-((qT_hi - qT_lo) * L.rdzc));
    t.gS = NF(0);
    t.Kf_lo = NF(0);
    t.Kc = Kc;
    if (need_kc) {
        const NF Kc_m = shfl_up1<NF, LPC>(Kc);
        const NF Kmin = jl_min(Kc, Kc_m);
        t.Kf_lo = (is_bot || is_top) ? Kc : Kmin;
    }
    if (RICHARDS) {
        const NF Kf_lo = t.Kf_lo;
        const NF Kf_up = shfl_up1<NF, LPC>(Kf_lo), Kf_dn = shfl_dn1<NF, LPC>(Kf_lo), psi_sh = shfl_up1<NF, LPC>(c.psi);
        const NF Kf_m = is_bot ? NF(0) : Kf_up;
        const NF Kf_p = is_top ? Kc : Kf_dn;
        const NF psi_m = is_bot ? psi_hb : psi_sh;
        const NF g_lo = (c.psi - psi_m) * L.rdzf_lo;
        const NF Ks_lo = upwind_conductivity(g_lo, Kf_m, Kf_lo, Kf_p);
        const NF qW_lo = -Ks_lo * g_lo;
        const NF qW_sh = shfl_dn1<NF, LPC>(qW_lo);
        const NF g_t = (psi_ht - c.psi) * L.rdzf_hi;
        const NF Ks_t = upwind_conductivity(g_t, Kf_lo, Kc, NF(0));
        const NF qW_t = -Ks_t * g_t;
        const NF qW_hi = is_top ? qW_t : qW_sh;
        const NF F_user = vb.Fvwc ? ldg(vb.Fvwc, cb) : p.vwc_forcing;
        const NF dtheta = -((qW_hi - qW_lo) * L.rdzc) + NF(0) + F_user;
        t.gS = NF(0) + div_const(dtheta, p.por, p.rpor);
    }
    return t;
}

// One Heun step (heun.jl:37-71) with every boundary kind, both stages on the column in registers: the generic-boundary
// counterpart of k_column<PROG_HEUN>.  `vs_arg`: the stage's view -- its boundary values are the series evaluated at t + dt
// where a series feeds them, else the state's.
template <class NF, bool RICHARDS, int HYD, int LPC>
__global__ void __launch_bounds__(TRM_STEP_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 8)))
    k_heun_generic(View<NF> v_arg, DevParams<NF> p_arg, View<NF> vs_arg, ColumnArgs<NF> a) {
    constexpr unsigned off_p = round_up_to((unsigned)sizeof(View<NF>), (unsigned)alignof(DevParams<NF>));
    constexpr unsigned off_vs = round_up_to(off_p + (unsigned)sizeof(DevParams<NF>), (unsigned)alignof(View<NF>));
    const View<NF>& v = v_arg;
    const DevParams<NF>& p = p_arg;
    constexpr int CPW = 64 / LPC;
    LaneInfo ln;
    ln.lane = threadIdx.x & 63;
    const int wave = (int)((blockIdx.x * (unsigned)blockDim.x + threadIdx.x) >> 6);
    ln.k = ln.lane % LPC;
    const int sub = ln.lane / LPC;
    const int Nz = v.Nz, Nh = (int)v.Nh;
    ln.is_bot = ln.k == 0;
    ln.is_top = ln.k == Nz - 1;
    const NF dt = a.dt;
    const int finalize = a.finalize, write_kf = a.write_kf;
    const bool need_kc = RICHARDS || write_kf;
    const int i = wave * CPW + sub;
    const bool colok = i < Nh;
    ln.act = colok && ln.k < Nz;
    ln.m_act = wave_ballot(colok) & wave_ballot(ln.k < Nz);
    ln.m_bot = ln.m_top = 0ull;      // (repair_saturation takes the ballots itself)
    const int ii = colok ? i : Nh - 1;
    const unsigned ib0 = (unsigned)ii * (unsigned)sizeof(NF);
    const unsigned cb0 = ((unsigned)ii * (unsigned)v.Nzp + (unsigned)(ln.k < Nz ? ln.k : Nz - 1)) * (unsigned)sizeof(NF);
    uint32_t viol = 0;
    bool bad = false;
    const bool seb = p.seb != 0;

    Cell<NF> c;
    c.U = ldg(v.U, cb0);
    c.sat = ldg(v.sat, cb0);
    c.psi = RICHARDS ? ldg(v.psi, cb0) : NF(0);
    c.T = ldg(v.T, cb0);
    c.liq = ldg(v.liq, cb0);
    const LevelGeom<NF> L = level_geom(v, ln.k);      // (behind the field loads: see level_geom)
    // compute_z_bcs! terms of the STATE (both explicit steps use them: the stage's clock is still t for its predictor)
    ColumnBC<NF> bc{};
    {
        NF eU_b = NF(0), eU_t = NF(0), eS_b = NF(0), eS_t = NF(0);
        if (v.bc.kind[0][0] == 2) eU_b = flux_term_bottom(ldg(bcval(v, 0, 0), ib0), v.g);
        if (RICHARDS && v.bc.kind[1][0] == 2) eS_b = flux_term_bottom(ldg(bcval(v, 1, 0), ib0), v.g);
        if (seb || v.bc.kind[0][1] == 2) eU_t = -flux_term_top(ldg(seb ? v.ghf : bcval(v, 0, 1), ib0), v.g);
        if (RICHARDS && (seb || v.bc.kind[1][1] == 2)) {
            const NF fS = ldg(seb ? v.infil : bcval(v, 1, 1), ib0);
            eS_t = -flux_term_top(seb ? -fS : fS, v.g);
        }
        bc.flux_U = ln.is_bot ? eU_b : (ln.is_top ? eU_t : NF(0));
        bc.flux_S = ln.is_bot ? eS_b : (ln.is_top ? eS_t : NF(0));
    }
    // (read with the other inputs, not behind the field stores: see column_program)
    const NF S_in = RICHARDS ? ldg(v.S, ib0) : NF(0), Ts_in = seb ? ldg(v.Ts, ib0) : NF(0);
    // stage 1
    const Tendency<NF> t = column_tendencies_generic<NF, RICHARDS, HYD, LPC>(v, p, L, ln, c, ii, cb0, need_kc, viol);
    NF gU = t.gU, gS = t.gS, z0s, z0;
    Cell<NF> s, n;
    const NF over_stage = column_advance<NF, RICHARDS, LPC>(v, L, ln, Nz, bc, c.U, c.sat, gU, gS, dt, s, z0s, bad);
    column_closure<NF, RICHARDS, HYD>(kernarg_reload<DevParams<NF>>(off_p), L, z0s, s, viol);
    // stage 2: tendencies at the stage with the stage's boundary values
    uint32_t viol_stage = 0;
    const Tendency<NF> t2 = column_tendencies_generic<NF, RICHARDS, HYD, LPC>(kernarg_reload<View<NF>>(off_vs), kernarg_reload<DevParams<NF>>(off_p), L, ln, s, ii, cb0, RICHARDS, viol_stage);
    viol |= viol_stage;
    gU = (t.gU + t2.gU) / NF(2);
    gS = RICHARDS ? (t.gS + t2.gS) / NF(2) : NF(0);
    const NF over = column_advance<NF, RICHARDS, LPC>(kernarg_reload<View<NF>>(0), L, ln, Nz, bc, c.U, c.sat, gU, gS, dt, n, z0, bad);
    column_closure<NF, RICHARDS, HYD>(kernarg_reload<DevParams<NF>>(off_p), L, z0, n, viol);

    NF Kf_out = t.Kf_lo, Kf_out_top = t.Kc;
    if (finalize && write_kf) {
        const DevParams<NF>& pf = kernarg_reload<DevParams<NF>>(off_p);
        const NF Kc_new = conductivity_hydraulic<NF, HYD, false>(pf, n.liq, fractions(pf, n.sat, n.liq, viol));
        const NF Kc_new_m = shfl_up1<NF, LPC>(Kc_new);
        const NF Kmin_new = jl_min(Kc_new, Kc_new_m);
        Kf_out = (ln.is_bot || ln.is_top) ? Kc_new : Kmin_new;
        Kf_out_top = Kc_new;
    }
    if (ln.act) {
        const View<NF>& vo = kernarg_reload<View<NF>>(0);
        const unsigned cb = block_local(cb0), ib = block_local(ib0);
        stg(vo.U, cb, n.U);
        stg(vo.T, cb, n.T);
        stg(vo.liq, cb, n.liq);
        if (RICHARDS) { stg(vo.sat, cb, n.sat); stg(vo.psi, cb, n.psi); }
        if (finalize) {
            stg(vo.G_U, cb, gU);
            if (RICHARDS) stg(vo.G_sat, cb, gS);
        }
        if (write_kf) {
            stg(vo.Kf, cb, Kf_out);
            if (ln.is_top) stg(vo.Kf_top, ib, Kf_out_top);
        }
        if (ln.is_top) {
            if (RICHARDS) {
                NF S = S_in;
                NF GS = NF(0) + jl_min(NF(0), S);
                const NF S_stage = (S + GS * dt) + over_stage;
                GS = (GS + (NF(0) + jl_min(NF(0), S_stage))) / NF(2);
                S = (S + GS * dt) + over;
                stg(vo.S, ib, S);
                stg(vo.wt, ib, z0);
                if (finalize) stg(vo.G_S, ib, GS);
            }
            if (seb) {
                stg(vo.top_T, ib, n.T);
                stg(vo.top_sat, ib, n.sat);
                stg(vo.top_liq, ib, n.liq);
                stg(vo.Ts, ib, Ts_in + NF(0) * dt);
            }
        }
        viol |= bad ? 1u : 0u;
    }
    if (viol && ln.act) atomicOr(v_arg.status, viol);
}

}  // namespace trm
