// trm_column_deep.hpp -- the fused ForwardEuler step for columns of 65 ... 128 levels: TWO soil levels per lane.
//
// The reference's own saturation-adjustment test steps a UniformSpacing(N = 100) column (test/soil/soil_hydrology_tests.jl:
// 93-123); with one level per lane such grids fell to the reference-order kernels (four times the per-cell time).  Here lane l
// of a wavefront holds levels 2l ("a") and 2l + 1 ("b") of ONE column: a column is one contiguous 16-byte-per-lane access
// per field, the k - 1 neighbour of a is the previous lane's b (one DPP shift), of b the lane's own a (free); likewise
// upwards.  Ballots come in pairs (a cells, b cells).  Every operation is the one k_column<PROG_EULER> / the reference-order
// kernels perform on the cell, in the same order: results are bit-identical (tests/test_gpu_deep_columns.py).
// Scope: ForwardEuler and Heun (both stages in registers, one launch) with every boundary kind (GENERIC), the multi-step program
// with the branch-free kinds, the derivation of T / liq, and the soil half of the vegetation-coupled LandModel (Heun stores what
// the 0-D processes need of the stage); Nz > 128 keeps the reference-order kernels.
#pragma once
#include "trm_column.hpp"

namespace trm {

template <class NF> struct Two { NF a, b; };
// level k - 1 / k + 1 of both cells of a lane
template <class NF> TRM_DEV Two<NF> below(const Two<NF>& x) { return Two<NF>{shift_up(x.b), x.a}; }
template <class NF> TRM_DEV Two<NF> above(const Two<NF>& x) { return Two<NF>{x.b, shift_dn(x.a)}; }
template <class NF> TRM_DEV Two<NF> ld2cells(const NF* base, unsigned byte_off) {
    // (one 16-byte access for fp64 pairs, 8 bytes for fp32: the two levels are adjacent in the z-fastest layout)
    struct alignas(2 * sizeof(NF)) Raw { NF a, b; };
    const Raw r = *reinterpret_cast<const Raw*>(reinterpret_cast<const char*>(base) + byte_off);
    return Two<NF>{r.a, r.b};
}
struct DeepLane { int lane, ka, kb; bool act_a, act_b, bot_a, top_a, top_b; };

// broadcast of a per-level value from level q (lane q / 2, cell q % 2) to the wave
template <class NF> TRM_DEV NF from_level(const Two<NF>& x, int q) {
    const NF va = __shfl(x.a, q >> 1, 64), vb = __shfl(x.b, q >> 1, 64);
    return (q & 1) ? vb : va;
}
// bit q of the result = the predicate of level q (a cells are the even levels, b cells the odd ones), as two 64-bit halves
struct Mask128 { unsigned long long lo, hi; };
TRM_DEV Mask128 level_mask(bool pa, bool pb) {
    const unsigned long long A = wave_ballot(pa), B = wave_ballot(pb);
    // interleave: level 2l <- A bit l, level 2l + 1 <- B bit l
    auto spread = [](unsigned long long x) {   // bits 0..31 of x to the even positions of a 64-bit word
        x &= 0xffffffffull;
        x = (x | (x << 16)) & 0x0000ffff0000ffffull;
        x = (x | (x << 8)) & 0x00ff00ff00ff00ffull;
        x = (x | (x << 4)) & 0x0f0f0f0f0f0f0f0full;
        x = (x | (x << 2)) & 0x3333333333333333ull;
        x = (x | (x << 1)) & 0x5555555555555555ull;
        return x;
    };
    Mask128 m;
    m.lo = spread(A) | (spread(B) << 1);
    m.hi = spread(A >> 32) | (spread(B >> 32) << 1);
    return m;
}
TRM_DEV bool any(const Mask128& m) { return (m.lo | m.hi) != 0ull; }
TRM_DEV int lowest(const Mask128& m) { return m.lo ? __builtin_ctzll(m.lo) : 64 + __builtin_ctzll(m.hi); }
TRM_DEV int highest(const Mask128& m) { return m.hi ? 127 - __builtin_clzll(m.hi) : 63 - __builtin_clzll(m.lo); }
TRM_DEV bool any_above(const Mask128& m, int q) {   // a bit at a level > q
    if (q >= 127) return false;
    if (q >= 63) return (m.hi >> (q - 63)) != 0ull;
    return (m.lo >> (q + 1)) != 0ull || m.hi != 0ull;
}
TRM_DEV bool any_below(const Mask128& m, int q) {   // a bit at a level < q
    if (q <= 0) return false;
    if (q <= 64) return (q == 64 ? m.lo : (m.lo & ((1ull << q) - 1ull))) != 0ull;
    return m.lo != 0ull || (m.hi & ((1ull << (q - 64)) - 1ull)) != 0ull;
}

// adjust_saturation_profile! (soil_hydrology.jl:185-219), two levels per lane: repair_saturation() of trm_kernels.hpp with the
// level sets as 128-bit masks and the cell of level q addressed as (lane q / 2, cell q % 2)
template <class NF>
TRM_DEV NF repair_saturation_deep(Two<NF>& s, const DeepLane& ln, int Nz, const Two<NF>& dzc, const Two<NF>& rdzc, NF dzc_top) {
    const bool over_a = ln.act_a && !ln.top_a && s.a > NF(1), over_b = ln.act_b && !ln.top_b && s.b > NF(1);
    const bool under_a = ln.act_a && !ln.bot_a && s.a < NF(0), under_b = ln.act_b && s.b < NF(0);
    const Mask128 any_over = level_mask(over_a, over_b);
    const Mask128 any_bad = level_mask(over_a || under_a, over_b || under_b);
    s.a = ln.bot_a ? s.a : s.a + NF(0);
    s.b = s.b + NF(0);
    if (any(any_bad)) {
        // thickness of the cells above / below (the edge cells keep their own, as neighbour_dz() does)
        const Two<NF> dz_up = above(dzc), rdz_up = above(rdzc), dz_dn = below(dzc), rdz_dn = below(rdzc);
        const bool edge_up_a = ln.ka >= Nz - 1, edge_up_b = ln.kb >= Nz - 1;
        const Two<NF> nb_dz_up{edge_up_a ? dzc.a : dz_up.a, edge_up_b ? dzc.b : dz_up.b}, nb_rdz_up{edge_up_a ? rdzc.a : rdz_up.a, edge_up_b ? rdzc.b : rdz_up.b};
        const Two<NF> nb_dz_dn{ln.bot_a ? dzc.a : dz_dn.a, dz_dn.b}, nb_rdz_dn{ln.bot_a ? rdzc.a : rdz_dn.a, rdz_dn.b};
        if (any(any_over)) {
            NF carry = NF(0);
            for (int q = lowest(any_over); q < Nz - 1; ++q) {
                Two<NF> cout{NF(0), NF(0)};
                if (ln.ka == q) {
                    s.a = s.a + carry;
                    const NF e = jl_max(s.a - NF(1), NF(0));
                    s.a = s.a - e;
                    cout.a = div_const(e * dzc.a, nb_dz_up.a, nb_rdz_up.a);
                }
                if (ln.kb == q) {
                    s.b = s.b + carry;
                    const NF e = jl_max(s.b - NF(1), NF(0));
                    s.b = s.b - e;
                    cout.b = div_const(e * dzc.b, nb_dz_up.b, nb_rdz_up.b);
                }
                carry = from_level(cout, q);
                if (!any_above(any_over, q) && wave_ballot(!(carry == NF(0))) == 0ull) break;
            }
            if (ln.top_a) s.a = s.a + carry;
            if (ln.top_b) s.b = s.b + carry;
        }
        const bool under2_a = ln.act_a && !ln.bot_a && !(jl_max(-s.a, NF(0)) == NF(0));
        const bool under2_b = ln.act_b && !(jl_max(-s.b, NF(0)) == NF(0));
        const Mask128 any_under = level_mask(under2_a, under2_b);
        if (any(any_under)) {
            NF pend = NF(0);
            for (int q = highest(any_under); q >= 1; --q) {
                Two<NF> pout{NF(0), NF(0)};
                if (ln.ka == q) {
                    s.a = s.a - pend;
                    const NF d = jl_max(-s.a, NF(0));
                    s.a = s.a + d;
                    pout.a = div_const(d * dzc.a, nb_dz_dn.a, nb_rdz_dn.a);
                }
                if (ln.kb == q) {
                    s.b = s.b - pend;
                    const NF d = jl_max(-s.b, NF(0));
                    s.b = s.b + d;
                    pout.b = div_const(d * dzc.b, nb_dz_dn.b, nb_rdz_dn.b);
                }
                pend = from_level(pout, q);
                if (!any_below(any_under, q) && wave_ballot(!(pend == NF(0))) == 0ull) break;
            }
            if (ln.bot_a) s.a = s.a - pend;
        }
    }
    // surface overflow joins surface_excess_water (the top cell); bottom clamp (the bottom cell)
    const NF e_a = ln.top_a ? jl_max(s.a - NF(1), NF(0)) : NF(0), e_b = ln.top_b ? jl_max(s.b - NF(1), NF(0)) : NF(0);
    s.a = s.a - e_a;
    s.b = s.b - e_b;
    s.a = ln.bot_a ? jl_max(s.a, NF(0)) : s.a;
    return (e_a + e_b) * dzc_top;      // (one of the two is the top cell's excess, the other +0)
}

#ifndef TRM_DEEP_WAVES
#define TRM_DEEP_WAVES 1
#endif
// word w of the level record at byte offset `rec` of the level table (see level_geom)
template <class NF> TRM_DEV NF level_word(const View<NF>& v, unsigned rec, int w) {
    return *reinterpret_cast<const NF*>(reinterpret_cast<const char*>(v.lvl) + rec + (unsigned)w * (unsigned)sizeof(NF));
}
// PROG: PROG_EULER, or PROG_HEUN -- both stages of the reference's Heun (heun.jl:37-71) on the column in registers, one launch
// per step, the sequence of column_program<PROG_HEUN> (trm_column.hpp) cell by cell.
// GENERIC: every boundary kind (Value / Gradient on temperature, liquid fraction, saturation, pressure head) and the per-cell
// vwc_forcing field, with the halo formulas of column_tendencies_generic (trm_column.hpp) on the edge cells.
// `vs_arg`: the Heun stage's view -- its boundary kinds and values (series evaluated at t + dt, arrays the caller of the two-call
// Heun wrote) are what the stage's tendencies see with GENERIC (k_heun_generic, trm_column.hpp); unused otherwise.
template <class NF, bool RICHARDS, int HYD, bool DERIVE = false, int PROG = PROG_EULER, bool GENERIC = false>
__global__ void __launch_bounds__(TRM_STEP_BLOCK) __attribute__((amdgpu_waves_per_eu(TRM_DEEP_WAVES, 8)))
    k_column_deep(View<NF> v_arg, DevParams<NF> p_arg, ColumnArgs<NF> a, View<NF> vs_arg) {
    static_assert(!GENERIC || PROG != PROG_MULTI, "generic boundary kinds on deep columns: one step per launch");
    // PROG_MULTI: a.nsteps ForwardEuler steps on the resident column (contexts without the surface energy balance and without
    // time series: constants between the steps), fields written once per launch -- column_program<PROG_MULTI>'s loop.
    constexpr unsigned off_p = round_up_to((unsigned)sizeof(View<NF>), (unsigned)alignof(DevParams<NF>));
    constexpr unsigned off_a = round_up_to(off_p + (unsigned)sizeof(DevParams<NF>), (unsigned)alignof(ColumnArgs<NF>));
    constexpr unsigned off_vs = round_up_to(off_a + (unsigned)sizeof(ColumnArgs<NF>), (unsigned)alignof(View<NF>));
    (void)off_vs;
    const View<NF>& v = v_arg;
    const DevParams<NF>& p = p_arg;
    DeepLane ln;
    ln.lane = threadIdx.x & 63;
    const int i = (int)((blockIdx.x * (unsigned)blockDim.x + threadIdx.x) >> 6);      // one column per wave
    const int Nz = v.Nz, Nh = (int)v.Nh;
    ln.ka = 2 * ln.lane;
    ln.kb = ln.ka + 1;
    const bool colok = i < Nh;
    ln.act_a = colok && ln.ka < Nz;
    ln.act_b = colok && ln.kb < Nz;
    ln.bot_a = ln.ka == 0;
    ln.top_a = ln.ka == Nz - 1;
    ln.top_b = ln.kb == Nz - 1;
    const bool is_top_lane = ln.top_a || ln.top_b;
    const int ii = colok ? i : Nh - 1;
    // Lanes beyond the top load the column's last pair again (an aligned 2-word access inside the column's pitch) and store
    // nothing.  With an odd number of levels the top lane's b cell is the first padding word: read, never stored, and nothing
    // of it reaches a real cell (the top cell takes its upper face, conductivities and fluxes from the boundary formulas).
    const int last_pair = (Nz - 1) & ~1;
    const int ka_ld = ln.ka <= last_pair ? ln.ka : last_pair;
    const unsigned ib0 = (unsigned)ii * (unsigned)sizeof(NF);
    const unsigned cb0 = ((unsigned)ii * (unsigned)v.Nzp + (unsigned)ka_ld) * (unsigned)sizeof(NF);
#if TRM_DEEP_SCALAR_INPUTS
    // One column per wave: the per-column inputs (boundary values, ground heat flux, infiltration, the 0-D fields) are wave-uniform and
    // come through the SCALAR memory path -- no vector registers, no vector load in the middle of the wave's life, no wait for vector
    // memory where conditional loads meet (k_column: col_req; profiles/r04/exp18)
    const unsigned ib_u = (unsigned)__builtin_amdgcn_readfirstlane((int)ib0);
    auto col_ld = [&](const NF* ptr, unsigned) -> NF { return sld_off<NF>(ptr, ib_u); };
#else
    auto col_ld = [&](const NF* ptr, unsigned off) -> NF { return ldg(ptr, off); };
#endif
    const NF dt = a.dt;
    const int finalize = a.finalize, write_kf = a.write_kf;
    const bool need_kc = RICHARDS || write_kf;
    uint32_t viol_a = 0, viol_b = 0;    // (per cell: only real cells report)
    bool bad = false;

    // per-level geometry of both cells (the level records of trm_kernels.hpp: level_geom).  Only what the tendencies and the
    // repair need is fetched here (thickness, its reciprocal, the face reciprocals); zC, psiz and zFlo serve the water table and
    // the pressure head at the END of a stage and are fetched there (level_word): 12 registers less across the stencil.
    struct EarlyGeom { NF dzc, rdzc, rdzf_lo, rdzf_hi; };
    auto early = [&](int k) {
        const int kk = k < Nz ? k : Nz - 1;
        const NF* q = reinterpret_cast<const NF*>(reinterpret_cast<const char*>(v.lvl) + (unsigned)kk * (unsigned)sizeof(LevelPack<NF>));
        return EarlyGeom{q[3], q[4], q[5], q[6]};
    };
    const EarlyGeom La = early(ln.ka), Lb = early(ln.kb);
    const Two<NF> dzc{La.dzc, Lb.dzc}, rdzc{La.rdzc, Lb.rdzc};

    // ---- the column comes in ---------------------------------------------------------------------------------------
    Two<NF> U = ld2cells(v.U, cb0), sat = ld2cells(v.sat, cb0);
    Two<NF> psi = RICHARDS ? ld2cells(v.psi, cb0) : Two<NF>{NF(0), NF(0)};
    Two<NF> T, liq;
    if (DERIVE) {   // (T, liq) of the incoming state re-derived from (U, sat) instead of being read (k_column: DERIVE_T_LIQ)
        uint32_t viol_in = 0;
        const DevParams<NF>& pd = kernarg_reload<DevParams<NF>>(off_p);
        energy_closure_wave(pd, U.a, sat.a, liq.a, T.a, viol_in);
        energy_closure_wave(pd, U.b, sat.b, liq.b, T.b, viol_in);
    } else {
        T = ld2cells(v.T, cb0);
        liq = ld2cells(v.liq, cb0);
    }
    const bool seb = p.seb != 0;
    const bool vTb = v.bc.kind[2][0] == 1, vTt = v.bc.kind[2][1] == 1;
    const NF bTb = vTb ? col_ld(bcval(v, 2, 0), ib0) : NF(0), bTt = vTt ? col_ld(bcval(v, 2, 1), ib0) : NF(0);

    // ---- compute_auxiliary! + compute_tendencies! (column_tendencies, trm_column.hpp, per cell) of a (T, liq, sat, psi) ----
    struct Tend { Two<NF> gU, gS, Kf_lo, Kc; };
    auto tendencies = [&](const View<NF>& v, const DevParams<NF>& p, const Two<NF>& T, const Two<NF>& liq, const Two<NF>& sat, const Two<NF>& psi,
                          NF bTb, NF bTt, bool need_kc, uint32_t& viol_a, uint32_t& viol_b) {
        uint32_t viol_old = 0;
        const Frac<NF> fa = fractions(p, sat.a, liq.a, viol_old), fb = fractions(p, sat.b, liq.b, viol_old);
        const Two<NF> kap{conductivity(p, fa), conductivity(p, fb)};
        const Two<NF> Kc{need_kc ? conductivity_hydraulic<NF, HYD, false>(p, liq.a, fa) : NF(0), need_kc ? conductivity_hydraulic<NF, HYD, false>(p, liq.b, fb) : NF(0)};
        const Two<NF> T_dn = below(T), kap_dn = below(kap);
        // temperature halos of the edge cells
        auto ext_b = [&](NF Tc) { return vTb ? Tc + div_const(Tc - bTb, v.g.hdzf_bot, v.g.rhdzf_bot) * (-v.g.dzf_bot) : Tc; };
        auto ext_t = [&](NF Tc) { return vTt ? Tc + div_const(bTt - Tc, v.g.hdzf_top, v.g.rhdzf_top) * v.g.dzf_top : Tc; };
        auto halo_kap = [&](NF kc, NF lq, uint32_t& vl) { return (!RICHARDS && p.halo_policy != 1) ? conductivity(p, fractions(p, NF(0), lq, vl)) : kc; };
        NF T_m_a, kap_m_a, T_h_a = NF(0), T_h_b = NF(0), kap_h_a = NF(0), kap_h_b = NF(0), psi_hb = NF(0), psi_ht_a = NF(0), psi_ht_b = NF(0);
        const NF T_m_b = T_dn.b, kap_m_b = kap_dn.b;
        if constexpr (GENERIC) {
            // halo cells of the edge cells from the boundary kinds (column_tendencies_generic): divergent branches of the two
            // edge lanes only
            const bool same_bot = v.bc.kind[3][0] != 1 && v.bc.kind[3][0] != 3 && (RICHARDS ? (v.bc.kind[1][0] != 1 && v.bc.kind[1][0] != 3) : p.halo_policy == 1);
            const bool same_top = v.bc.kind[3][1] != 1 && v.bc.kind[3][1] != 3 && (RICHARDS ? (v.bc.kind[1][1] != 1 && v.bc.kind[1][1] != 3) : p.halo_policy == 1);
            T_m_a = T_dn.a;
            kap_m_a = kap_dn.a;
            if (ln.bot_a) {
                T_m_a = halo_bottom(v.bc.kind[2][0], bcval(v, 2, 0), ii, T.a, v.g);
                kap_m_a = kap.a;
                if (!same_bot) {
                    const NF lh = halo_bottom(v.bc.kind[3][0], bcval(v, 3, 0), ii, liq.a, v.g);
                    const NF sh = sat_halo<NF, RICHARDS>(v, p, 0, ii, sat.a);
                    kap_m_a = conductivity(p, fractions(p, sh, lh, viol_a));
                }
                if (RICHARDS) psi_hb = halo_bottom(v.bc.kind[4][0], bcval(v, 4, 0), ii, psi.a, v.g);
            }
            auto top_halo = [&](bool is_top, NF Tc, NF lq, NF sc, NF psic, NF kapc, NF& T_h, NF& kap_h, NF& psi_ht, uint32_t& vl) {
                if (!is_top) return;
                T_h = halo_top(v.bc.kind[2][1], bcval(v, 2, 1), ii, Tc, v.g);
                kap_h = kapc;
                if (!same_top) {
                    const NF lh = halo_top(v.bc.kind[3][1], bcval(v, 3, 1), ii, lq, v.g);
                    const NF sh = sat_halo<NF, RICHARDS>(v, p, 1, ii, sc);
                    kap_h = conductivity(p, fractions(p, sh, lh, vl));
                }
                if (RICHARDS) psi_ht = halo_top(v.bc.kind[4][1], bcval(v, 4, 1), ii, psic, v.g);
            };
            top_halo(ln.top_a, T.a, liq.a, sat.a, psi.a, kap.a, T_h_a, kap_h_a, psi_ht_a, viol_a);
            top_halo(ln.top_b, T.b, liq.b, sat.b, psi.b, kap.b, T_h_b, kap_h_b, psi_ht_b, viol_b);
        } else {
            T_m_a = ln.bot_a ? ext_b(T.a) : T_dn.a;
            kap_h_a = halo_kap(kap.a, liq.a, viol_a);
            kap_h_b = halo_kap(kap.b, liq.b, viol_b);
            kap_m_a = ln.bot_a ? kap_h_a : kap_dn.a;
            T_h_a = ext_t(T.a);
            T_h_b = ext_t(T.b);
        }
        const Two<NF> qT_lo{-(NF(0.5) * (kap.a + kap_m_a)) * ((T.a - T_m_a) * La.rdzf_lo), -(NF(0.5) * (kap.b + kap_m_b)) * ((T.b - T_m_b) * Lb.rdzf_lo)};
        const Two<NF> qT_up = above(qT_lo);
        const NF qT_hi_a = ln.top_a ? -(NF(0.5) * (kap_h_a + kap.a)) * ((T_h_a - T.a) * La.rdzf_hi) : qT_up.a;
        const NF qT_hi_b = ln.top_b ? -(NF(0.5) * (kap_h_b + kap.b)) * ((T_h_b - T.b) * Lb.rdzf_hi) : qT_up.b;
        Tend t;
        t.gU = Two<NF>{NF(0) + (-((qT_hi_a - qT_lo.a) * La.rdzc)), NF(0) + (-((qT_hi_b - qT_lo.b) * Lb.rdzc))};
        t.gS = Two<NF>{NF(0), NF(0)};
        t.Kf_lo = Two<NF>{NF(0), NF(0)};
        t.Kc = Kc;
        if (need_kc) {   // face conductivities (soil_hydrology.jl:145-163)
            const Two<NF> Kc_dn = below(Kc);
            const NF Kmin_a = jl_min(Kc.a, Kc_dn.a), Kmin_b = jl_min(Kc.b, Kc_dn.b);
            t.Kf_lo.a = (ln.bot_a || ln.top_a) ? Kc.a : Kmin_a;
            t.Kf_lo.b = ln.top_b ? Kc.b : Kmin_b;
        }
        if (RICHARDS) {  // Darcy fluxes (soil_hydrology_rre.jl:95-131)
            const Two<NF> Kf_lo = t.Kf_lo;
            const Two<NF> Kf_dn = below(Kf_lo), Kf_up = above(Kf_lo), psi_dn = below(psi);
            const NF Kf_m_a = ln.bot_a ? NF(0) : Kf_dn.a, Kf_m_b = Kf_dn.b;
            const NF Kf_p_a = ln.top_a ? Kc.a : Kf_up.a, Kf_p_b = ln.top_b ? Kc.b : Kf_up.b;
            const NF psi_m_a = ln.bot_a ? (GENERIC ? psi_hb : psi.a) : psi_dn.a, psi_m_b = psi_dn.b;
            const NF g_lo_a = (psi.a - psi_m_a) * La.rdzf_lo, g_lo_b = (psi.b - psi_m_b) * Lb.rdzf_lo;
            const Two<NF> qW_lo{-upwind_conductivity(g_lo_a, Kf_m_a, Kf_lo.a, Kf_p_a) * g_lo_a, -upwind_conductivity(g_lo_b, Kf_m_b, Kf_lo.b, Kf_p_b) * g_lo_b};
            const Two<NF> qW_up = above(qW_lo);
            NF qW_t_a, qW_t_b, F_a = p.vwc_forcing, F_b = p.vwc_forcing;
            if constexpr (GENERIC) {
                // boundary face above the top cell with the halo cell's pressure head; the user forcing per cell when the field is set
                const NF g_t_a = (psi_ht_a - psi.a) * La.rdzf_hi, g_t_b = (psi_ht_b - psi.b) * Lb.rdzf_hi;
                qW_t_a = -upwind_conductivity(g_t_a, Kf_lo.a, Kc.a, NF(0)) * g_t_a;
                qW_t_b = -upwind_conductivity(g_t_b, Kf_lo.b, Kc.b, NF(0)) * g_t_b;
                if (v.Fvwc) {
                    const Two<NF> F = ld2cells(v.Fvwc, cb0);
                    F_a = F.a; F_b = F.b;
                }
            } else {
                qW_t_a = -jl_min(Kc.a, NF(0)) * (psi.a - psi.a);
                qW_t_b = -jl_min(Kc.b, NF(0)) * (psi.b - psi.b);
            }
            const NF qW_hi_a = ln.top_a ? qW_t_a : qW_up.a, qW_hi_b = ln.top_b ? qW_t_b : qW_up.b;
            const NF dth_a = -((qW_hi_a - qW_lo.a) * La.rdzc) + NF(0) + F_a, dth_b = -((qW_hi_b - qW_lo.b) * Lb.rdzc) + NF(0) + F_b;
            t.gS.a = NF(0) + div_const(dth_a, p.por, p.rpor);
            t.gS.b = NF(0) + div_const(dth_b, p.por, p.rpor);
        }
        return t;
    };
    // ---- compute_z_bcs! + explicit_step! + hydrology closure's repair and water table (column_advance) of the STATE's (U, sat)
    unsigned late_a = (unsigned)(ln.ka < Nz ? ln.ka : Nz - 1) * (unsigned)sizeof(LevelPack<NF>), late_b = (unsigned)(ln.kb < Nz ? ln.kb : Nz - 1) * (unsigned)sizeof(LevelPack<NF>);
    auto advance = [&](Two<NF>& gU, Two<NF>& gS, const Two<NF>& flux_U, const Two<NF>& flux_S, Two<NF>& Un, Two<NF>& sn, NF& z0) {
        gU.a += flux_U.a; gU.b += flux_U.b;
        Un = Two<NF>{U.a + gU.a * dt, U.b + gU.b * dt};
        bad = bad || (ln.act_a && is_nan(Un.a)) || (ln.act_b && is_nan(Un.b));
        sn = sat;
        z0 = NF(0);
        NF over = NF(0);
        if (RICHARDS) {
            gS.a += flux_S.a; gS.b += flux_S.b;
            sn.a = sat.a + gS.a * dt;
            sn.b = sat.b + gS.b * dt;
            bad = bad || (ln.act_a && is_nan(sn.a)) || (ln.act_b && is_nan(sn.b));
            over = repair_saturation_deep<NF>(sn, ln, Nz, dzc, rdzc, v.g.dzc_top);
            asm volatile("" : "+v"(late_a), "+v"(late_b));   // (the late geometry is addressed from here on: not hoisted above the stencil)
            // compute_water_table! (soil_hydrology.jl:170-175): lower face of the first unsaturated cell from the bottom
            const Mask128 unsat = level_mask(ln.act_a && sn.a < NF(1), ln.act_b && sn.b < NF(1));
            const Two<NF> zFlo{level_word(v, late_a, 2), level_word(v, late_b, 2)};
            const int first = any(unsat) ? lowest(unsat) : -1;
            const NF z_first = from_level(zFlo, first >= 0 ? first : 0);
            z0 = first >= 0 ? z_first : v.g.zF_top;
        }
        return over;
    };
    // ---- closures (column_closure): (U, sat) -> (T, liq, psi), parameters fetched afresh -----------------------------------
    auto closure = [&](const Two<NF>& Un, const Two<NF>& sn, NF z0, Two<NF>& ln_, Two<NF>& Tn, Two<NF>& psin) {
        const DevParams<NF>& p2 = kernarg_reload<DevParams<NF>>(off_p);
        // (one ballot decision for the wave's phase-change divide per cell set, as energy_closure_wave)
        energy_closure_wave(p2, Un.a, sn.a, ln_.a, Tn.a, viol_a);
        energy_closure_wave(p2, Un.b, sn.b, ln_.b, Tn.b, viol_b);
        psin = Two<NF>{NF(0), NF(0)};
        if (RICHARDS) {
            psin.a = pressure_head<NF, HYD>(p2, sn.a, level_word(v, late_a, 0), level_word(v, late_a, 1), z0);
            psin.b = pressure_head<NF, HYD>(p2, sn.b, level_word(v, late_b, 0), level_word(v, late_b, 1), z0);
        }
    };

    Tend t = tendencies(v, p, T, liq, sat, psi, bTb, bTt, need_kc, viol_a, viol_b);   // tendencies at the STATE (hydraulic_conductivity comes from here)
    // The boundary flux terms and the 0-D inputs are fetched HERE: after the first stencil (its registers are free again), in front
    // of every store (no load sits behind a store: see column_program).
    unsigned ib_late = ib0;
    asm volatile("" : "+v"(ib_late));
    NF eU_b = NF(0), eU_t = NF(0), eS_b = NF(0), eS_t = NF(0);
    if (v.bc.kind[0][0] == 2) eU_b = flux_term_bottom(col_ld(bcval(v, 0, 0), ib_late), v.g);
    if (RICHARDS && v.bc.kind[1][0] == 2) eS_b = flux_term_bottom(col_ld(bcval(v, 1, 0), ib_late), v.g);
    if (seb || v.bc.kind[0][1] == 2) eU_t = -flux_term_top(col_ld(seb ? v.ghf : bcval(v, 0, 1), ib_late), v.g);
    if (RICHARDS && (seb || v.bc.kind[1][1] == 2)) {
        const NF fS = col_ld(seb ? v.infil : bcval(v, 1, 1), ib_late);
        eS_t = -flux_term_top(seb ? -fS : fS, v.g);
    }
    const NF S_in = RICHARDS ? col_ld(v.S, ib_late) : NF(0), Ts_in = seb ? col_ld(v.Ts, ib_late) : NF(0);
    const Two<NF> flux_U{ln.bot_a ? eU_b : (ln.top_a ? eU_t : NF(0)), ln.top_b ? eU_t : NF(0)};
    const Two<NF> flux_S{ln.bot_a ? eS_b : (ln.top_a ? eS_t : NF(0)), ln.top_b ? eS_t : NF(0)};

    Two<NF> gU = t.gU, gS = t.gS, Un, sn, ln_, Tn, psin;
    NF z0 = NF(0), over = NF(0), over_stage = NF(0);
    if (PROG == PROG_HEUN) {
        // stage 1: Euler predictor with the state's boundary fluxes and its closures (the stage never leaves the registers)
        const Two<NF> G1U = t.gU, G1S = t.gS;
        Two<NF> Us, ss, ls, Ts, ps;
        NF z0s;
        over_stage = advance(gU, gS, flux_U, flux_S, Us, ss, z0s);
        closure(Us, ss, z0s, ls, Ts, ps);
        if (a.stage_T) {   // (wave-uniform: the vegetation-coupled LandModel evaluates its 0-D processes AT the stage: k_column<PROG_HEUN>)
            auto stage_cell = [&](bool act, unsigned cb_, NF s_, NF l_, NF t_) {
                if (!act) return;
                const unsigned cb = block_local(cb_);
                stg(a.stage_sat, cb, s_);
                stg(a.stage_liq, cb, l_);
                stg(a.stage_T, cb, t_);
            };
            const unsigned sa = ((unsigned)ii * (unsigned)v.Nzp + (unsigned)ln.ka) * (unsigned)sizeof(NF);
            stage_cell(ln.act_a, sa, ss.a, ls.a, Ts.a);
            stage_cell(ln.act_b, sa + (unsigned)sizeof(NF), ss.b, ls.b, Ts.b);
        }
        // stage 2: tendencies at the stage, its temperature boundary values taken at t + dt (heun.jl:52-59); with GENERIC every
        // boundary kind and value of the stage's view
        const NF bTb2 = (!GENERIC && vTb) ? col_ld(a.bcT_bot_stage, ib_late) : NF(0), bTt2 = (!GENERIC && vTt) ? col_ld(a.bcT_top_stage, ib_late) : NF(0);
        uint32_t vs_a = 0, vs_b = 0;
        const Tend t2 = tendencies(kernarg_reload<View<NF>>(GENERIC ? off_vs : 0u), kernarg_reload<DevParams<NF>>(off_p), Ts, ls, ss, ps, bTb2, bTt2, RICHARDS, vs_a, vs_b);
        viol_a |= vs_a; viol_b |= vs_b;
        // average_tendencies! (heun.jl:27-35), then the step of the STATE with its own boundary fluxes
        gU = Two<NF>{(G1U.a + t2.gU.a) / NF(2), (G1U.b + t2.gU.b) / NF(2)};
        gS = RICHARDS ? Two<NF>{(G1S.a + t2.gS.a) / NF(2), (G1S.b + t2.gS.b) / NF(2)} : Two<NF>{NF(0), NF(0)};
    }
    over = advance(gU, gS, flux_U, flux_S, Un, sn, z0);
    closure(Un, sn, z0, ln_, Tn, psin);
    NF S_multi = S_in, GS_multi = NF(0);
    if (PROG == PROG_MULTI) {
        if (RICHARDS) {   // surface_excess_water carried in the register: tendency min(0, S), Euler update, overflow -- per step
            GS_multi = NF(0) + jl_min(NF(0), S_multi);
            S_multi = (S_multi + GS_multi * dt) + over;
        }
        for (int step = 1; step < a.nsteps; ++step) {
            // (the loop reads its kernel arguments afresh every iteration, see kernarg_reload)
            U = Un; sat = sn; T = Tn; liq = ln_; psi = psin;
            t = tendencies(kernarg_reload<View<NF>>(0), kernarg_reload<DevParams<NF>>(off_p), T, liq, sat, psi, bTb, bTt, need_kc, viol_a, viol_b);
            gU = t.gU; gS = t.gS;
            over = advance(gU, gS, flux_U, flux_S, Un, sn, z0);
            closure(Un, sn, z0, ln_, Tn, psin);
            if (RICHARDS) {
                GS_multi = NF(0) + jl_min(NF(0), S_multi);
                S_multi = (S_multi + GS_multi * dt) + over;
            }
        }
    }

    Two<NF> Kf_out = t.Kf_lo;
    const Two<NF> Kc = t.Kc;
    NF Kf_out_top = ln.top_a ? Kc.a : Kc.b;
    if (finalize && write_kf) {
        const DevParams<NF>& pf = kernarg_reload<DevParams<NF>>(off_p);
        const Two<NF> Kn{conductivity_hydraulic<NF, HYD, false>(pf, ln_.a, fractions(pf, sn.a, ln_.a, viol_a)),
                         conductivity_hydraulic<NF, HYD, false>(pf, ln_.b, fractions(pf, sn.b, ln_.b, viol_b))};
        const Two<NF> Kn_dn = below(Kn);
        const NF Kmin_a = jl_min(Kn.a, Kn_dn.a), Kmin_b = jl_min(Kn.b, Kn_dn.b);
        Kf_out.a = (ln.bot_a || ln.top_a) ? Kn.a : Kmin_a;
        Kf_out.b = ln.top_b ? Kn.b : Kmin_b;
        Kf_out_top = ln.top_a ? Kn.a : Kn.b;
    }
    // surface_excess_water after the step, formed before the first store (column_program)
    NF S = NF(0), GS = NF(0), S_stage = NF(0);
    if (RICHARDS && PROG == PROG_MULTI) {
        S = S_multi;
        GS = GS_multi;
    } else if (RICHARDS) {
        S = S_in;
        GS = NF(0) + jl_min(NF(0), S);
        if (PROG == PROG_HEUN) {
            S_stage = (S + GS * dt) + over_stage;
            GS = (GS + (NF(0) + jl_min(NF(0), S_stage))) / NF(2);
        }
        S = (S + GS * dt) + over;
    }
    // ---- the column goes out -------------------------------------------------------------------------------------------
    const View<NF>& vo = kernarg_reload<View<NF>>(0);
    auto store_cell = [&](bool act, unsigned cb_, NF u, NF t_, NF l, NF s, NF ps, NF kf, NF gu, NF gs) {
        if (!act) return;
        const unsigned cb = block_local(cb_);
        stg(vo.U, cb, u);
        stg(vo.T, cb, t_);
        stg(vo.liq, cb, l);
        if (RICHARDS) { stg(vo.sat, cb, s); stg(vo.psi, cb, ps); }
        if (finalize) {
            stg(vo.G_U, cb, gu);
            if (RICHARDS) stg(vo.G_sat, cb, gs);
        }
        if (write_kf) stg(vo.Kf, cb, kf);
    };
    const unsigned cba = ((unsigned)ii * (unsigned)v.Nzp + (unsigned)ln.ka) * (unsigned)sizeof(NF), cbb = cba + (unsigned)sizeof(NF);
    store_cell(ln.act_a, cba, Un.a, Tn.a, ln_.a, sn.a, psin.a, Kf_out.a, gU.a, gS.a);
    store_cell(ln.act_b, cbb, Un.b, Tn.b, ln_.b, sn.b, psin.b, Kf_out.b, gU.b, gS.b);
    if (colok && is_top_lane) {
        const unsigned ib = block_local(ib0);
        const NF Tt = ln.top_a ? Tn.a : Tn.b, st = ln.top_a ? sn.a : sn.b, lt = ln.top_a ? ln_.a : ln_.b;
        if (write_kf) stg(vo.Kf_top, ib, Kf_out_top);
        if (RICHARDS) {
            stg(vo.S, ib, S);
            stg(vo.wt, ib, z0);
            if (finalize) stg(vo.G_S, ib, GS);
            if (PROG == PROG_HEUN && a.stage_S) stg(a.stage_S, ib, S_stage);
        }
        if (seb) {
            stg(vo.top_T, ib, Tt);
            stg(vo.top_sat, ib, st);
            stg(vo.top_liq, ib, lt);
            stg(vo.Ts, ib, Ts_in + NF(0) * dt);
        }
    }
    const uint32_t viol = (ln.act_a ? viol_a : 0u) | (ln.act_b ? viol_b : 0u) | (bad ? 1u : 0u);
    if (viol) atomicOr(v_arg.status, viol);
}

}  // namespace trm
