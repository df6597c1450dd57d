// trm_column_wide.hpp -- the fused step for columns of 129 ... 256 levels: M = 4 soil levels per lane, one column per wavefront.
//
// k_column_deep (two levels per lane, trm_column_deep.hpp) generalised over the number of levels a lane holds: lane l owns levels
// M l ... M l + M - 1 of ONE column -- a field is one M-word access per lane, the k - 1 neighbour of a lane's lowest cell is the
// previous lane's highest (one DPP shift), of the others the lane's own cell below (free); likewise upwards.  Ballots come in
// sets of M (one per cell slot); level q is cell (lane q / M, slot q % M).  Every operation is the one k_column / k_column_deep /
// the reference-order kernels perform on the cell, in the same order: results are bit-identical (tests/test_gpu_deep_columns.py).
// Scope: ForwardEuler and Heun (both stages in registers, one launch per step), every boundary kind (GENERIC); no multi-step
// program, no derivation of T / liq (a 256-level state is far beyond the cache whatever is done: the kernel is there so that such
// grids do not fall to the reference-order kernels, 3 - 10 x slower).  Deeper than 256 levels: the reference-order kernels.
#pragma once
#include "trm_column.hpp"

namespace trm {

template <class NF, int M> struct Lv { NF x[M]; };
template <class NF, int M> TRM_DEV Lv<NF, M> lv_fill(NF a) { Lv<NF, M> r; for (int j = 0; j < M; ++j) r.x[j] = a; return r; }
// level k - 1 / k + 1 of every cell of a lane
template <class NF, int M> TRM_DEV Lv<NF, M> below(const Lv<NF, M>& v) {
    Lv<NF, M> r;
    r.x[0] = shift_up(v.x[M - 1]);
    for (int j = 1; j < M; ++j) r.x[j] = v.x[j - 1];
    return r;
}
template <class NF, int M> TRM_DEV Lv<NF, M> above(const Lv<NF, M>& v) {
    Lv<NF, M> r;
    for (int j = 0; j + 1 < M; ++j) r.x[j] = v.x[j + 1];
    r.x[M - 1] = shift_dn(v.x[0]);
    return r;
}
template <class NF, int M> TRM_DEV Lv<NF, M> ld_cells(const NF* base, unsigned byte_off) {
    struct alignas(16) Raw { NF x[M]; };      // (the group of M levels is adjacent in the z-fastest layout: 16-byte accesses)
    const Raw r = *reinterpret_cast<const Raw*>(reinterpret_cast<const char*>(base) + byte_off);
    Lv<NF, M> o;
    for (int j = 0; j < M; ++j) o.x[j] = r.x[j];
    return o;
}
// one ballot per cell slot: bit l of b[j] = the predicate of level M l + j
template <int M> struct LevelSet { unsigned long long b[M]; };
template <int M> TRM_DEV bool any(const LevelSet<M>& s) { unsigned long long a = 0; for (int j = 0; j < M; ++j) a |= s.b[j]; return a != 0ull; }
template <int M> TRM_DEV int lowest(const LevelSet<M>& s) {
    int q = 1 << 30;
    for (int j = 0; j < M; ++j) if (s.b[j]) { const int c = M * __builtin_ctzll(s.b[j]) + j; q = c < q ? c : q; }
    return q;
}
template <int M> TRM_DEV int highest(const LevelSet<M>& s) {
    int q = -1;
    for (int j = 0; j < M; ++j) if (s.b[j]) { const int c = M * (63 - __builtin_clzll(s.b[j])) + j; q = c > q ? c : q; }
    return q;
}
template <int M> TRM_DEV bool any_above(const LevelSet<M>& s, int q) {      // a level > q in the set
    bool r = false;
    for (int j = 0; j < M; ++j) {
        // M l + j > q  <=>  l >= first, first = floor((q - j) / M) + 1 (0 when q < j)
        const int first = q < j ? 0 : (q - j) / M + 1;
        r = r || (first < 64 && (s.b[j] >> first) != 0ull);
    }
    return r;
}
template <int M> TRM_DEV bool any_below(const LevelSet<M>& s, int q) {      // a level < q in the set
    bool r = false;
    for (int j = 0; j < M; ++j) {
        // M l + j < q  <=>  l < count, count = ceil((q - j) / M) (none when q <= j)
        const int count = q <= j ? 0 : (q - j + M - 1) / M;
        r = r || (count > 0 && (count >= 64 ? s.b[j] : (s.b[j] & ((1ull << count) - 1ull))) != 0ull);
    }
    return r;
}
// broadcast of a per-level value from level q (lane q / M, slot q % M) to the wave
template <class NF, int M> TRM_DEV NF from_level(const Lv<NF, M>& v, int q) {
    NF pick = v.x[0];
    for (int j = 1; j < M; ++j) pick = (q % M == j) ? v.x[j] : pick;      // (wave-uniform select of the slot)
    return __shfl(pick, q / M, 64);
}
template <int M> struct WideLane { int lane, k0; bool act[M], top[M], bot; };

// adjust_saturation_profile! (soil_hydrology.jl:185-219), M levels per lane: repair_saturation() of trm_kernels.hpp with the level
// sets as M ballots and the cell of level q addressed as (lane q / M, slot q % M)
template <class NF, int M>
TRM_DEV NF repair_saturation_wide(Lv<NF, M>& s, const WideLane<M>& ln, int Nz, const Lv<NF, M>& dzc, const Lv<NF, M>& rdzc, NF dzc_top) {
    LevelSet<M> over, bad;
    for (int j = 0; j < M; ++j) {
        const bool o = ln.act[j] && !ln.top[j] && s.x[j] > NF(1);
        const bool u = ln.act[j] && !(ln.bot && j == 0) && s.x[j] < NF(0);
        over.b[j] = wave_ballot(o);
        bad.b[j] = wave_ballot(o || u);
    }
    for (int j = 0; j < M; ++j) s.x[j] = (ln.bot && j == 0) ? s.x[j] : s.x[j] + NF(0);
    if (any(bad)) {
        // thickness of the cells above / below (the edge cells keep their own, as neighbour_dz() does)
        const Lv<NF, M> dz_up = above(dzc), rdz_up = above(rdzc), dz_dn = below(dzc), rdz_dn = below(rdzc);
        Lv<NF, M> nb_dz_up, nb_rdz_up, nb_dz_dn, nb_rdz_dn;
        for (int j = 0; j < M; ++j) {
            const bool edge_up = ln.k0 + j >= Nz - 1, edge_dn = ln.bot && j == 0;
            nb_dz_up.x[j] = edge_up ? dzc.x[j] : dz_up.x[j];
            nb_rdz_up.x[j] = edge_up ? rdzc.x[j] : rdz_up.x[j];
            nb_dz_dn.x[j] = edge_dn ? dzc.x[j] : dz_dn.x[j];
            nb_rdz_dn.x[j] = edge_dn ? rdzc.x[j] : rdz_dn.x[j];
        }
        if (any(over)) {
            NF carry = NF(0);
            for (int q = lowest(over); q < Nz - 1; ++q) {
                Lv<NF, M> cout = lv_fill<NF, M>(NF(0));
                for (int j = 0; j < M; ++j)
                    if (ln.k0 + j == q) {
                        s.x[j] = s.x[j] + carry;
                        const NF e = jl_max(s.x[j] - NF(1), NF(0));
                        s.x[j] = s.x[j] - e;
                        cout.x[j] = div_const(e * dzc.x[j], nb_dz_up.x[j], nb_rdz_up.x[j]);
                    }
                carry = from_level(cout, q);
                if (!any_above(over, q) && wave_ballot(!(carry == NF(0))) == 0ull) break;
            }
            for (int j = 0; j < M; ++j) if (ln.top[j]) s.x[j] = s.x[j] + carry;
        }
        LevelSet<M> under;
        for (int j = 0; j < M; ++j) under.b[j] = wave_ballot(ln.act[j] && !(ln.bot && j == 0) && !(jl_max(-s.x[j], NF(0)) == NF(0)));
        if (any(under)) {
            NF pend = NF(0);
            for (int q = highest(under); q >= 1; --q) {
                Lv<NF, M> pout = lv_fill<NF, M>(NF(0));
                for (int j = 0; j < M; ++j)
                    if (ln.k0 + j == q) {
                        s.x[j] = s.x[j] - pend;
                        const NF d = jl_max(-s.x[j], NF(0));
                        s.x[j] = s.x[j] + d;
                        pout.x[j] = div_const(d * dzc.x[j], nb_dz_dn.x[j], nb_rdz_dn.x[j]);
                    }
                pend = from_level(pout, q);
                if (!any_below(under, q) && wave_ballot(!(pend == NF(0))) == 0ull) break;
            }
            if (ln.bot) s.x[0] = s.x[0] - pend;
        }
    }
    // surface overflow joins surface_excess_water (the top cell); bottom clamp (the bottom cell)
    NF e_sum = NF(0);
    for (int j = 0; j < M; ++j) {
        const NF e = ln.top[j] ? jl_max(s.x[j] - NF(1), NF(0)) : NF(0);
        s.x[j] = s.x[j] - e;
        e_sum = j == 0 ? e : e_sum + e;      // (one of them is the top cell's excess, the others +0)
    }
    s.x[0] = ln.bot ? jl_max(s.x[0], NF(0)) : s.x[0];
    return e_sum * dzc_top;
}

// PROG: PROG_EULER or PROG_HEUN.  GENERIC: every boundary kind, the per-cell vwc_forcing field.  `vs_arg`: the Heun stage's view
// (its boundary kinds and values serve the stage's tendencies with GENERIC; see k_column_deep).
template <class NF, bool RICHARDS, int HYD, int M, int PROG = PROG_EULER, bool GENERIC = false>
__global__ void __launch_bounds__(TRM_STEP_BLOCK) __attribute__((amdgpu_waves_per_eu(1, 8)))
    k_column_wide(View<NF> v_arg, DevParams<NF> p_arg, ColumnArgs<NF> a, View<NF> vs_arg) {
    static_assert(PROG == PROG_EULER || PROG == PROG_HEUN, "k_column_wide: one step per launch");
    constexpr unsigned off_p = round_up_to((unsigned)sizeof(View<NF>), (unsigned)alignof(DevParams<NF>));
    constexpr unsigned off_a = round_up_to(off_p + (unsigned)sizeof(DevParams<NF>), (unsigned)alignof(ColumnArgs<NF>));
    constexpr unsigned off_vs = round_up_to(off_a + (unsigned)sizeof(ColumnArgs<NF>), (unsigned)alignof(View<NF>));
    (void)off_vs;
    typedef Lv<NF, M> L4;
    const View<NF>& v = v_arg;
    const DevParams<NF>& p = p_arg;
    WideLane<M> ln;
    ln.lane = threadIdx.x & 63;
    const int i = (int)((blockIdx.x * (unsigned)blockDim.x + threadIdx.x) >> 6);      // one column per wave
    const int Nz = v.Nz, Nh = (int)v.Nh;
    ln.k0 = M * ln.lane;
    const bool colok = i < Nh;
    ln.bot = ln.lane == 0;
    bool top_lane = false;
    for (int j = 0; j < M; ++j) {
        ln.act[j] = colok && ln.k0 + j < Nz;
        ln.top[j] = ln.k0 + j == Nz - 1;
        top_lane = top_lane || ln.top[j];
    }
    const int ii = colok ? i : Nh - 1;
    // Lanes beyond the top load the column's last group again (an aligned M-word access inside the column's pitch: the pitch is a
    // multiple of 32) and store nothing.  Slots of the top lane beyond the top cell are padding words: read, never stored, and
    // nothing of them reaches a real cell (the top cell takes its upper face, conductivities and fluxes from the boundary formulas).
    const int last_group = ((Nz - 1) / M) * M;
    const int k_ld = ln.k0 <= last_group ? ln.k0 : last_group;
    const unsigned ib0 = (unsigned)ii * (unsigned)sizeof(NF);
    // (one column per wave: the per-column inputs through the scalar memory path, as in k_column_deep -- profiles/r04/exp18)
#if TRM_DEEP_SCALAR_INPUTS
    const unsigned ib_u = (unsigned)__builtin_amdgcn_readfirstlane((int)ib0);
    auto col_ld = [&](const NF* ptr, unsigned) -> NF { return sld_off<NF>(ptr, ib_u); };
#else
    auto col_ld = [&](const NF* ptr, unsigned off) -> NF { return ldg(ptr, off); };
#endif
    const unsigned cb0 = ((unsigned)ii * (unsigned)v.Nzp + (unsigned)k_ld) * (unsigned)sizeof(NF);
    const NF dt = a.dt;
    const int finalize = a.finalize, write_kf = a.write_kf;
    const bool need_kc = RICHARDS || write_kf;
    uint32_t viol[M];
    for (int j = 0; j < M; ++j) viol[j] = 0;
    bool bad = false;

    // per-level geometry of the lane's cells (the level records of trm_kernels.hpp: level_geom): thickness, its reciprocal, the
    // face reciprocals here; zC, psiz and zFlo at the END of a stage (level_word), as in k_column_deep
    unsigned rec[M];
    L4 dzc, rdzc, rdzf_lo, rdzf_hi;
    for (int j = 0; j < M; ++j) {
        const int kk = ln.k0 + j < Nz ? ln.k0 + j : Nz - 1;
        rec[j] = (unsigned)kk * (unsigned)sizeof(LevelPack<NF>);
        const NF* q = reinterpret_cast<const NF*>(reinterpret_cast<const char*>(v.lvl) + rec[j]);
        dzc.x[j] = q[3]; rdzc.x[j] = q[4]; rdzf_lo.x[j] = q[5]; rdzf_hi.x[j] = q[6];
    }
    auto level_word = [&](const View<NF>& vw, unsigned r, int w) {
        return *reinterpret_cast<const NF*>(reinterpret_cast<const char*>(vw.lvl) + r + (unsigned)w * (unsigned)sizeof(NF));
    };

    // ---- the column comes in ---------------------------------------------------------------------------------------
    const L4 U = ld_cells<NF, M>(v.U, cb0), sat = ld_cells<NF, M>(v.sat, cb0);
    const L4 psi = RICHARDS ? ld_cells<NF, M>(v.psi, cb0) : lv_fill<NF, M>(NF(0));
    const L4 T = ld_cells<NF, M>(v.T, cb0), liq = ld_cells<NF, M>(v.liq, cb0);
    const bool seb = p.seb != 0;
    const bool vTb = v.bc.kind[2][0] == 1, vTt = v.bc.kind[2][1] == 1;
    const NF bTb = vTb ? col_ld(bcval(v, 2, 0), ib0) : NF(0), bTt = vTt ? col_ld(bcval(v, 2, 1), ib0) : NF(0);

    // ---- compute_auxiliary! + compute_tendencies! (column_tendencies / column_tendencies_generic, trm_column.hpp, per cell) -----
    struct Tend { L4 gU, gS, Kf_lo, Kc; };
    auto tendencies = [&](const View<NF>& v, const DevParams<NF>& p, const L4& T, const L4& liq, const L4& sat, const L4& psi, NF bTb, NF bTt, bool need_kc,
                          uint32_t* viol) {
        L4 kap, Kc;
        for (int j = 0; j < M; ++j) {
            const Frac<NF> f = fractions_unchecked(p, sat.x[j], liq.x[j]);
            kap.x[j] = conductivity(p, f);
            Kc.x[j] = need_kc ? conductivity_hydraulic<NF, HYD, false>(p, liq.x[j], f) : NF(0);
        }
        const L4 T_dn = below(T), kap_dn = below(kap);
        L4 T_m = T_dn, kap_m = kap_dn, T_h = lv_fill<NF, M>(NF(0)), kap_h = lv_fill<NF, M>(NF(0)), psi_ht = lv_fill<NF, M>(NF(0));
        NF psi_hb = NF(0);
        if constexpr (GENERIC) {
            const bool same_bot = v.bc.kind[3][0] != 1 && v.bc.kind[3][0] != 3 && (RICHARDS ? (v.bc.kind[1][0] != 1 && v.bc.kind[1][0] != 3) : p.halo_policy == 1);
            const bool same_top = v.bc.kind[3][1] != 1 && v.bc.kind[3][1] != 3 && (RICHARDS ? (v.bc.kind[1][1] != 1 && v.bc.kind[1][1] != 3) : p.halo_policy == 1);
            if (ln.bot) {
                T_m.x[0] = halo_bottom(v.bc.kind[2][0], bcval(v, 2, 0), ii, T.x[0], v.g);
                kap_m.x[0] = kap.x[0];
                if (!same_bot) {
                    const NF lh = halo_bottom(v.bc.kind[3][0], bcval(v, 3, 0), ii, liq.x[0], v.g);
                    const NF sh = sat_halo<NF, RICHARDS>(v, p, 0, ii, sat.x[0]);
                    kap_m.x[0] = conductivity(p, fractions(p, sh, lh, viol[0]));
                }
                if (RICHARDS) psi_hb = halo_bottom(v.bc.kind[4][0], bcval(v, 4, 0), ii, psi.x[0], v.g);
            }
            for (int j = 0; j < M; ++j) {
                if (!ln.top[j]) continue;
                T_h.x[j] = halo_top(v.bc.kind[2][1], bcval(v, 2, 1), ii, T.x[j], v.g);
                kap_h.x[j] = kap.x[j];
                if (!same_top) {
                    const NF lh = halo_top(v.bc.kind[3][1], bcval(v, 3, 1), ii, liq.x[j], v.g);
                    const NF sh = sat_halo<NF, RICHARDS>(v, p, 1, ii, sat.x[j]);
                    kap_h.x[j] = conductivity(p, fractions(p, sh, lh, viol[j]));
                }
                if (RICHARDS) psi_ht.x[j] = halo_top(v.bc.kind[4][1], bcval(v, 4, 1), ii, psi.x[j], v.g);
            }
        } else {
            auto ext_b = [&](NF Tc) { return vTb ? Tc + div_const(Tc - bTb, v.g.hdzf_bot, v.g.rhdzf_bot) * (-v.g.dzf_bot) : Tc; };
            auto ext_t = [&](NF Tc) { return vTt ? Tc + div_const(bTt - Tc, v.g.hdzf_top, v.g.rhdzf_top) * v.g.dzf_top : Tc; };
            for (int j = 0; j < M; ++j) {
                kap_h.x[j] = (!RICHARDS && p.halo_policy != 1) ? conductivity(p, fractions(p, NF(0), liq.x[j], viol[j])) : kap.x[j];
                T_h.x[j] = ext_t(T.x[j]);
            }
            T_m.x[0] = ln.bot ? ext_b(T.x[0]) : T_dn.x[0];
            kap_m.x[0] = ln.bot ? kap_h.x[0] : kap_dn.x[0];
        }
        L4 qT_lo;
        for (int j = 0; j < M; ++j) qT_lo.x[j] = -(NF(0.5) * (kap.x[j] + kap_m.x[j])) * ((T.x[j] - T_m.x[j]) * rdzf_lo.x[j]);
        const L4 qT_up = above(qT_lo);
        Tend t;
        for (int j = 0; j < M; ++j) {
            const NF qT_hi = ln.top[j] ? -(NF(0.5) * (kap_h.x[j] + kap.x[j])) * ((T_h.x[j] - T.x[j]) * rdzf_hi.x[j]) : qT_up.x[j];
            t.gU.x[j] = NF(0) + (-((qT_hi - qT_lo.x[j]) * rdzc.x[j]));
            t.gS.x[j] = NF(0);
            t.Kf_lo.x[j] = NF(0);
        }
        t.Kc = Kc;
        if (need_kc) {   // face conductivities (soil_hydrology.jl:145-163)
            const L4 Kc_dn = below(Kc);
            for (int j = 0; j < M; ++j) {
                const NF Kmin = jl_min(Kc.x[j], Kc_dn.x[j]);
                t.Kf_lo.x[j] = ((ln.bot && j == 0) || ln.top[j]) ? Kc.x[j] : Kmin;
            }
        }
        if (RICHARDS) {  // Darcy fluxes (soil_hydrology_rre.jl:95-131)
            const L4 Kf_lo = t.Kf_lo;
            const L4 Kf_dn = below(Kf_lo), Kf_up = above(Kf_lo), psi_dn = below(psi);
            L4 qW_lo;
            for (int j = 0; j < M; ++j) {
                const bool b = ln.bot && j == 0;
                const NF Kf_m = b ? NF(0) : Kf_dn.x[j];
                const NF Kf_p = ln.top[j] ? Kc.x[j] : Kf_up.x[j];
                const NF psi_m = b ? (GENERIC ? psi_hb : psi.x[j]) : psi_dn.x[j];
                const NF g_lo = (psi.x[j] - psi_m) * rdzf_lo.x[j];
                qW_lo.x[j] = -upwind_conductivity(g_lo, Kf_m, Kf_lo.x[j], Kf_p) * g_lo;
            }
            const L4 qW_up = above(qW_lo);
            L4 F = lv_fill<NF, M>(p.vwc_forcing);
            if constexpr (GENERIC) {
                if (v.Fvwc) F = ld_cells<NF, M>(v.Fvwc, cb0);
            }
            for (int j = 0; j < M; ++j) {
                NF qW_t;
                if constexpr (GENERIC) {
                    const NF g_t = (psi_ht.x[j] - psi.x[j]) * rdzf_hi.x[j];
                    qW_t = -upwind_conductivity(g_t, Kf_lo.x[j], Kc.x[j], NF(0)) * g_t;
                } else {
                    qW_t = -jl_min(Kc.x[j], NF(0)) * (psi.x[j] - psi.x[j]);
                }
                const NF qW_hi = ln.top[j] ? qW_t : qW_up.x[j];
                const NF dth = -((qW_hi - qW_lo.x[j]) * rdzc.x[j]) + NF(0) + F.x[j];
                t.gS.x[j] = NF(0) + div_const(dth, p.por, p.rpor);
            }
        }
        return t;
    };
    // ---- compute_z_bcs! + explicit_step! + the hydrology closure's repair and water table of the STATE's (U, sat) ------------------
    auto advance = [&](L4& gU, L4& gS, const L4& flux_U, const L4& flux_S, L4& Un, L4& sn, NF& z0) {
        for (int j = 0; j < M; ++j) {
            gU.x[j] += flux_U.x[j];
            Un.x[j] = U.x[j] + gU.x[j] * dt;
            bad = bad || (ln.act[j] && is_nan(Un.x[j]));
        }
        sn = sat;
        z0 = NF(0);
        NF over = NF(0);
        if (RICHARDS) {
            for (int j = 0; j < M; ++j) {
                gS.x[j] += flux_S.x[j];
                sn.x[j] = sat.x[j] + gS.x[j] * dt;
                bad = bad || (ln.act[j] && is_nan(sn.x[j]));
            }
            over = repair_saturation_wide<NF, M>(sn, ln, Nz, dzc, rdzc, v.g.dzc_top);
            // compute_water_table! (soil_hydrology.jl:170-175): lower face of the first unsaturated cell from the bottom
            LevelSet<M> unsat;
            L4 zFlo;
            for (int j = 0; j < M; ++j) {
                unsat.b[j] = wave_ballot(ln.act[j] && sn.x[j] < NF(1));
                zFlo.x[j] = level_word(v, rec[j], 2);
            }
            const bool found = any(unsat);
            const NF z_first = from_level(zFlo, found ? lowest(unsat) : 0);
            z0 = found ? z_first : v.g.zF_top;
        }
        return over;
    };
    // ---- closures: (U, sat) -> (T, liq, psi), parameters fetched afresh ---------------------------------------------------------
    auto closure = [&](const L4& Un, const L4& sn, NF z0, L4& ln_, L4& Tn, L4& psin) {
        const DevParams<NF>& p2 = kernarg_reload<DevParams<NF>>(off_p);
        for (int j = 0; j < M; ++j) {
            energy_closure_wave(p2, Un.x[j], sn.x[j], ln_.x[j], Tn.x[j], viol[j]);
            psin.x[j] = RICHARDS ? pressure_head<NF, HYD>(p2, sn.x[j], level_word(v, rec[j], 0), level_word(v, rec[j], 1), z0) : NF(0);
        }
    };

    Tend t = tendencies(v, p, T, liq, sat, psi, bTb, bTt, need_kc, viol);     // tendencies at the STATE (hydraulic_conductivity comes from here)
    // the boundary flux terms and the 0-D inputs: after the first stencil, in front of every store (column_program)
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
    L4 flux_U, flux_S;
    for (int j = 0; j < M; ++j) {
        const bool b = ln.bot && j == 0;
        flux_U.x[j] = b ? eU_b : (ln.top[j] ? eU_t : NF(0));
        flux_S.x[j] = b ? eS_b : (ln.top[j] ? eS_t : NF(0));
    }

    L4 gU = t.gU, gS = t.gS, Un, sn, ln_, Tn, psin;
    NF z0 = NF(0), over = NF(0), over_stage = NF(0);
    if (PROG == PROG_HEUN) {
        // stage 1: Euler predictor with the state's boundary fluxes and its closures (the stage never leaves the registers)
        const L4 G1U = t.gU, G1S = t.gS;
        L4 Us, ss, ls, Ts, ps;
        NF z0s;
        over_stage = advance(gU, gS, flux_U, flux_S, Us, ss, z0s);
        closure(Us, ss, z0s, ls, Ts, ps);
        if (a.stage_T) {   // (wave-uniform: the vegetation-coupled LandModel evaluates its 0-D processes AT the stage: k_column<PROG_HEUN>)
            const unsigned cbs = ((unsigned)ii * (unsigned)v.Nzp + (unsigned)ln.k0) * (unsigned)sizeof(NF);
            for (int j = 0; j < M; ++j) {
                if (!ln.act[j]) continue;
                const unsigned cb = block_local(cbs + (unsigned)j * (unsigned)sizeof(NF));
                stg(a.stage_sat, cb, ss.x[j]);
                stg(a.stage_liq, cb, ls.x[j]);
                stg(a.stage_T, cb, Ts.x[j]);
            }
        }
        // stage 2: tendencies at the stage, its temperature boundary values taken at t + dt (heun.jl:52-59); with GENERIC every
        // boundary kind and value of the stage's view
        const NF bTb2 = (!GENERIC && vTb) ? col_ld(a.bcT_bot_stage, ib_late) : NF(0), bTt2 = (!GENERIC && vTt) ? col_ld(a.bcT_top_stage, ib_late) : NF(0);
        uint32_t vs[M];
        for (int j = 0; j < M; ++j) vs[j] = 0;
        const Tend t2 = tendencies(kernarg_reload<View<NF>>(GENERIC ? off_vs : 0u), kernarg_reload<DevParams<NF>>(off_p), Ts, ls, ss, ps, bTb2, bTt2, RICHARDS, vs);
        for (int j = 0; j < M; ++j) {
            viol[j] |= vs[j];
            // average_tendencies! (heun.jl:27-35), then the step of the STATE with its own boundary fluxes
            gU.x[j] = (G1U.x[j] + t2.gU.x[j]) / NF(2);
            gS.x[j] = RICHARDS ? (G1S.x[j] + t2.gS.x[j]) / NF(2) : NF(0);
        }
    }
    over = advance(gU, gS, flux_U, flux_S, Un, sn, z0);
    closure(Un, sn, z0, ln_, Tn, psin);

    L4 Kf_out = t.Kf_lo;
    L4 Kc_top = t.Kc;
    if (finalize && write_kf) {
        const DevParams<NF>& pf = kernarg_reload<DevParams<NF>>(off_p);
        L4 Kn;
        for (int j = 0; j < M; ++j) Kn.x[j] = conductivity_hydraulic<NF, HYD, false>(pf, ln_.x[j], fractions(pf, sn.x[j], ln_.x[j], viol[j]));
        const L4 Kn_dn = below(Kn);
        for (int j = 0; j < M; ++j) {
            const NF Kmin = jl_min(Kn.x[j], Kn_dn.x[j]);
            Kf_out.x[j] = ((ln.bot && j == 0) || ln.top[j]) ? Kn.x[j] : Kmin;
        }
        Kc_top = Kn;
    }
    // surface_excess_water after the step, formed before the first store (column_program)
    NF S = NF(0), GS = NF(0), S_stage = NF(0);
    if (RICHARDS) {
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
    const unsigned cba = ((unsigned)ii * (unsigned)v.Nzp + (unsigned)ln.k0) * (unsigned)sizeof(NF);
    for (int j = 0; j < M; ++j) {
        if (!ln.act[j]) continue;
        const unsigned cb = block_local(cba + (unsigned)j * (unsigned)sizeof(NF));
        stg(vo.U, cb, Un.x[j]);
        stg(vo.T, cb, Tn.x[j]);
        stg(vo.liq, cb, ln_.x[j]);
        if (RICHARDS) { stg(vo.sat, cb, sn.x[j]); stg(vo.psi, cb, psin.x[j]); }
        if (finalize) {
            stg(vo.G_U, cb, gU.x[j]);
            if (RICHARDS) stg(vo.G_sat, cb, gS.x[j]);
        }
        if (write_kf) stg(vo.Kf, cb, Kf_out.x[j]);
    }
    if (colok && top_lane) {
        const unsigned ib = block_local(ib0);
        NF Tt = Tn.x[0], st = sn.x[0], lt = ln_.x[0], Kt = Kc_top.x[0];
        for (int j = 1; j < M; ++j) {
            Tt = ln.top[j] ? Tn.x[j] : Tt; st = ln.top[j] ? sn.x[j] : st; lt = ln.top[j] ? ln_.x[j] : lt; Kt = ln.top[j] ? Kc_top.x[j] : Kt;
        }
        if (write_kf) stg(vo.Kf_top, ib, Kt);
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
    uint32_t flags = bad ? 1u : 0u;
    for (int j = 0; j < M; ++j) flags |= ln.act[j] ? viol[j] : 0u;
    if (flags) atomicOr(v_arg.status, flags);
}

}  // namespace trm
