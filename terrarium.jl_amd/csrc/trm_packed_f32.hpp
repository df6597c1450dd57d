// Fused Euler step in fp32 with TWO columns per lane (packed math).
//
// fp32 instructions issue at the fp64 rate unless they are the packed v_pk_{add,mul,fma}_f32 forms, which do two
// operations per lane.  The step kernel is bound by instruction issue, so the 0.1-degree fp32 configuration
// (BASELINE C5) gains nothing from its halved bytes with the scalar kernel.  Here every lane carries the SAME soil
// level of two neighbouring columns in a 2-vector: sums, products and fmas become packed instructions, the per-wave
// overhead (kernel arguments, per-level records, wave-uniform boundary logic) is shared by twice the cells; compares,
// selects, divides, min/max, DPP moves and ballots stay per component.
//
// Scope: trm_step with TRM_F32, reference-default hydraulics (BrooksCorey with r^(-5), linear K), the branch-free
// boundary kinds, Nz <= 64.  Everything else takes the scalar kernels.  Every operation is the scalar kernel's, in
// the same order (k_step_wave, trm_kernels.hpp), so the two agree bit for bit
// (tests/test_gpu_parity.py::test_packed_fp32_equals_scalar_bitwise).
#pragma once
#include "trm_kernels.hpp"

namespace trm {
namespace pk {

typedef float v2f __attribute__((ext_vector_type(2)));
struct M2 { bool x, y; };   // per-component predicate

TRM_DEV v2f splat(float a) { return v2f{a, a}; }
TRM_DEV v2f sel(M2 m, v2f a, v2f b) { return v2f{m.x ? a.x : b.x, m.y ? a.y : b.y}; }
TRM_DEV v2f sel(bool m, v2f a, v2f b) { return v2f{m ? a.x : b.x, m ? a.y : b.y}; }   // lane predicate (level)
TRM_DEV M2 lt(v2f a, v2f b) { return M2{a.x < b.x, a.y < b.y}; }
TRM_DEV M2 ge(v2f a, v2f b) { return M2{a.x >= b.x, a.y >= b.y}; }
TRM_DEV M2 eq(v2f a, v2f b) { return M2{a.x == b.x, a.y == b.y}; }
TRM_DEV M2 operator||(M2 a, M2 b) { return M2{a.x || b.x, a.y || b.y}; }
TRM_DEV v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
TRM_DEV v2f min2(v2f a, v2f b) { return v2f{jl_min(a.x, b.x), jl_min(a.y, b.y)}; }
TRM_DEV v2f max2(v2f a, v2f b) { return v2f{jl_max(a.x, b.x), jl_max(a.y, b.y)}; }
TRM_DEV v2f div2(v2f a, v2f b) { return v2f{div_nr(a.x, b.x), div_nr(a.y, b.y)}; }
TRM_DEV v2f up2(v2f a) { return v2f{shift_up(a.x), shift_up(a.y)}; }
TRM_DEV v2f dn2(v2f a) { return v2f{shift_dn(a.x), shift_dn(a.y)}; }
TRM_DEV v2f ld2(const float* base, unsigned off0, unsigned off1) { return v2f{ldg(base, off0), ldg(base, off1)}; }

// div_const (trm_device.hpp) on both components
TRM_DEV v2f div_const2(v2f a, float b, float rb) {
    const v2f q = a * rb;
    const v2f r = fma2(-q, splat(b), a);
    const v2f q2 = fma2(r, splat(rb), q);
    return sel(eq(q, splat(0.0f)), q, q2);
}

// div_const_nsz (trm_device.hpp) on both components: no select for the sign of a zero quotient -- where it cannot matter
TRM_DEV v2f div_const2_nsz(v2f a, float b, float rb) {
#if !TRM_CUT_NSZ
    return div_const2(a, b, rb);
#endif
    const v2f q = a * rb;
    const v2f r = fma2(-q, splat(b), a);
    return fma2(r, splat(rb), q);
}

struct Frac2 { v2f water, ice, air; };
TRM_DEV Frac2 fractions2_unchecked(const DevParams<float>& p, v2f sat, v2f liq) {
    Frac2 f;
    const v2f wi = sat * p.por;
    f.water = wi * liq;
    f.ice = wi * (splat(1.0f) - liq);
    f.air = (splat(1.0f) - sat) * p.por;
    return f;
}
TRM_DEV Frac2 fractions2(const DevParams<float>& p, v2f sat, v2f liq, uint32_t& viol) {
    // (bit 1: component x out of bounds, bit 2: component y -- folded into the status flag by the caller, for the
    // components that are real cells only: the copy a tail lane carries is not repaired and may be out of bounds)
    const bool okx = (0.0f <= sat.x && sat.x <= 1.0f) && (0.0f <= liq.x && liq.x <= 1.0f);
    const bool oky = (0.0f <= sat.y && sat.y <= 1.0f) && (0.0f <= liq.y && liq.y <= 1.0f);
    viol |= (okx ? 0u : 2u) | (oky ? 0u : 4u);
    Frac2 f;
    const v2f wi = sat * p.por;
    f.water = wi * liq;
    f.ice = wi * (splat(1.0f) - liq);
    f.air = (splat(1.0f) - sat) * p.por;
    return f;
}
TRM_DEV v2f conductivity2(const DevParams<float>& p, const Frac2& f) {
    v2f s = f.water * p.sk_water;
    s = s + f.ice * p.sk_ice;
    s = s + f.air * p.sk_air;
    s = s + p.kterm_mineral;
    s = s + p.kterm_organic;
    return s * s;
}
TRM_DEV v2f heat_capacity2(const DevParams<float>& p, const Frac2& f) {
    v2f s = f.water * p.c_water;
    s = s + f.ice * p.c_ice;
    s = s + f.air * p.c_air;
    s = s + p.cterm_mineral;
    s = s + p.cterm_organic;
    return s;
}
TRM_DEV v2f conductivity_linear2(const DevParams<float>& p, const Frac2& f) {
    const v2f theta_sat = f.water + f.ice + f.air;
    return div2(f.water * p.K_sat, theta_sat);
}
// liquid fraction of the free-water closure on both components (liquid_fraction_wave, trm_column.hpp)
TRM_DEV v2f liquid_fraction2(const DevParams<float>& p, v2f U, v2f sat) {
    const v2f Lth = sat * p.L * p.por;
    const v2f nLth = -Lth;
    const float eps = Limits<float>::eps();
    const M2 thawed = ge(U, splat(0.0f)), frozen0 = lt(U, nLth);
    const bool needx = !thawed.x && !(frozen0.x && Lth.x > eps), needy = !thawed.y && !(frozen0.y && Lth.y > eps);
    if (wave_ballot(needx || needy) == 0ull) return sel(thawed, splat(1.0f), splat(-0.0f));
    const v2f den = nLth + eps;
    const v2f sd = v2f{(nLth.x == 0.0f) ? Limits<float>::inf() : div_nr(U.x, den.x), (nLth.y == 0.0f) ? Limits<float>::inf() : div_nr(U.y, den.y)};
    const v2f x = splat(1.0f) - sd;
    const v2f bm = v2f{(U.x >= nLth.x) ? x.x : copysign_(0.0f, x.x), (U.y >= nLth.y) ? x.y : copysign_(0.0f, x.y)};
    return sel(thawed, splat(1.0f), bm);
}
// energy_closure (trm_device.hpp) on both components; CHECK as energy_closure_wave (trm_column.hpp): 0 none, 1 full, 2 the
// saturation comes out of this step's repair (in [0, 1] or NaN by construction: `sat == sat` is the whole check in the common
// path, where the liquid fraction is 1 or -0.0).  Returns the volumetric fractions it formed.
template <int CHECK = 1> TRM_DEV Frac2 energy_closure2(const DevParams<float>& p, v2f U, v2f sat, v2f& liq, v2f& T, uint32_t& viol) {
    const v2f Lth = sat * p.L * p.por;   // (p.L * sat) * por: multiplication by the scalar commutes bit for bit
    const v2f nLth = -Lth;
    // liq = (U >= 0) ? 1 : boolmul(U >= -Lth, 1 - safediv(U, -Lth))
    // (energy_closure_wave, trm_column.hpp: the phase-change divide only when a cell of the wave needs it; a frozen cell
    // with L_theta > eps gets the -0.0 the reference's `false * (1 - x)` gives)
    const float eps = Limits<float>::eps();
    const M2 thawed = ge(U, splat(0.0f)), frozen0 = lt(U, nLth);
    const bool needx = !thawed.x && !(frozen0.x && Lth.x > eps), needy = !thawed.y && !(frozen0.y && Lth.y > eps);
    bool okx, oky;
    if (wave_ballot(needx || needy) == 0ull) {
        liq = sel(thawed, splat(1.0f), splat(-0.0f));
        okx = (CHECK == 2 && TRM_CUT_CHECK) ? sat.x == sat.x : (0.0f <= sat.x && sat.x <= 1.0f);
        oky = (CHECK == 2 && TRM_CUT_CHECK) ? sat.y == sat.y : (0.0f <= sat.y && sat.y <= 1.0f);
        if (!TRM_CUT_CHECK) { okx = okx && (0.0f <= liq.x && liq.x <= 1.0f); oky = oky && (0.0f <= liq.y && liq.y <= 1.0f); }
    } else {
        const v2f den = nLth + eps;
        const v2f sd = v2f{(nLth.x == 0.0f) ? Limits<float>::inf() : div_nr(U.x, den.x), (nLth.y == 0.0f) ? Limits<float>::inf() : div_nr(U.y, den.y)};
        const v2f x = splat(1.0f) - sd;
        const v2f bm = v2f{(U.x >= nLth.x) ? x.x : copysign_(0.0f, x.x), (U.y >= nLth.y) ? x.y : copysign_(0.0f, x.y)};
        liq = sel(thawed, splat(1.0f), bm);
        okx = (0.0f <= sat.x && sat.x <= 1.0f) && (0.0f <= liq.x && liq.x <= 1.0f);
        oky = (0.0f <= sat.y && sat.y <= 1.0f) && (0.0f <= liq.y && liq.y <= 1.0f);
    }
    // (bit 1: component x out of bounds, bit 2: component y -- see fractions2)
    if (CHECK != 0) viol |= (okx ? 0u : 2u) | (oky ? 0u : 4u);
    const Frac2 f = fractions2_unchecked(p, sat, liq);
    const v2f C = heat_capacity2(p, f);
    const M2 frozen = lt(U, nLth);
    const v2f num = sel(frozen, U + Lth, U);
    const v2f quo = div2(num, C);
    T = sel(frozen || ge(U, splat(0.0f)), quo, splat(0.0f));
    return f;
}
// pow_int_m5 (trm_device.hpp) on both components
TRM_DEV v2f pow_int_m5_2(v2f x) {
    const v2f rx = div2(splat(1.0f), x);
    const v2f l0 = -fma2(x, rx, splat(-1.0f)) * rx;
    const v2f ynlo = splat(0.0f) + (l0 + splat(0.0f));
    v2f err = rx * 2.0f * l0;
    const v2f x2 = rx * rx;
    const v2f l2 = fma2(rx, rx, -x2) + err;
    err = x2 * 2.0f * l2;
    const v2f x4 = x2 * x2;
    const v2f l4 = fma2(x2, x2, -x4) + err;
    err = fma2(rx, l4, x4 * ynlo);
    const v2f a = fma2(x4, rx, err);
#if TRM_CUT_POWRARE
    // (an overflowing power is the rare case, decided per wave: see pow_int_m5)
    const unsigned long long all_finite = wave_ballot(is_finite(x4.x)) & wave_ballot(is_finite(err.x)) & wave_ballot(is_finite(x4.y)) & wave_ballot(is_finite(err.y));
    if (all_finite == wave_ballot(true)) return a;
    rare_path();
#endif
    const v2f b = x4 * rx;
    return v2f{(is_finite(x4.x) && is_finite(err.x)) ? a.x : b.x, (is_finite(x4.y) && is_finite(err.y)) ? a.y : b.y};
}
// pressure_head<float, HYD_BC_LINEAR> on both components (z0: per-component water table)
// (sat has been through the repair: never -0.0, see swrc_psi_bc's NSZ)
TRM_DEV v2f pressure_head2(const DevParams<float>& p, v2f sat, float z, float psiz, v2f z0) {
    const v2f theta = sat * p.por;
    const v2f r = div_const2_nsz(theta - p.theta_res, p.theta_span, p.rtheta_span);
    const v2f v = pow_int_m5_2(r) * (-p.bc_psi_s);
    const v2f psim = sel(lt(theta, splat(p.por)), v, splat(-p.bc_psi_s));
    const v2f psih = max2(splat(0.0f), z0 - z);
    return psih + psim + psiz;
}
// van Genuchten retention / Mualem conductivity (HYD_VG_N2): the square and cube roots, the reciprocal powers and the ice
// impedance are per-component instruction sequences whatever the layout, so the scalar functions are called on each
// component (identical bits by construction); everything around them -- composition, stencil, closures -- stays packed.
template <int HYD> TRM_DEV v2f conductivity_hydraulic2(const DevParams<float>& p, v2f liq, const Frac2& f) {
    if (HYD == HYD_BC_LINEAR) return conductivity_linear2(p, f);
    return v2f{conductivity_vg<float, false, true>(p, liq.x, Frac<float>{f.water.x, f.ice.x, f.air.x}),
               conductivity_vg<float, false, true>(p, liq.y, Frac<float>{f.water.y, f.ice.y, f.air.y})};
}
template <int HYD> TRM_DEV v2f pressure_head_hyd2(const DevParams<float>& p, v2f sat, float z, float psiz, v2f z0) {
    if (HYD == HYD_BC_LINEAR) return pressure_head2(p, sat, z, psiz, z0);
    return v2f{pressure_head<float, HYD_VG_N2, true>(p, sat.x, z, psiz, z0.x), pressure_head<float, HYD_VG_N2, true>(p, sat.y, z, psiz, z0.y)};
}
// upwind_conductivity on both components
TRM_DEV v2f upwind2(v2f g, v2f Kdn, v2f Kmid, v2f Kup) { return min2(Kmid, sel(lt(g, splat(0.0f)), Kdn, Kup)); }

}  // namespace pk

// grid: one wave per 2 * (64 / LPC) columns
// (Deriving T and liq from (U, sat) instead of reading them, as k_column can, was measured here as well: C5 562 vs 533 us
// per step -- the packed kernel is not short of bytes -- and is not offered.)
// DERIVE_LIQ: the incoming liquid fraction is re-derived from (U, sat) instead of being read (legal when the stored fields
// are the closure of the stored state, trm_ctx::closure_consistent): one of the five field reads less.
// the three granules of one column as the scalar path delivers them (fp32: one granule per value; see FrontArgs, trm_kernels.hpp)
struct FrontGranulesF {
    unsigned long long w[3];
    TRM_DEV void load(const unsigned long long* gran, unsigned byte_off_uniform) {
        for (int n = 0; n < 3; ++n) w[n] = sld_off<unsigned long long>(gran, byte_off_uniform + (unsigned)n * 8u);
    }
    TRM_DEV unsigned mismatch(unsigned epoch) const {      // (integers behind a barrier for the optimiser: stays on the scalar unit)
        unsigned bad = ((unsigned)(w[0] >> 32) ^ epoch) | ((unsigned)(w[1] >> 32) ^ epoch) | ((unsigned)(w[2] >> 32) ^ epoch);
        asm volatile("" : "+s"(bad));
        return bad;
    }
    TRM_DEV float value(int q) const { return __builtin_bit_cast(float, (unsigned)w[q]); }
};
// FRONT (LandModel, BCSIG_LAND): ground heat flux, infiltration and the new skin temperature come from the surface workgroups of THIS
// launch as granules (k_step_pk_land below; the fp64 form is column_program<FRONT> in trm_column.hpp)
template <bool RICHARDS, int LPC, int HYD, int DERIVE = DERIVE_NONE, int BCSIG = BCSIG_RUNTIME, bool FRONT = false>
TRM_DEV void step_pk_program(const View<float>& v_arg, const DevParams<float>& p_arg, float dt, int finalize, int write_kf, unsigned block, int staged = 0,
                             const FrontArgs* fa = nullptr) {
    static_assert(!FRONT || (BCSIG == BCSIG_LAND && RICHARDS), "the in-launch surface processes feed the LandModel step");
    constexpr unsigned off_p = round_up_to((unsigned)sizeof(View<float>), (unsigned)alignof(DevParams<float>));
    const View<float>& v = v_arg;
    const DevParams<float>& p = p_arg;
    using namespace pk;
    typedef float NF;
    constexpr int CPW = 64 / LPC;   // column PAIRS per wave
    TRM_PHASE("addressing");
    const int lane = threadIdx.x & 63;
    // (the wave index and what follows from it alone on the scalar unit; lane predicates from wave-uniform masks: lane_in)
#if TRM_CUT_MASKS
    const int wave = __builtin_amdgcn_readfirstlane((int)((block * (unsigned)TRM_STEP_BLOCK + threadIdx.x) >> 6));
    const int k = lane % LPC, sub = lane / LPC;
    const int Nz = v.Nz, Nh = (int)v.Nh;
    const bool upper = CPW == 2 && lane_in(0xffffffff00000000ull);
    const unsigned long long m_bot = level_lanes<LPC>(0), m_top = level_lanes<LPC>(Nz - 1);      // (handed to repair_saturation as they are)
    const bool is_bot = lane_in(m_bot), is_top = lane_in(m_top);
    const LevelGeom<NF> L = level_geom(v, k);
    const bool need_kc = RICHARDS || write_kf;

    const int pair_w0 = wave * CPW * 2;                  // (uniform) first column of the wave
    const int i0 = pair_w0 + sub * 2, i1 = i0 + 1;
    // columns pair_w0 + {0, 1} live in the lower half-wave (or the whole wave), + {2, 3} in the upper one
    const unsigned long long half_lo = CPW == 1 ? ~0ull : 0x00000000ffffffffull, half_hi = CPW == 1 ? 0ull : 0xffffffff00000000ull;
    const unsigned long long m_lev = levels_below<LPC>(Nz);
    const unsigned long long m_act0 = ((pair_w0 < Nh ? half_lo : 0ull) | (pair_w0 + 2 < Nh ? half_hi : 0ull)) & m_lev;
    const unsigned long long m_act1 = ((pair_w0 + 1 < Nh ? half_lo : 0ull) | (pair_w0 + 3 < Nh ? half_hi : 0ull)) & m_lev;
    const bool act0 = lane_in(m_act0), act1 = lane_in(m_act1);
    const int j0 = i0 < Nh ? i0 : Nh - 1, j1 = i1 < Nh ? i1 : Nh - 1;
#else      // (A/B builds: round 3's lane-wise form)
    const int wave = (int)((block * (unsigned)blockDim.x + threadIdx.x) >> 6);
    const int k = lane % LPC, sub = lane / LPC;
    const int Nz = v.Nz, Nh = (int)v.Nh;
    const bool upper = sub != 0;
    const bool is_bot = k == 0, is_top = k == Nz - 1;
    const unsigned long long m_bot = 0ull, m_top = 0ull;      // (repair_saturation takes the ballots itself)
    const LevelGeom<NF> L = level_geom(v, k);
    const bool need_kc = RICHARDS || write_kf;
    const int i0 = (wave * CPW + sub) * 2, i1 = i0 + 1;
    const bool act0 = i0 < Nh && k < Nz, act1 = i1 < Nh && k < Nz;
    const int j0 = i0 < Nh ? i0 : Nh - 1, j1 = i1 < Nh ? i1 : Nh - 1;
    const int pair_w0 = __builtin_amdgcn_readfirstlane(wave * CPW) * 2;
    const unsigned long long m_lev = wave_ballot(k < Nz), m_act0 = wave_ballot(i0 < Nh) & m_lev, m_act1 = wave_ballot(i1 < Nh) & m_lev;
#endif
    const unsigned kk = (unsigned)(k < Nz ? k : Nz - 1);
    const unsigned ib0 = (unsigned)j0 * 4u, ib1 = (unsigned)j1 * 4u;
    const unsigned cb0 = ((unsigned)j0 * (unsigned)v.Nzp + kk) * 4u, cb1 = ((unsigned)j1 * (unsigned)v.Nzp + kk) * 4u;
    uint32_t viol = 0;
    // The per-column inputs come through the scalar memory path (sld, trm_kernels.hpp): the four columns of the wave (two per
    // half-wave) from s_loads, selected per half-wave, instead of two vector loads per input in which every lane of a column reads
    // the same address (profiles/r03/exp27: C5 450.4 -> 447.0 us, a 12 696-column shard 15.3 -> 15.0).
    staged &= 1;
    // (byte offsets in scalar registers: the loads use them as they are, no 64-bit address per load -- sld_off)
    auto clampo = [&](int i) { return (unsigned)(i < Nh ? i : Nh - 1) * 4u; };
    const unsigned q0 = clampo(pair_w0), q1 = clampo(pair_w0 + 1), q2 = clampo(pair_w0 + 2), q3 = clampo(pair_w0 + 3);
    auto col_ld2 = [&](const float* ptr) -> v2f {
        const float a0 = sld_off<float>(ptr, q0), a1 = sld_off<float>(ptr, q1);
        if (CPW == 1) return v2f{a0, a1};
        const float a2 = sld_off<float>(ptr, q2), a3 = sld_off<float>(ptr, q3);
        return v2f{upper ? a2 : a0, upper ? a3 : a1};
    };

    TRM_PHASE("loads+derive");
#if TRM_LOAD_POINTERS_UPFRONT
    {   // the base pointers of every field read in ONE batch of scalar loads in front of the first: fetched where each is first used,
        // the later ones put a scalar load and its wait between the field loads.  (k_step_pk only: in k_column the same
        // statement made the scheduler put the level records first and wait for them in front of the field loads)
        const float* const pU = v.U; const float* const ps = v.sat; const float* const pT = v.T; const float* const pp = v.psi;
        asm volatile("" : : "s"(pU), "s"(ps), "s"(pT), "s"(pp));
    }
#endif
    const v2f U = ld2(v.U, cb0, cb1), sat = ld2(v.sat, cb0, cb1);
    v2f psi = splat(0.0f);
    // ---- the per-column inputs
    constexpr bool SIG = BCSIG >= 0;      // (the boundary kinds as compile-time constants: BCSIG, trm_kernels.hpp)
    const bool vTb = SIG ? (BCSIG & BCSIG_T_BOT) != 0 : v.bc.kind[2][0] == 1, vTt = SIG ? (BCSIG & BCSIG_T_TOP) != 0 : v.bc.kind[2][1] == 1;
    const bool seb = SIG ? (BCSIG & BCSIG_LAND) != 0 : p.seb != 0;
    // flux conditions: a term for the edge lane of every condition that is SET, nothing otherwise (column_program, trm_column.hpp)
    const bool fUb = SIG ? (BCSIG & BCSIG_FU_BOT) != 0 : v.bc.kind[0][0] == 2, fUt = seb || (SIG ? (BCSIG & BCSIG_FU_TOP) != 0 : v.bc.kind[0][1] == 2);
    const bool fSb = RICHARDS && (SIG ? (BCSIG & BCSIG_FS_BOT) != 0 : v.bc.kind[1][0] == 2), fSt = RICHARDS && (seb || (SIG ? (BCSIG & BCSIG_FS_TOP) != 0 : v.bc.kind[1][1] == 2));
    v2f in_Tb = splat(0.0f), in_Tt = splat(0.0f), in_Ub = splat(0.0f), in_Ut = splat(0.0f), in_Sb = splat(0.0f), in_St = splat(0.0f), in_wt = splat(0.0f);
    v2f S_in = splat(0.0f), Ts_in = splat(0.0f);
    FrontGranulesF fg[4] = {};
    bool front_ready = true;
    auto request_inputs = [&] {
        if (vTb) in_Tb = col_ld2(bcval(v, 2, 0));
        if (vTt) in_Tt = col_ld2(bcval(v, 2, 1));
        if (fUb) in_Ub = col_ld2(bcval(v, 0, 0));
        if (fUt && !FRONT) in_Ut = col_ld2(seb ? v.ghf : bcval(v, 0, 1));
        if (fSb) in_Sb = col_ld2(bcval(v, 1, 0));
        if (fSt && !FRONT) in_St = col_ld2(seb ? v.infil : bcval(v, 1, 1));
        if (RICHARDS && DERIVE == DERIVE_LIQ_PSI) in_wt = col_ld2(v.wt);
        // surface_excess_water and the skin temperature of the two columns: with the other inputs.  Vector memory retires in order,
        // loads and stores through the one counter: a load issued behind a store holds the whole wave until that store has been
        // acknowledged by memory (the top-lane block used to do that three times per wave).
        if (RICHARDS) S_in = col_ld2(v.S);
        if (seb && !FRONT) Ts_in = col_ld2(v.Ts);
        if constexpr (FRONT) {   // the granules of the wave's columns through the scalar path: valid if the surface workgroups have published them
            constexpr unsigned GB = 3u * 8u / 4u;      // granule bytes per column / bytes per float: q* are byte offsets of floats
            fg[0].load(fa->gran, q0 * GB); fg[1].load(fa->gran, q1 * GB);
            if (CPW == 2) { fg[2].load(fa->gran, q2 * GB); fg[3].load(fa->gran, q3 * GB); }
            const unsigned e = fa->epoch;
            front_ready = (fg[0].mismatch(e) | fg[1].mismatch(e) | (CPW == 1 ? 0u : (fg[2].mismatch(e) | fg[3].mismatch(e)))) == 0u;
            auto pick = [&](int q) { return CPW == 1 ? v2f{fg[0].value(q), fg[1].value(q)} : v2f{upper ? fg[2].value(q) : fg[0].value(q), upper ? fg[3].value(q) : fg[1].value(q)}; };
            in_Ut = pick(0); in_St = pick(1); Ts_in = pick(2);
        }
    };
    // (they travel through the scalar path, whose loads return out of order: requested in front of the derivation they hold up its
    // parameter reloads -- C5 +4 %, profiles/r04/exp8 -- so the request stays behind it; k_column's vector path requests early)
    constexpr bool EARLY = false;
    if (EARLY) request_inputs();
    if (RICHARDS && DERIVE == DERIVE_LIQ_PSI) {
        // saturation_to_pressure! of the stored state, from the stored saturation and the stored water table (what the step that
        // wrote the field evaluated: same function, same operands, same bits) instead of a fourth field read
        if (!EARLY) in_wt = col_ld2(v.wt);
        psi = pressure_head_hyd2<HYD>(kernarg_reload<DevParams<float>>(off_p), sat, L.zC, L.psiz, in_wt);
    } else if (RICHARDS) {
        psi = ld2(v.psi, cb0, cb1);
    }
    v2f T, liq;
    if (DERIVE == DERIVE_T_LIQ) {      // both re-derived from (U, sat): two field reads less (k_column: DERIVE_T_LIQ)
        uint32_t viol_in = 0;
        (void)energy_closure2<0>(kernarg_reload<DevParams<float>>(off_p), U, sat, liq, T, viol_in);
    } else {
        T = ld2(v.T, cb0, cb1);
        liq = (DERIVE == DERIVE_LIQ || DERIVE == DERIVE_LIQ_PSI) ? liquid_fraction2(kernarg_reload<DevParams<float>>(off_p), U, sat) : ld2(v.liq, cb0, cb1);
    }

    TRM_PHASE_FENCE("cell properties", T, liq, psi);
    // (bounds of the incoming state were flagged by the launch that produced it)
    const Frac2 f = fractions2_unchecked(p, sat, liq);
    const v2f kap = conductivity2(p, f);
    const v2f Kc = need_kc ? conductivity_hydraulic2<HYD>(p, liq, f) : splat(0.0f);

    v2f T_sh = up2(T), kap_sh = up2(kap);
    v2f flux_U = splat(0.0f), flux_S = splat(0.0f);
    TRM_PHASE_FENCE("inputs", T_sh, kap_sh);
    if (!EARLY) request_inputs();
    // ---- boundary conditions: one wave-uniform branch per condition that is not set (k_step_wave, GENERIC_BC = false)
    v2f T_ext_b = T, T_ext_t = T;
    if (vTb) T_ext_b = T + div_const2_nsz(T - in_Tb, v.g.hdzf_bot, v.g.rhdzf_bot) * (-v.g.dzf_bot);      // (nsz: see column_tendencies)
    if (vTt) T_ext_t = T + div_const2_nsz(in_Tt - T, v.g.hdzf_top, v.g.rhdzf_top) * v.g.dzf_top;
    v2f T_m = sel(is_bot, T_ext_b, T_sh);
    v2f T_h = T_ext_t;
    v2f kap_halo = kap;
    if (!RICHARDS && p.halo_policy != 1) kap_halo = conductivity2(p, fractions2(p, splat(0.0f), liq, viol));
    v2f kap_m = sel(is_bot, kap_halo, kap_sh);
    v2f kap_h = kap_halo;
#if TRM_CUT_FLUX
    if (fUb) flux_U = sel(is_bot, div_const2_nsz(in_Ub * v.g.Az, v.g.V_bot, v.g.rV_bot), flux_U);
    if (fUt) flux_U = sel(is_top, -div_const2_nsz(in_Ut * v.g.Az, v.g.V_top, v.g.rV_top), flux_U);
    if (fSb) flux_S = sel(is_bot, div_const2_nsz(in_Sb * v.g.Az, v.g.V_bot, v.g.rV_bot), flux_S);
    if (fSt) flux_S = sel(is_top, -div_const2_nsz((seb ? -in_St : in_St) * v.g.Az, v.g.V_top, v.g.rV_top), flux_S);
#else
    {   // both edge terms always formed, two selects per variable (the form measured faster: no select inside the branches)
        v2f eU_b = splat(0.0f), eU_t = splat(0.0f), eS_b = splat(0.0f), eS_t = splat(0.0f);
        if (fUb) eU_b = div_const2_nsz(in_Ub * v.g.Az, v.g.V_bot, v.g.rV_bot);
        if (fUt) eU_t = -div_const2_nsz(in_Ut * v.g.Az, v.g.V_top, v.g.rV_top);
        flux_U = sel(is_bot, eU_b, sel(is_top, eU_t, splat(0.0f)));
        if (RICHARDS) {
            if (fSb) eS_b = div_const2_nsz(in_Sb * v.g.Az, v.g.V_bot, v.g.rV_bot);
            if (fSt) eS_t = -div_const2_nsz((seb ? -in_St : in_St) * v.g.Az, v.g.V_top, v.g.rV_top);
            flux_S = sel(is_bot, eS_b, sel(is_top, eS_t, splat(0.0f)));
        }
    }
#endif
    TRM_PHASE_FENCE("tendencies", flux_U, flux_S, T_m, T_h, kap_m, kap_h, psi);
    // ---- heat
    const v2f qT_lo = -((kap + kap_m) * 0.5f) * ((T - T_m) * L.rdzf_lo);
    const v2f qT_sh = dn2(qT_lo);
    const v2f qT_top = -((kap_h + kap) * 0.5f) * ((T_h - T) * L.rdzf_hi);
    const v2f qT_hi = sel(is_top, qT_top, qT_sh);
    v2f gU = splat(0.0f) + (-((qT_hi - qT_lo) * L.rdzc));
    // ---- Richards
    v2f gS = splat(0.0f), Kf_lo = splat(0.0f);
    if (need_kc) {
        const v2f Kmin = min2(Kc, up2(Kc));
        Kf_lo = sel(is_bot || is_top, Kc, Kmin);
    }
    if (RICHARDS) {
        const v2f Kf_up = up2(Kf_lo), Kf_dn = dn2(Kf_lo), psi_sh = up2(psi);
        const v2f Kf_m = Kf_up;      // (the bottom lane's gradient is never negative: see column_tendencies)
        const v2f Kf_p = sel(is_top, Kc, Kf_dn);
        const v2f psi_m = sel(is_bot, psi, psi_sh);
        const v2f g_lo = (psi - psi_m) * L.rdzf_lo;
        const v2f Ks_lo = upwind2(g_lo, Kf_m, Kf_lo, Kf_p);
        const v2f qW_lo = -Ks_lo * g_lo;
        const v2f qW_sh = dn2(qW_lo);
        const v2f zero_or_nan = psi - psi;
        const v2f qW_t = -min2(Kc, splat(0.0f)) * zero_or_nan;
        const v2f qW_hi = sel(is_top, qW_t, qW_sh);
        const v2f dtheta = -((qW_hi - qW_lo) * L.rdzc) + splat(0.0f) + p.vwc_forcing;
        gS = splat(0.0f) + div_const2_nsz(dtheta, p.por, p.rpor);
    }
    bool front_timeout = false;
    if constexpr (FRONT) if (!front_ready) {
        // the scalar read came before the surface workgroups had published: poll the granules (vector loads at agent scope; lanes
        // 0 .. 2 of a half-wave read the granules of its first column, 3 .. 5 of its second), here, where the values are first needed
        TRM_PHASE("rare+ granule poll");
        const unsigned long long* gp = fa->gran + (size_t)(k < 3 ? j0 : j1) * 3 + (k < 6 ? k % 3 : 2);
        unsigned long long g = 0;
        bool ok = false;
        for (int spin = 0; spin < FRONT_SPIN_LIMIT && !ok; ++spin) {
            g = ld_agent(gp);
            ok = wave_ballot((unsigned)(g >> 32) != fa->epoch) == 0ull;
            if (!ok) __builtin_amdgcn_s_sleep(4);
        }
        const int w = (int)(unsigned)g;
        auto value = [&](int q, int col, int base) { return __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_readlane(w, base + col * 3 + q)); };
        auto pick = [&](int q) {
            const v2f lo = v2f{value(q, 0, 0), value(q, 1, 0)};
            if (CPW == 1) return lo;
            const v2f hi = v2f{value(q, 0, LPC), value(q, 1, LPC)};
            return v2f{upper ? hi.x : lo.x, upper ? hi.y : lo.y};
        };
        v2f ut = pick(0), st = pick(1), ts = pick(2);
        if (!ok) {   // gave up: no hang, the columns of the wave are flagged and NaN
            front_timeout = true;
            ut = st = ts = splat(__builtin_nanf(""));
        }
        Ts_in = ts;
        flux_U = sel(is_top, -div_const2_nsz(ut * v.g.Az, v.g.V_top, v.g.rV_top), splat(0.0f));
        flux_S = sel(is_top, -div_const2_nsz((-st) * v.g.Az, v.g.V_top, v.g.rV_top), splat(0.0f));
        TRM_PHASE("rare-");
    }
#if TRM_CUT_FLUX
    if (fUb || fUt) gU += flux_U;
    if (fSb || fSt) gS += flux_S;
#else
    gU += flux_U;
    if (RICHARDS) gS += flux_S;
#endif
    TRM_PHASE_FENCE("advance", gU, gS, Kf_lo);
    // ---- explicit Euler update
    const v2f Unew = U + gU * dt;
    bool bad = (act0 && is_nan(Unew.x)) || (act1 && is_nan(Unew.y));
    v2f snew = sat, z0 = splat(0.0f), GS_top = splat(0.0f), S_out = splat(0.0f);
    if (RICHARDS) {
        snew = sat + gS * dt;
        bad = bad || (act0 && is_nan(snew.x)) || (act1 && is_nan(snew.y));
        float sx = snew.x, sy = snew.y;
        const float over0 = repair_saturation<NF, LPC>(v, sx, k, Nz, m_act0, is_bot, is_top, L, m_bot, m_top);
        const float over1 = repair_saturation<NF, LPC>(v, sy, k, Nz, m_act1, is_bot, is_top, L, m_bot, m_top);
        snew = v2f{sx, sy};
        z0 = v2f{water_table<NF, LPC>(sx, m_act0, lane, L), water_table<NF, LPC>(sy, m_act1, lane, L)};
        // surface_excess_water: tendency min(0, S) once per column, Euler update, overflow (stored with everything else below)
        GS_top = v2f{0.0f + jl_min(0.0f, S_in.x), 0.0f + jl_min(0.0f, S_in.y)};
        S_out = v2f{(S_in.x + GS_top.x * dt) + over0, (S_in.y + GS_top.y * dt) + over1};
    }
    TRM_PHASE_FENCE("closure", snew, z0, S_out, GS_top);
    // ---- closures
    v2f ln, Tn;
    const DevParams<float>& p2 = kernarg_reload<DevParams<float>>(off_p);   // (second half of the step: see kernarg_reload)
    const Frac2 f_new = energy_closure2<RICHARDS ? 2 : 1>(p2, Unew, snew, ln, Tn, viol);
    const v2f psin = RICHARDS ? pressure_head_hyd2<HYD>(p2, snew, L.zC, L.psiz, z0) : splat(0.0f);
    TRM_PHASE_FENCE("outputs", ln, Tn);
    v2f Kf_out = Kf_lo, Kf_out_top = Kc;
    if (finalize && write_kf) {
        const v2f Kc_new = conductivity_hydraulic2<HYD>(p, ln, f_new);      // (the closure has checked this composition)
        const v2f Kmin_new = min2(Kc_new, up2(Kc_new));
        Kf_out = sel(is_bot || is_top, Kc_new, Kmin_new);
        Kf_out_top = Kc_new;
    }
    constexpr int cpb = (TRM_STEP_BLOCK / 64) * CPW * 2;      // columns per workgroup
    auto store = [&](bool act, int cib, unsigned cb_, unsigned ib_, float u, float t, float l, float s, float ps, float kf, float kft, float gu, float gs,
                     float S_new, float GS, float wt, float Ts_old /* already stepped */) {
        if (!act) return;
        // (block_local in EVERY block that stores: instruction selection works one basic block at a time, and an offset defined in
        // another block has been widened to 64 bits there -- each store then pays a 64-bit address add and loses the saddr form)
        if (is_top && staged) {   // the column's 0-D outputs through the workgroup's staging table (store_small_outputs, trm_column.hpp)
            float* st = small_stage<float>();
            if (write_kf) st[SMALL_KF_TOP * cpb + cib] = kft;
            if (RICHARDS) {
                if (finalize) st[SMALL_G_S * cpb + cib] = GS;
                st[SMALL_S * cpb + cib] = S_new;
                st[SMALL_WT * cpb + cib] = wt;
            }
            if (seb) {
                st[SMALL_TS * cpb + cib] = Ts_old;
                st[SMALL_TOP_T * cpb + cib] = t;
                st[SMALL_TOP_SAT * cpb + cib] = s;
                st[SMALL_TOP_LIQ * cpb + cib] = l;
            }
        } else if (is_top) {   // the column's 0-D state (top lane)
            if (RICHARDS) {
                if (finalize) stg(v.G_S, block_local(ib_), GS);
                const unsigned ib = block_local(ib_);
                stg(v.S, ib, S_new);
                stg(v.wt, ib, wt);
            }
            if (seb) stg(v.Ts, block_local(ib_), Ts_old);
        }
        if (finalize) {   // state.tendencies of the last step (k_step_wave)
            const unsigned cb = block_local(cb_);
            stg(v.G_U, cb, gu);
            if (RICHARDS) stg(v.G_sat, cb, gs);
        }
        {
            const unsigned cb = block_local(cb_);
            stg(v.U, cb, u);
            stg(v.T, cb, t);
            stg(v.liq, cb, l);
            if (RICHARDS) { stg(v.sat, cb, s); stg(v.psi, cb, ps); }
        }
        if (is_top && seb && !staged) { const unsigned ib = block_local(ib_); stg(v.top_T, ib, t); stg(v.top_sat, ib, s); stg(v.top_liq, ib, l); }
        if (write_kf) {
            stg(v.Kf, block_local(cb_), kf);
            if (is_top && !staged) stg(v.Kf_top, block_local(ib_), kft);
        }
    };
    // every loaded value has been consumed before the first store is issued (nothing is waited for behind the stores)
    v2f Ts_new = splat(0.0f);
    if (seb) Ts_new = Ts_in + splat(0.0f) * dt;     // zero-tendency prognostic skin_temperature
    asm volatile("" : "+v"(Ts_new), "+v"(S_out), "+v"(GS_top));
    TRM_PHASE("stores");
    const int cib0 = ((int)(threadIdx.x >> 6) * CPW + sub) * 2;
    store(act0, cib0, cb0, ib0, Unew.x, Tn.x, ln.x, snew.x, psin.x, Kf_out.x, Kf_out_top.x, gU.x, gS.x, S_out.x, GS_top.x, z0.x, Ts_new.x);
    store(act1, cib0 + 1, cb1, ib1, Unew.y, Tn.y, ln.y, snew.y, psin.y, Kf_out.y, Kf_out_top.y, gU.y, gS.y, S_out.y, GS_top.y, z0.y, Ts_new.y);
    if (staged) {
        const unsigned enabled = (write_kf ? 1u << SMALL_KF_TOP : 0u) | (RICHARDS ? (1u << SMALL_S) | (1u << SMALL_WT) : 0u) |
                                 ((RICHARDS && finalize) ? 1u << SMALL_G_S : 0u) |
                                 (seb ? (1u << SMALL_TOP_T) | (1u << SMALL_TOP_SAT) | (1u << SMALL_TOP_LIQ) | (1u << SMALL_TS) : 0u);
        store_small_outputs<float, cpb>(enabled, block, Nh);
    }
    // (a flag raised by the clamped copy of the last column in an odd-sized shard repeats that column's own flag)
    const uint32_t flags = (bad ? 1u : 0u) | ((((viol & 2u) && act0) || ((viol & 4u) && act1)) ? 2u : 0u) | (front_timeout ? 4u : 0u);
    if (flags) atomicOr(v.status, flags);
}
template <bool RICHARDS, int LPC, int HYD, int DERIVE = DERIVE_NONE, int BCSIG = BCSIG_RUNTIME>
__global__ void __launch_bounds__(TRM_STEP_BLOCK) k_step_pk(View<float> v_arg, DevParams<float> p_arg, float dt, int finalize, int write_kf, int staged) {
    step_pk_program<RICHARDS, LPC, HYD, DERIVE, BCSIG>(v_arg, p_arg, dt, finalize, write_kf, xcd_block<TRM_XCD_REMAP_PK != 0>(blockIdx.x, gridDim.x), staged);
}
// LandModel in fp32, ONE launch per step (TRM_OPT_SURFACE_IN_LAUNCH): the first workgroups evaluate the 0-D surface processes of 256
// columns each (surface_front), the others run the packed step, which receives ground heat flux, infiltration and the new skin
// temperature from them through granules (k_column_land, trm_column.hpp, in fp64)
template <int LPC, int HYD, int DERIVE>
__global__ void __launch_bounds__(TRM_STEP_BLOCK) k_step_pk_land(View<float> v_arg, DevParams<float> p_arg, float dt, int finalize, int write_kf, int staged, FrontArgs fa) {
    // (as k_column_land: the branch from the head of the View, the surface workgroups' arguments fetched inside their branch -- read as
    //  `v_arg.x` their scalar loads were hoisted in front of it and parked in vector lanes by every COLUMN wave, 20 v_writelane each)
    const int chain_blocks = (int)((v_arg.Nh + (TRM_STEP_BLOCK - 1)) / TRM_STEP_BLOCK);
    if ((int)blockIdx.x < chain_blocks) {
        __builtin_amdgcn_s_setprio(3);
        const long i = (long)blockIdx.x * TRM_STEP_BLOCK + threadIdx.x;
        constexpr unsigned off_p = round_up_to((unsigned)sizeof(View<float>), (unsigned)alignof(DevParams<float>));
        constexpr unsigned off_fa = round_up_to(off_p + (unsigned)sizeof(DevParams<float>) + 4u * (unsigned)sizeof(int), (unsigned)alignof(FrontArgs));
        const View<float>& v = kernarg_reload<View<float>>(0);
        if (i - (long)(threadIdx.x & 63u) < v.Nh) surface_front<float, true, HYD>(v, kernarg_reload<DevParams<float>>(off_p), kernarg_reload<FrontArgs>(off_fa), i);      // (wave-uniform)
        return;
    }
    step_pk_program<true, LPC, HYD, DERIVE, BCSIG_LAND, true>(v_arg, p_arg, dt, finalize, write_kf, blockIdx.x - (unsigned)chain_blocks, staged, &fa);
}
// LandModel in fp32: the packed column step of one half of the columns beside the surface processes of the other half in one
// launch (k_land_euler, trm_column.hpp).  (View, DevParams) first: step_pk_program re-reads them from the kernarg segment.
template <bool RICHARDS, int LPC, int HYD, bool TOP_ARRAYS>
__global__ void __launch_bounds__(TRM_STEP_BLOCK) k_land_pk(View<float> v_col, DevParams<float> p_arg, float dt, int finalize, int write_kf, View<float> v_surf, int surface_blocks) {
    if ((int)blockIdx.x < surface_blocks) {
        const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
        if (i < v_surf.Nh) surface_program<float, RICHARDS, HYD, true, TOP_ARRAYS>(v_surf, p_arg, i);
        return;
    }
    step_pk_program<RICHARDS, LPC, HYD, DERIVE_NONE>(v_col, p_arg, dt, finalize, write_kf, blockIdx.x - (unsigned)surface_blocks);
}

}  // namespace trm
