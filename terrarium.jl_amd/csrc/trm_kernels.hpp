// trm_kernels.hpp -- HIP kernels of the SoilModel / LandModel(bare ground) step.
//
// Data layout (HBM): one plain buffer per variable (SoA), stored COLUMN-MAJOR IN z:
// element (column i, level k) lives at i * Nzp + k, k = 0 the bottom layer, Nzp the level
// pitch (32 for Nz <= 32, 64 for Nz <= 64, otherwise Nz rounded up to 32).  A soil column is
// therefore one contiguous, 256/512-byte aligned segment, and the mapping "lane = soil level"
// reads and writes it with a single fully coalesced access per variable: two columns per
// wavefront at Nz <= 32, one at Nz <= 64.  No halos in memory: halo values are pure functions
// of the edge cell and the boundary condition and are formed in registers.  The C ABI
// (trm_upload / trm_download) converts from / to the reference's x-fastest interior layout.
//
// Why lane = level: the global N145 grid has only 56 951 columns.  One lane per column gives
// 890 wavefronts for 1024 SIMDs, and the step is bound by instruction issue (about 300 fp64
// instructions and 4 divides per cell), not by HBM: measured 64 us per step with a rolling
// register stencil.  Spreading the vertical axis over the lanes gives 28 000+ independent
// waves, the vertical stencil becomes DPP wave shifts, the water table a ballot, and the
// sequential saturation repair a ballot-guarded, range-limited lane-serial loop: 30 us.  (An
// intermediate version kept the x-fastest layout and transposed 32-column tiles through LDS:
// 57-82 us, limited by the load -> barrier -> compute -> barrier -> store phases.  See DESIGN.md.)
//
// Implementations behind the same C ABI:
//   * k_step_wave -- ONE launch per time step (update_state! + explicit_step! + closure!), or two
//     for Heun (predictor into the stage buffers, corrector from the stage's tendencies);
//   * k_step_pk (trm_packed_f32.hpp) -- the same step in fp32 with two columns per lane and packed math;
//   * k_* (unfused) -- one launch per reference kernel in the reference's order:
//     the stand-alone compute_* / closure entry points, Nz > 64, and the A/B comparator for what
//     fusion buys;
//   * k_surface (LandModel's 0-D surface energy balance), k_interp_series (time series inputs).
#pragma once
#include "trm_device.hpp"

namespace trm {

template <class NF> struct View {
    // (member order = order of use in the fused step kernel, see DevParams)
    long Nh;
    int Nz, Nzp;
    // per-level grid records for the lane = level kernels: 8 words {zC, psiz, zF, dzc, rdzc, rdzf[k], rdzf[k+1], 0}
    const NF* lvl;
    // 3-D, [Nh][Nzp]
    NF *U, *sat, *T, *liq, *psi;
    BcSet bc;
    BcGeom<NF> g;
    NF *S, *wt;
    NF *Kf;
    NF *Kf_top;  // hydraulic conductivity of the top face (face Nz), [Nh]
    uint32_t* status;
    NF *G_U, *G_sat, *G_S;
    const NF* Fvwc;   // per-cell vwc_forcing [Nh][Nzp], or null: the scalar DevParams::vwc_forcing applies
    // 2-D, [Nh]
    NF *Ts, *ghf, *infil, *swu, *lwu, *rnet, *Hs, *Hl, *evap, *runoff;
    // LandModel: (T, sat, liq) of the top cell, [Nh] each, written by the fused step for the next k_surface launch
    // (a coalesced read instead of one cache line per column); null without the surface energy balance
    NF *top_T, *top_sat, *top_liq;
    const NF *Tair, *pres, *wind, *qair, *rain, *swd, *lwd;
    const NF *albedo, *emissivity;   // PrescribedAlbedo inputs, [Nh]
    // grid (device arrays): zC[Nz], zF[Nz+1], dzc[Nz], rdzc[Nz], rdzf[Nz+1] (face f lies below cell f),
    // psiz[Nz] = zC - z_surface (elevation head)
    const NF *zC, *zF, *dzc, *rdzc, *rdzf, *psiz;
    // the per-column outputs of the column program in the order of its staged store (column_program, SMALL_*): Kf_top, S, wt,
    // G_S, top_T, top_sat, top_liq, Ts -- the same pointers as above, as a table a lane can index
    NF* small[8];
};
// The boundary-condition SIGNATURE of a launch as a compile-time constant (template parameter BCSIG of the per-step column programs;
// -1 = read the kinds at run time).  The kinds are wave-uniform run-time values, each one a scalar compare and a branch around a
// conditional load in front of the column's arithmetic; compiled in they cost the Euler program 3 ... 5 % of its time
// (profiles/r04/exp6_diag_bc_signature.log).  The launchers instantiate the signatures of the reference's examples and models.
enum { BCSIG_RUNTIME = -1,
       BCSIG_T_BOT = 1, BCSIG_T_TOP = 2,      // Value on temperature, bottom / top
       BCSIG_FU_BOT = 4, BCSIG_FS_BOT = 8,    // Flux on internal energy / saturation, bottom
       BCSIG_FU_TOP = 16, BCSIG_FS_TOP = 32,  // Flux on internal energy / saturation, top
       BCSIG_LAND = 64 };                     // LandModel: ground heat flux and infiltration wired to the top (land_model.jl:56-61)
// which closure fields of the incoming state a fused step re-derives from (U, sat) instead of reading (TRM_OPT_DERIVE_CLOSURE_FIELDS)
enum { DERIVE_NONE = 0, DERIVE_T_LIQ = 1, DERIVE_LIQ = 2,
       DERIVE_LIQ_PSI = 3,     // (packed fp32 step only) the liquid fraction AND the pressure head: psi = psi(sat, water table), two reads less
       DERIVE_ALL = 4 };       // (fp64 column program, Richards) T, liq AND psi: the step reads U and sat alone
enum { SMALL_KF_TOP = 0, SMALL_S, SMALL_WT, SMALL_G_S, SMALL_TOP_T, SMALL_TOP_SAT, SMALL_TOP_LIQ, SMALL_TS, SMALL_COUNT };
template <class NF> inline void fill_small_table(View<NF>& v) {
    v.small[SMALL_KF_TOP] = v.Kf_top; v.small[SMALL_S] = v.S; v.small[SMALL_WT] = v.wt; v.small[SMALL_G_S] = v.G_S;
    v.small[SMALL_TOP_T] = v.top_T; v.small[SMALL_TOP_SAT] = v.top_sat; v.small[SMALL_TOP_LIQ] = v.top_liq; v.small[SMALL_TS] = v.Ts;
}

template <class NF> TRM_DEV const NF* bcval(const View<NF>& v, int var, int side) { return (const NF*)v.bc.value[var][side]; }

// halo of the saturation field: prognostic under Richards (BC-driven, default = edge),
// never filled under NoFlow unless the MIRROR policy is selected.
template <class NF, bool RICHARDS> TRM_DEV NF sat_halo(const View<NF>& v, const DevParams<NF>& p, int side, long i, NF edge) {
    if (RICHARDS) {
        int kind = v.bc.kind[1][side];
        return side ? halo_top(kind, bcval(v, 1, 1), i, edge, v.g) : halo_bottom(kind, bcval(v, 1, 0), i, edge, v.g);
    }
    return p.halo_policy == 1 ? edge : NF(0);
}

// cell index of the one-thread-per-cell kernels (level fastest => coalesced)
#define TRM_CELL_INDEX(v)                                                  \
    const long gid__ = (long)blockIdx.x * blockDim.x + threadIdx.x;       \
    const long i = gid__ / (v).Nzp;                                        \
    const int k = (int)(gid__ % (v).Nzp);                                  \
    if (i >= (v).Nh || k >= (v).Nz) return;                                \
    const long c = gid__;

// ===========================================================================
// Unfused kernels: one per reference kernel
// ===========================================================================

// compute_hydraulics_kernel! (soil_hydrology.jl:145-163, 297-300)
template <class NF, int HYD> __global__ void k_hydraulics(View<NF> v, DevParams<NF> p) {
    TRM_CELL_INDEX(v);
    uint32_t viol = 0;
    const int Nz = v.Nz;
    auto Kc = [&](long cc) {
        NF s = v.sat[cc], l = v.liq[cc];
        return conductivity_hydraulic<NF, HYD>(p, l, fractions(p, s, l, viol));
    };
    if (k <= 0) {
        v.Kf[c] = Kc(c);
    } else if (k >= Nz - 1) {
        NF val = Kc(c);
        v.Kf[c] = val;
        v.Kf_top[i] = val;
    } else {
        v.Kf[c] = jl_min(Kc(c), Kc(c - 1));
    }
    if (viol) atomicOr(v.status, viol);
}

// bare-ground evaporation + direct runoff + fused SEB kernel x2 (land_model.jl:79-88), one thread
// per column.  FROM_STATE: take the top-face hydraulic conductivity from (sat, liq) of the top cell
// (what compute_hydraulics! would store there, soil_hydrology.jl:156-158) instead of reading the
// hydraulic_conductivity field -- used in front of the fused step kernel, which does not
// materialise K before the surface processes run.
// TOP_ARRAYS (with FROM_STATE): the top cell's (T, sat, liq) come from the compact per-column arrays the fused step wrote.
template <class NF> TRM_DEV NF ldg(const NF* base, unsigned byte_off);          // (defined with the fused step's helpers below)
template <class NF> TRM_DEV void stg(NF* base, unsigned byte_off, NF x);
TRM_DEV unsigned block_local(unsigned byte_off);
template <class NF, bool RICHARDS, int HYD, bool FROM_STATE, bool TOP_ARRAYS> TRM_DEV void surface_program(const View<NF>& v, const DevParams<NF>& p, long i) {
    const long top = i * v.Nzp + (v.Nz - 1);
    // (scalar base + 32-bit byte offset for every per-column access: ldg / stg -- 22 accesses without a 64-bit address each)
    const unsigned ib = (unsigned)i * (unsigned)sizeof(NF);
    SebIn<NF> in = {ldg(v.Tair, ib), ldg(v.pres, ib), ldg(v.wind, ib), ldg(v.qair, ib), ldg(v.rain, ib), ldg(v.swd, ib), ldg(v.lwd, ib), NF(0), NF(0), NF(0)};
    seb_radiation_inputs(p, v.albedo, v.emissivity, ib, in);
    SebOut<NF> o;
    uint32_t viol = 0;
    const unsigned ib2 = block_local(ib);      // (behind seb_radiation_inputs' branch: see block_local)
    const NF T_top = TOP_ARRAYS ? ldg(v.top_T, ib2) : v.T[top], sat_top = TOP_ARRAYS ? ldg(v.top_sat, ib2) : v.sat[top];
    NF Kf_top;
    const NF liq_top = (FROM_STATE && TOP_ARRAYS) ? ldg(v.top_liq, ib2) : v.liq[top];
    if (FROM_STATE) {
        Kf_top = conductivity_hydraulic<NF, HYD, false>(p, liq_top, fractions(p, sat_top, liq_top, viol));   // (in front of the fused step)
    } else {
        Kf_top = v.Kf[top];
    }
    surface_processes(p, in, ldg(v.Ts, ib2), T_top, sat_top, liq_top, Kf_top, ldg(v.S, ib2), RICHARDS, v.dzc[v.Nz - 1], o);
    const unsigned ob = block_local(ib);
    stg(v.Ts, ob, o.Ts); stg(v.ghf, ob, o.ghf); stg(v.swu, ob, o.swu); stg(v.lwu, ob, o.lwu); stg(v.rnet, ob, o.rnet);
    stg(v.Hs, ob, o.Hs); stg(v.Hl, ob, o.Hl); stg(v.evap, ob, o.evap); stg(v.infil, ob, o.infil); stg(v.runoff, ob, o.runoff);
}
// ---- LandModel: the surface processes INSIDE the step launch (TRM_OPT_SURFACE_IN_LAUNCH, k_column_land in trm_column.hpp) -------
// k_surface in front of every column launch is a latency-bound chain of ~450 dependent fp64 instructions on 890 waves (N145): 5.4 us
// of a 30 us step, ~4 of them the fixed cost of a launch between two others.  Here the first workgroups of the column launch
// evaluate the chain -- one lane per column, as k_surface does -- and hand its three results the column program needs (ground heat
// flux, infiltration, the new skin temperature) to the column workgroups of the SAME launch, which need them only at the explicit
// step, half-way through their arithmetic.  Hand-off after MI355X_MICROARCH.md (inter-workgroup visibility, recipe R2: the data is
// the flag): every 32-bit half of a value travels in an 8-byte granule {epoch, half} written by ONE agent-scope (write-through)
// store and read by agent-scope loads; a granule whose tag is the launch's epoch is valid, whatever the caches did.  The epoch
// counts the context's launches of this kind (never 0: the granules start zeroed), so nothing is cleared between launches.
// The column program reads its granules first through the scalar path (a stale line there shows an old tag, never a wrong value)
// and polls them with vector loads only if that read came too early; the poll is bounded -- a wave that gives up raises
// TRM_STATUS_HANDOFF_TIMEOUT and NaN in its columns instead of hanging the device (the surface workgroups are the first of the
// grid and wait for nothing, so this needs a dispatcher that starts later workgroups first AND fills the device with them).
#ifndef TRM_STEP_BLOCK
#define TRM_STEP_BLOCK 256
#endif
struct FrontArgs {
    unsigned long long* gran;   // fp64 [Nh][6]: {ghf.lo, ghf.hi, infil.lo, infil.hi, Ts.lo, Ts.hi}, fp32 [Nh][3]: {ghf, infil, Ts}; each (epoch << 32) | 32 bits
    unsigned epoch;
    int chain_blocks;           // the first workgroups of the grid evaluate the surface processes, 256 columns each
    unsigned tag_bias;          // 0; TRM_DEBUG_HANDOFF_TAG_BIAS=1 (tests): the surface workgroups publish under ANOTHER tag, i.e. never for the
                                // column waves of their launch -- whose bounded wait must then end with TRM_STATUS_HANDOFF_TIMEOUT, not hang
};
enum { FRONT_GHF = 0, FRONT_INFIL = 2, FRONT_TS = 4, FRONT_GRANULES = 6, FRONT_SPIN_LIMIT = 1 << 14 };
// granules per column: one per 32-bit half of the three values -- 6 in fp64 (offsets FRONT_GHF / FRONT_INFIL / FRONT_TS), 3 in fp32
// (value q at granule q)
template <class NF> constexpr int front_granules() { return 3 * (int)(sizeof(NF) / 4); }
TRM_DEV void st_agent(unsigned long long* p, unsigned long long x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
TRM_DEV unsigned long long ld_agent(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// compute_auxiliary! of the surface processes of column i (land_model.jl:79-88): surface_program<FROM_STATE, TOP_ARRAYS> except
// for where the results go -- the skin temperature is NOT stored (the column program stores skin_temperature + 0 * dt, as
// explicit_step! leaves it, from the value it is handed here).
// Called by EVERY lane of a surface wave whose first column exists (the lanes beyond the last column repeat it and store nothing of
// their own: the transposed granule store needs the whole wave).
template <class NF, bool RICHARDS, int HYD> TRM_DEV void surface_front(const View<NF>& v, const DevParams<NF>& p, const FrontArgs& fa, long i_lane) {
    constexpr int G = front_granules<NF>();
    const bool real = i_lane < v.Nh;
    const long i = real ? i_lane : v.Nh - 1;
    const unsigned ib = (unsigned)i * (unsigned)sizeof(NF);
    SebIn<NF> in = {ldg(v.Tair, ib), ldg(v.pres, ib), ldg(v.wind, ib), ldg(v.qair, ib), ldg(v.rain, ib), ldg(v.swd, ib), ldg(v.lwd, ib), NF(0), NF(0), NF(0)};
    seb_radiation_inputs(p, v.albedo, v.emissivity, ib, in);
    SebOut<NF> o;
    uint32_t viol = 0;
    const unsigned ib2 = block_local(ib);
    const NF T_top = ldg(v.top_T, ib2), sat_top = ldg(v.top_sat, ib2), liq_top = ldg(v.top_liq, ib2);
    const NF Kf_top = conductivity_hydraulic<NF, HYD, false>(p, liq_top, fractions(p, sat_top, liq_top, viol));
    surface_processes(p, in, ldg(v.Ts, ib2), T_top, sat_top, liq_top, Kf_top, ldg(v.S, ib2), RICHARDS, v.dzc[v.Nz - 1], o);
    // The wave's 64 columns own 64 G consecutive granules.  Written lane by lane they are 8-byte pieces 8 G bytes apart, each its
    // own write-through transaction of a partial line (measured: +2.7 us on the N145 step, profiles/r05/exp3c); transposed through
    // LDS, store j of lane l writes granule 64 j + l -- 512 contiguous bytes per instruction, whole lines, every granule still ONE
    // 8-byte store of one lane.
    {
        __shared__ unsigned long long stage[(TRM_STEP_BLOCK / 64) * 64 * FRONT_GRANULES];
        const int lane = (int)(threadIdx.x & 63u), wv = (int)(threadIdx.x >> 6);
        unsigned long long* st = stage + wv * 64 * G;
        const unsigned long long tag = (unsigned long long)(fa.epoch + fa.tag_bias) << 32;
        unsigned long long* mine = st + lane * G;
        const NF vals[3] = {o.ghf, o.infil, o.Ts};
        for (int q = 0; q < 3; ++q) {
            if constexpr (sizeof(NF) == 8) {
                const unsigned long long b = __builtin_bit_cast(unsigned long long, vals[q]);
                mine[2 * q] = tag | (b & 0xffffffffull);
                mine[2 * q + 1] = tag | (b >> 32);
            } else {
                mine[q] = tag | (unsigned long long)__builtin_bit_cast(unsigned, vals[q]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // (LDS operations of a wave complete in order; this orders the compiler)
        __builtin_amdgcn_wave_barrier();
        const long first = (i_lane - lane) * G, total = (long)v.Nh * G;     // (i_lane - lane: the wave's first column)
        for (int j = 0; j < G; ++j) {
            const long q = first + j * 64 + lane;
            if (q < total) st_agent(fa.gran + q, st[j * 64 + lane]);
        }
    }
    if (real) {
        const unsigned ob = block_local(ib);
        stg(v.ghf, ob, o.ghf); stg(v.swu, ob, o.swu); stg(v.lwu, ob, o.lwu); stg(v.rnet, ob, o.rnet);
        stg(v.Hs, ob, o.Hs); stg(v.Hl, ob, o.Hl); stg(v.evap, ob, o.evap); stg(v.infil, ob, o.infil); stg(v.runoff, ob, o.runoff);
    }
}
template <class NF, bool RICHARDS, int HYD, bool FROM_STATE, bool TOP_ARRAYS> __global__ void k_surface(View<NF> v, DevParams<NF> p) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < v.Nh) surface_program<NF, RICHARDS, HYD, FROM_STATE, TOP_ARRAYS>(v, p, i);
}

// compute_tendencies_kernel! for SoilHydrology{RichardsEq} (soil_hydrology_rre.jl:150-162) and
// SoilEnergyBalance (soil_energy.jl:153-156), hydrology first (soil_coupled.jl:80-90).
template <class NF, bool RICHARDS> __global__ void k_tendencies(View<NF> v, DevParams<NF> p) {
    TRM_CELL_INDEX(v);
    const int Nz = v.Nz;
    uint32_t viol = 0;
    if (RICHARDS) {
        NF psi0 = v.psi[c];
        NF psim = (k > 0) ? v.psi[c - 1] : halo_bottom(v.bc.kind[4][0], bcval(v, 4, 0), i, psi0, v.g);
        NF psip = (k < Nz - 1) ? v.psi[c + 1] : halo_top(v.bc.kind[4][1], bcval(v, 4, 1), i, psi0, v.g);
        // face conductivities k-1 .. k+2 (halo faces are never written: 0; face Nz lives in Kf_top)
        auto face = [&](int f) { return f < 0 || f > Nz ? NF(0) : (f == Nz ? v.Kf_top[i] : v.Kf[i * v.Nzp + f]); };
        NF Km = face(k - 1), K0 = face(k), K1 = face(k + 1), K2 = face(k + 2);
        NF g_lo = (psi0 - psim) * v.rdzf[k];
        NF g_hi = (psip - psi0) * v.rdzf[k + 1];
        NF Klo = upwind_conductivity(g_lo, Km, K0, K1);
        NF Khi = upwind_conductivity(g_hi, K0, K1, K2);
        NF q_lo = -Klo * g_lo, q_hi = -Khi * g_hi;
        NF dtheta = -((q_hi - q_lo) * v.rdzc[k]) + NF(0) + (v.Fvwc ? v.Fvwc[c] : p.vwc_forcing);
        v.G_sat[c] += div_const(dtheta, p.por, p.rpor);
        if (k == 0) v.G_S[i] += jl_min(NF(0), v.S[i]);
    }
    {
        NF T0 = v.T[c], s0 = v.sat[c], l0 = v.liq[c];
        NF Tm, sm, lm, Tp, sp, lp;
        if (k > 0) { Tm = v.T[c - 1]; sm = v.sat[c - 1]; lm = v.liq[c - 1]; }
        else {
            Tm = halo_bottom(v.bc.kind[2][0], bcval(v, 2, 0), i, T0, v.g);
            lm = halo_bottom(v.bc.kind[3][0], bcval(v, 3, 0), i, l0, v.g);
            sm = sat_halo<NF, RICHARDS>(v, p, 0, i, s0);
        }
        if (k < Nz - 1) { Tp = v.T[c + 1]; sp = v.sat[c + 1]; lp = v.liq[c + 1]; }
        else {
            Tp = halo_top(v.bc.kind[2][1], bcval(v, 2, 1), i, T0, v.g);
            lp = halo_top(v.bc.kind[3][1], bcval(v, 3, 1), i, l0, v.g);
            sp = sat_halo<NF, RICHARDS>(v, p, 1, i, s0);
        }
        NF k0 = conductivity(p, fractions(p, s0, l0, viol));
        NF km = conductivity(p, fractions(p, sm, lm, viol));
        NF kp = conductivity(p, fractions(p, sp, lp, viol));
        NF q_lo = -(NF(0.5) * (k0 + km)) * ((T0 - Tm) * v.rdzf[k]);
        NF q_hi = -(NF(0.5) * (kp + k0)) * ((Tp - T0) * v.rdzf[k + 1]);
        v.G_U[c] += -((q_hi - q_lo) * v.rdzc[k]);
    }
    if (viol) atomicOr(v.status, viol);
}

// compute_z_bcs! + explicit_step_*_kernel! (abstract_timestepper.jl:65-141)
template <class NF, bool RICHARDS> __global__ void k_explicit_step(View<NF> v, DevParams<NF> p, NF dt) {
    TRM_CELL_INDEX(v);
    const int Nz = v.Nz;
    NF gU = v.G_U[c];
    NF gS = RICHARDS ? v.G_sat[c] : NF(0);
    if (k == 0) {
        if (v.bc.kind[0][0] == 2) gU += flux_term_bottom(bcval(v, 0, 0)[i], v.g);
        if (RICHARDS && v.bc.kind[1][0] == 2) gS += flux_term_bottom(bcval(v, 1, 0)[i], v.g);
    }
    if (k == Nz - 1) {
        if (p.seb) gU -= flux_term_top(v.ghf[i], v.g);                       // land_model.jl:56-58
        else if (v.bc.kind[0][1] == 2) gU -= flux_term_top(bcval(v, 0, 1)[i], v.g);
        if (RICHARDS) {
            if (p.seb) gS -= flux_term_top(-v.infil[i], v.g);                // land_model.jl:57-61
            else if (v.bc.kind[1][1] == 2) gS -= flux_term_top(bcval(v, 1, 1)[i], v.g);
        }
    }
    if (k == 0 || k == Nz - 1) {  // compute_z_bcs! modifies the stored tendency
        v.G_U[c] = gU;
        if (RICHARDS) v.G_sat[c] = gS;
    }
    NF u = v.U[c] + gU * dt;
    v.U[c] = u;
    bool bad = is_nan(u);
    if (RICHARDS) {
        NF s = v.sat[c] + gS * dt;
        v.sat[c] = s;
        bad = bad || is_nan(s);
        if (k == 0) v.S[i] = v.S[i] + v.G_S[i] * dt;
    }
    if (k == 0 && p.seb) v.Ts[i] = v.Ts[i] + NF(0) * dt;  // skin_temperature: prognostic with zero tendency
    if (bad) atomicOr(v.status, 1u);
}

// pressure_to_saturation_kernel! (soil_hydraulic_closures.jl:74-100)
template <class NF> __global__ void k_pressure_to_saturation(View<NF> v, DevParams<NF> p) {
    TRM_CELL_INDEX(v);
    NF z = v.zC[k];
    NF psiz = v.psiz[k];
    NF psih = jl_max(NF(0), v.wt[i] - z);
    NF psim = v.psi[c] - psih - psiz;
    v.sat[c] = swrc_theta(p, psim, p.por) / p.por;
}
// energy_to_temperature_kernel! / temperature_to_energy_kernel! (soil_energy_closures.jl:163-171)
template <class NF> __global__ void k_closure_energy(View<NF> v, DevParams<NF> p) {
    TRM_CELL_INDEX(v);
    (void)i; (void)k;
    uint32_t viol = 0;
    NF l, t;
    energy_closure(p, v.U[c], v.sat[c], l, t, viol);
    v.liq[c] = l;
    v.T[c] = t;
    if (viol) atomicOr(v.status, viol);
}
template <class NF> __global__ void k_invclosure_energy(View<NF> v, DevParams<NF> p) {
    TRM_CELL_INDEX(v);
    (void)i; (void)k;
    uint32_t viol = 0;
    NF l, u;
    energy_invclosure(p, v.T[c], v.sat[c], l, u, viol);
    v.liq[c] = l;
    v.U[c] = u;
    if (viol) atomicOr(v.status, viol);
}
// Heun: state.tendencies .= (state.tendencies + stage.tendencies) / 2 (heun.jl:27-35)
// update_inputs! for time series sources (input_sources.jl:162-168): dst = v2 * f + v1 * (1 - f) in double,
// rounded once to NF; f == 0 with v1 == v2 (same node) is the plain copy.  Up to 16 series per launch
// (blockIdx.y = series).
// raster: the Raster input source's rule x1 + eps * (x2 - x1) / dt (ext/TerrariumRastersExt.jl:96-121) with f = eps,
// g = dt [s]; (x2 - x1) is formed in NF as the reference's array expression does, the rest in double.
template <class NF> struct SeriesJob { NF* dst; const NF* v1; const NF* v2; double f, g; int raster; };
template <class NF> struct SeriesJobs { SeriesJob<NF> job[16]; };
template <class NF> __global__ void k_interp_series(SeriesJobs<NF> jobs, long Nh) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Nh) return;
    const SeriesJob<NF>& j = jobs.job[blockIdx.y];
    if (j.v1 == j.v2) {
        j.dst[i] = j.v1[i];
    } else {
        const double a = (double)j.v1[i], b = (double)j.v2[i];
        if (j.raster) j.dst[i] = (NF)(a + j.f * (double)(NF)(j.v2[i] - j.v1[i]) / j.g);
        else j.dst[i] = (NF)(b * j.f + a * (1.0 - j.f));
    }
}

template <class NF> __global__ void k_average(NF* a, const NF* b, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = (a[i] + b[i]) / NF(2);
}

// hydrology closure!, generic fallback with one thread per column and a sequential walk in z
// (any Nz; used when Nz > 64): adjust_saturation_profile! (soil_hydrology.jl:185-219),
// compute_water_table! (soil_hydrology.jl:170-175) and, if WITH_PSI, saturation_to_pressure!.
template <class NF, bool WITH_PSI, int HYD, bool WITH_ADJUST> __global__ void k_closure_hydrology_seq(View<NF> v, DevParams<NF> p) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= v.Nh) return;
    const int Nz = v.Nz;
    NF* s = v.sat + i * v.Nzp;
    if (WITH_ADJUST) {
        for (int k = 0; k <= Nz - 2; ++k) {
            NF sk = s[k];
            NF e = jl_max(sk - NF(1), NF(0));
            s[k] = sk - e;
            s[k + 1] += div_const(e * v.dzc[k], v.dzc[k + 1], v.rdzc[k + 1]);
        }
        for (int k = Nz - 1; k >= 1; --k) {
            NF sk = s[k];
            NF d = jl_max(-sk, NF(0));
            s[k] = sk + d;
            s[k - 1] -= div_const(d * v.dzc[k], v.dzc[k - 1], v.rdzc[k - 1]);
        }
        NF st = s[Nz - 1];
        NF e = jl_max(st - NF(1), NF(0));
        s[Nz - 1] = st - e;
        v.S[i] += e * v.dzc[Nz - 1];
        s[0] = jl_max(s[0], NF(0));
    }
    int idx = -1;
    for (int k = 0; k < Nz; ++k)
        if (idx < 0 && s[k] < NF(1)) idx = k;
    // (the reference scan also visits the halo above the top cell, which maps to the same surface node
    // as "not found")
    NF z0 = v.zF[idx >= 0 ? idx : Nz];
    v.wt[i] = z0;
    if (WITH_PSI)
        for (int k = 0; k < Nz; ++k) v.psi[i * v.Nzp + k] = pressure_head<NF, HYD>(p, s[k], v.zC[k], v.psiz[k], z0);
}

// ===========================================================================
// lane = level machinery
// ===========================================================================
template <class NF, int LPC> TRM_DEV NF shfl_from(NF x, int src_k) { return __shfl(x, src_k, LPC); }
// Neighbour exchange k-1 / k+1 as DPP whole-wave shifts (wave_shr:1 / wave_shl:1, gfx9 family): a VALU
// move per dword instead of a round trip through the LDS crossbar (ds_bpermute).  The shift runs over
// the whole wavefront, so with two columns per wave the lanes at a column edge receive the
// neighbouring column's value -- exactly the lanes (bottom / top level) whose input is replaced by
// the halo / boundary-face value anyway.
TRM_DEV int dpp_shr1(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x138, 0xf, 0xf, true); }
TRM_DEV int dpp_shl1(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x130, 0xf, 0xf, true); }
TRM_DEV double shift_up(double x) {  // lane l receives lane l-1 (level k-1)
    union { double d; int w[2]; } u;
    u.d = x;
    u.w[0] = dpp_shr1(u.w[0]);
    u.w[1] = dpp_shr1(u.w[1]);
    return u.d;
}
TRM_DEV double shift_dn(double x) {  // lane l receives lane l+1 (level k+1)
    union { double d; int w[2]; } u;
    u.d = x;
    u.w[0] = dpp_shl1(u.w[0]);
    u.w[1] = dpp_shl1(u.w[1]);
    return u.d;
}
TRM_DEV float shift_up(float x) { return __builtin_bit_cast(float, dpp_shr1(__builtin_bit_cast(int, x))); }
TRM_DEV float shift_dn(float x) { return __builtin_bit_cast(float, dpp_shl1(__builtin_bit_cast(int, x))); }
template <class NF, int LPC> TRM_DEV NF shfl_up1(NF x) { return shift_up(x); }
template <class NF, int LPC> TRM_DEV NF shfl_dn1(NF x) { return shift_dn(x); }
// ballot -> set of LEVELS at which any of the wave's columns has the bit set
template <int LPC> TRM_DEV unsigned long long level_bits(unsigned long long ballot) {
    if (LPC == 64) return ballot;
    return (ballot | (ballot >> 32)) & 0xffffffffull;
}
template <int LPC> TRM_DEV unsigned long long group_mask(int lane) {
    if (LPC == 64) return ~0ull;
    return (lane & 32) ? 0xffffffff00000000ull : 0x00000000ffffffffull;
}

// A wave-uniform 64-bit lane mask as a per-lane predicate WITHOUT a vector compare: selects and exec masks take the scalar
// register pair as it is.  "Which lanes hold the bottom / top cell, a real cell, the second column of the wave" are functions of
// launch constants -- formed on the scalar unit (the lane-wise compares they replace were 1 vector instruction each).
TRM_DEV bool lane_in(unsigned long long mask) { return __builtin_amdgcn_inverse_ballot_w64(mask); }
// the lanes of level k of every column of the wave (LPC lanes per column)
template <int LPC> TRM_DEV unsigned long long level_lanes(int k) {
    const unsigned long long one = 1ull << k;
    return LPC == 64 ? one : (one | (one << 32));
}
// the lanes of levels 0 .. n - 1 of every column of the wave
template <int LPC> TRM_DEV unsigned long long levels_below(int n) {
    if (LPC == 64) return n >= 64 ? ~0ull : ((1ull << n) - 1ull);
    const unsigned long long m = n >= 32 ? 0xffffffffull : ((1ull << n) - 1ull);
    return m | (m << 32);
}
// value of lane `src` (a lane index, per lane) -- ds_bpermute without HIP's own lane arithmetic (__shfl recomputes the lane id
// with two v_mbcnt)
TRM_DEV double lane_read(double x, int src_lane) {
    union { double d; int w[2]; } u;
    u.d = x;
    u.w[0] = __builtin_amdgcn_ds_bpermute(src_lane << 2, u.w[0]);
    u.w[1] = __builtin_amdgcn_ds_bpermute(src_lane << 2, u.w[1]);
    return u.d;
}
TRM_DEV float lane_read(float x, int src_lane) { return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane << 2, __builtin_bit_cast(int, x))); }

// per-lane grid constants of level k (fixed for the whole kernel: lane <-> level)
// Global accesses with a 32-bit BYTE offset: `scalar base + zero-extended vector offset` is the one form the
// backend maps onto global_load/store's saddr addressing, so a cell's ~30 accesses share a single offset
// register instead of one 64-bit address computation each.  trm_create checks Nh * Nzp * sizeof(NF) < 2^32.
template <class NF> TRM_DEV NF ldg(const NF* base, unsigned byte_off) {
    return *reinterpret_cast<const NF*>(reinterpret_cast<const char*>(base) + byte_off);
}
// A value that is the same for every lane of a column (boundary values, the 0-D fields) through the SCALAR memory path: element
// `idx` (wave-uniform) of a per-column array.  A vector load of it returns 64 lanes' worth of data through the vector return path
// for 8 or 16 useful bytes -- as expensive there as a full field read (profiles/r03/exp24: the column program without these loads
// runs 13 % faster at C4, 6.6 % at 8 x N145).  The constant address space makes the backend select s_load for a uniform address;
// the arrays are not written by the wave before it reads them, and a kernel boundary invalidates the scalar cache.
template <class NF> TRM_DEV NF sld(const NF* base, int idx_uniform) {
    typedef const NF __attribute__((address_space(4))) * cptr;
    return ((cptr)(uintptr_t)base)[idx_uniform];
}
// The same with a wave-uniform 32-bit BYTE offset: the backend then uses the scalar load's own register offset
// (`s_load_dwordx2 sdst, sbase, soffset`) instead of forming a 64-bit address per load on the scalar ALU (4 instructions per
// load: ~100 of the 260 scalar instructions per wave of the packed fp32 LandModel step).  T may be a pair (one wider load).
template <class T> TRM_DEV T sld_off(const void* base, unsigned byte_off_uniform) {
#if !TRM_CUT_SOFF
    return sld((const T*)base, (int)(byte_off_uniform / (unsigned)sizeof(T)));
#endif
    typedef const char __attribute__((address_space(4))) * bptr;
    typedef const T __attribute__((address_space(4))) * cptr;
    return *(cptr)((bptr)(uintptr_t)base + byte_off_uniform);
}
template <class NF> TRM_DEV void stg(NF* base, unsigned byte_off, NF x) {
    *reinterpret_cast<NF*>(reinterpret_cast<char*>(base) + byte_off) = x;
}
// Instruction selection works one basic block at a time: an offset defined in another block has already been
// widened to 64 bits there.  Re-materialising it (a no-op the optimiser cannot see through) keeps the
// zero-extension, and with it the saddr form, inside the block that does the access.
TRM_DEV unsigned block_local(unsigned byte_off) {
    asm volatile("" : "+v"(byte_off));
    return byte_off;
}

template <class NF> struct LevelGeom {
    NF zC, psiz, zFlo, dzc, rdzc, rdzf_lo, rdzf_hi, zF_top, dzc_top;
};
template <class NF> struct alignas(16) LevelPack { NF w[8]; };
template <class NF> TRM_DEV LevelGeom<NF> level_geom(const View<NF>& v, int k) {
    const int Nz = v.Nz;
    const int kk = k < Nz ? k : Nz - 1;
    // one 16-byte-aligned record per level: 4 (fp64) / 2 (fp32) wide loads instead of 7 narrow ones.
    // CALL IT BEHIND THE FIELD LOADS of a kernel: the record's unused eighth value lands in a register pair nobody reads, the allocator
    // hands that pair to whatever arithmetic follows, and if that is the address arithmetic of the field loads the write-after-write
    // hazard puts an `s_waitcnt vmcnt` -- a full trip to memory -- between the record and the fields (round 4: found in every
    // k_column instance that reads T / liq; profiles/tools/check_load_order.py).  (Loading 7 values does not help: the backend
    // widens the last 8-byte piece to 16 again.)
    const LevelPack<NF> q = *reinterpret_cast<const LevelPack<NF>*>(reinterpret_cast<const char*>(v.lvl) + (unsigned)kk * (unsigned)sizeof(LevelPack<NF>));
    LevelGeom<NF> L;
    L.zC = q.w[0]; L.psiz = q.w[1]; L.zFlo = q.w[2]; L.dzc = q.w[3]; L.rdzc = q.w[4]; L.rdzf_lo = q.w[5]; L.rdzf_hi = q.w[6];
    L.zF_top = v.g.zF_top; L.dzc_top = v.g.dzc_top;
    return L;
}
// thickness of the neighbouring cells (serial repair passes only)
template <class NF> struct NeighbourDz { NF dzc_up, rdzc_up, dzc_dn, rdzc_dn; };
template <class NF> TRM_DEV NeighbourDz<NF> neighbour_dz(const View<NF>& v, int k) {
    const int Nz = v.Nz;
    const int kk = k < Nz ? k : Nz - 1;
    const int ku = kk + 1 < Nz ? kk + 1 : kk, kd = kk > 0 ? kk - 1 : 0;
    NeighbourDz<NF> n;
    n.dzc_up = v.dzc[ku]; n.rdzc_up = v.rdzc[ku]; n.dzc_dn = v.dzc[kd]; n.rdzc_dn = v.rdzc[kd];
    return n;
}

// adjust_saturation_profile! (soil_hydrology.jl:185-219) on a column held one level per lane.
// Both passes are sequential in z.  One ballot decides: if no cell below the top is oversaturated the
// upward pass only applies its `+ 0`; the downward pass then sees the same profile, so the same ballot also
// covers "no cell above the bottom is negative", and the whole repair reduces to `s + 0` plus the top
// overflow and the bottom clamp, all lane-local.  Otherwise both passes run as lane-serial loops.
// Returns excess * dz_top in the TOP lane (0 elsewhere): the column's overflow into surface_excess_water.
template <class NF, int LPC>
TRM_DEV NF repair_saturation(const View<NF>& v, NF& snew, int k, int Nz, unsigned long long m_act, bool is_bot, bool is_top,
                             const LevelGeom<NF>& L, unsigned long long m_bot_known = 0ull, unsigned long long m_top_known = 0ull) {
    // (max(s - 1, 0) != 0  <=>  s > 1  and  max(-s, 0) != 0  <=>  s < 0, NaN included: both sides false)
    // Every ballot is taken of ONE compare and the masks are combined on the scalar unit: the ballot of a conjunction goes
    // through a vector register (v_cndmask 0/1 + v_cmp_ne).  m_act: the wave's mask of lanes that hold a cell.
    // (a caller that formed is_bot / is_top FROM wave-uniform masks hands the masks over: the ballot of such a bool is a round trip
    //  through a vector register, v_cndmask 0/1 + v_cmp_ne, for a value the scalar unit already holds)
    const unsigned long long m_top = m_top_known ? m_top_known : wave_ballot(is_top), m_bot = m_bot_known ? m_bot_known : wave_ballot(is_bot);
    const unsigned long long any_over = wave_ballot(snew > NF(1)) & m_act & ~m_top;
    const unsigned long long any_bad = any_over | (wave_ballot(snew < NF(0)) & m_act & ~m_bot);
    // every cell but the bottom one receives `+ carry` / `+ deficit`; with nothing to move that is `+ 0`
    // (idempotent, and absorbed by a later non-zero addend), so apply it once up front
    snew = is_bot ? snew : snew + NF(0);
    if (any_bad != 0ull) {
        TRM_PHASE("rare+ repair");
        // thickness of the cells above / below, from the neighbouring lanes' level records (wave-uniform branch: every lane
        // executes the shifts; the edge lanes, which would receive the next column's value, keep their own as
        // neighbour_dz() does -- no loads in this rare path, so nothing of it is ever pending in the common one)
        NeighbourDz<NF> nb;
        {
            const NF dz_up = shfl_dn1<NF, LPC>(L.dzc), rdz_up = shfl_dn1<NF, LPC>(L.rdzc);
            const NF dz_dn = shfl_up1<NF, LPC>(L.dzc), rdz_dn = shfl_up1<NF, LPC>(L.rdzc);
            const bool edge_up = k >= Nz - 1;
            nb.dzc_up = edge_up ? L.dzc : dz_up;
            nb.rdzc_up = edge_up ? L.rdzc : rdz_up;
            nb.dzc_dn = is_bot ? L.dzc : dz_dn;
            nb.rdzc_dn = is_bot ? L.rdzc : rdz_dn;
        }
        // Lane-serial passes, restricted to the levels that can change: the upward pass starts at the lowest
        // oversaturated level and stops once the carry is zero with no oversaturated level left above;
        // outside that range the reference's updates are the `+ 0` already applied.
        if (any_over != 0ull) {
            const unsigned long long lv = level_bits<LPC>(any_over);
            NF carry = NF(0);
            for (int q = __builtin_ctzll(lv); q < Nz - 1; ++q) {
                NF cout = NF(0);
                if (k == q) {
                    snew = snew + carry;
                    NF e = jl_max(snew - NF(1), NF(0));
                    snew = snew - e;
                    cout = div_const(e * L.dzc, nb.dzc_up, nb.rdzc_up);
                }
                carry = shfl_from<NF, LPC>(cout, q);
                if ((lv >> (q + 1)) == 0ull && wave_ballot(!(carry == NF(0))) == 0ull) break;
            }
            if (is_top) snew = snew + carry;
        }
        const unsigned long long any_under = wave_ballot(!(jl_max(-snew, NF(0)) == NF(0))) & m_act & ~m_bot;
        if (any_under != 0ull) {
            const unsigned long long lv = level_bits<LPC>(any_under);
            NF pend = NF(0);
            for (int q = 63 - __builtin_clzll(lv); q >= 1; --q) {
                NF pout = NF(0);
                if (k == q) {
                    snew = snew - pend;
                    NF d = jl_max(-snew, NF(0));
                    snew = snew + d;
                    pout = div_const(d * L.dzc, nb.dzc_dn, nb.rdzc_dn);
                }
                pend = shfl_from<NF, LPC>(pout, q);
                if ((lv & ((1ull << q) - 1ull)) == 0ull && wave_ballot(!(pend == NF(0))) == 0ull) break;
            }
            if (is_bot) snew = snew - pend;
        }
    }
    TRM_PHASE("rare-");
    // surface overflow joins surface_excess_water (top lane); bottom clamp (bottom lane)
    const NF e_top = is_top ? jl_max(snew - NF(1), NF(0)) : NF(0);
    snew = snew - e_top;
    snew = is_bot ? jl_max(snew, NF(0)) : snew;
    return e_top * L.dzc_top;
}
// compute_water_table! (soil_hydrology.jl:170-175, kernel_utils.jl:7-16): lower face of the first
// unsaturated cell from the bottom, the surface if there is none.
// The search runs on the SCALAR unit: one ballot, the first set bit of each column's half of it (s_ff1), the source lane of
// the lookup selected per half-wave -- the lane-wise form (the ballot masked per lane, 64-bit ffs in vector registers) was ~20
// vector instructions.
template <class NF, int LPC> TRM_DEV NF water_table(NF sat, unsigned long long m_act, int lane, const LevelGeom<NF>& L) {
#if !TRM_CUT_WT
    {
        const unsigned long long unsat_l = wave_ballot(sat < NF(1)) & m_act & group_mask<LPC>(lane);
        const int first = unsat_l ? (__ffsll((long long)unsat_l) - 1) % LPC : -1;
        const NF z_first = shfl_from<NF, LPC>(L.zFlo, first >= 0 ? first : 0);
        return first >= 0 ? z_first : L.zF_top;
    }
#endif
    const unsigned long long unsat = wave_ballot(sat < NF(1)) & m_act;
    if (LPC == 64) {
        const int src = unsat ? __builtin_ctzll(unsat) : 0;                       // (uniform)
        const NF z_first = lane_read(L.zFlo, src);
        return unsat ? z_first : L.zF_top;
    }
    const unsigned lo = (unsigned)unsat, hi = (unsigned)(unsat >> 32);
    const int src_lo = lo ? __builtin_ctz(lo) : 0, src_hi = 32 + (hi ? __builtin_ctz(hi) : 0);    // (uniform: one per column)
    const bool upper = lane_in(0xffffffff00000000ull);
    const NF z_first = lane_read(L.zFlo, upper ? src_hi : src_lo);
    const unsigned long long found = (lo ? 0x00000000ffffffffull : 0ull) | (hi ? 0xffffffff00000000ull : 0ull);
    return lane_in(found) ? z_first : L.zF_top;
}

// hydrology closure! with lane = level (Nz <= 64): grid = ceil(Nh / (64 / LPC)) waves
template <class NF, bool WITH_PSI, int HYD, bool WITH_ADJUST, int LPC>
__global__ void k_closure_hydrology_wave(View<NF> v, DevParams<NF> p) {
    constexpr int CPW = 64 / LPC;
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int k = lane % LPC, sub = lane / LPC;
    const long i = wave * CPW + sub;
    const int Nz = v.Nz;
    const bool act = i < v.Nh && k < Nz;
    const bool is_bot = k == 0, is_top = k == Nz - 1;
    const LevelGeom<NF> L = level_geom(v, k);
    const long c = (i < v.Nh ? i : v.Nh - 1) * v.Nzp + (k < Nz ? k : Nz - 1);
    NF s = v.sat[c];
    if (WITH_ADJUST) {
        NF over = repair_saturation<NF, LPC>(v, s, k, Nz, wave_ballot(act), is_bot, is_top, L);
        if (act) {
            v.sat[c] = s;
            if (is_top) v.S[i] += over;
        }
    }
    NF z0 = water_table<NF, LPC>(s, wave_ballot(act), lane, L);
    if (act) {
        if (is_bot) v.wt[i] = z0;
        if (WITH_PSI) v.psi[c] = pressure_head<NF, HYD>(p, s, L.zC, L.psiz, z0);
    }
}

// ===========================================================================
// Fused step (TRM_KERNEL_FUSED): update_state! + explicit_step! + closure! (+ the K of
// compute_auxiliary! when finalizing) in ONE launch, forward_euler.jl:19-31.
// lane = level, one column per LPC lanes; every wave is independent (no LDS, no barrier):
// 5 coalesced loads, ~300 fp64 instructions, 6 coalesced stores per cell.
// ===========================================================================

// k_step_wave serves every boundary kind -- Gradient conditions, Value conditions on liquid fraction / saturation /
// pressure head, a per-cell vwc_forcing field -- with the halo values formed by the edge lanes.  The common case (Value
// on temperature, Flux on the prognostics: everything the reference's models and examples set up) takes the branch-free
// column programs of trm_column.hpp (k_column), which also hold Heun and the multi-step program.
//
// Diagnostic variants (memory-only, compute-only) are NOT in this translation unit:
// profiles/tools/make_diag_variants.py derives them from the shipped sources into build/diag/.
// Kernel arguments re-read through an opaque pointer into the kernarg segment.  The step kernels are short of scalar
// registers: View + DevParams hold ~200 scalars, and a value used both early and late in the kernel (or, in a loop, in
// every iteration) is kept in an SGPR throughout -- what does not fit is parked in VGPR lanes at a v_writelane /
// v_readlane (VALU) each: 57-150 extra VALU per wave were measured.  Read through a pointer the optimiser cannot see
// through, the second half of the kernel fetches its scalars afresh (s_load from the constant cache), the first half's
// die early, and nothing spills.
template <class T> TRM_DEV const T& kernarg_reload(unsigned offset) {
    typedef const __attribute__((address_space(4))) char* kptr;
    kptr kp = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    return *(const T*)(kp + offset);
}

// Staging table of a workgroup's per-column outputs: [SMALL_COUNT][columns per workgroup] (at most 8 x 16 values).
template <class NF> TRM_DEV NF* small_stage() {
    __shared__ NF table[SMALL_COUNT * 16];
    return table;
}
// After a barrier, lane l of the workgroup's first waves stores entry l of the table: array l / cpb, column l % cpb of the
// workgroup -- every array receives one contiguous run of cpb values from a single instruction.  The array pointers come from
// the kernel argument segment (View::small), indexed per lane.  `enabled`: bit per array (wave-uniform).
// CPB (columns per workgroup) is a compile-time constant -- every launch of these programs has TRM_STEP_BLOCK threads -- so the
// lane's array and column are a shift and a mask (a run-time divisor cost the storing wave a 12-instruction division), and the
// lane's array pointer is requested BEFORE the barrier: that vector load from the argument segment used to sit, a full trip to
// memory, between the barrier and the last store of every workgroup.
template <class NF, int CPB> TRM_DEV void store_small_outputs(unsigned enabled, unsigned block, int Nh) {
    static_assert((CPB & (CPB - 1)) == 0 && SMALL_COUNT * CPB <= TRM_STEP_BLOCK, "columns per workgroup: a power of two");
    const int l = (int)threadIdx.x;
    const int slot = l / CPB, col = l % CPB;
    const long i = (long)block * CPB + col;
    const bool mine = l < SMALL_COUNT * CPB && i < Nh && ((enabled >> slot) & 1u);
    NF* dst = nullptr;
    if (mine) dst = kernarg_reload<View<NF>>(0).small[slot];
    __syncthreads();
    if (mine) dst[i] = small_stage<NF>()[l];
}

constexpr unsigned round_up_to(unsigned x, unsigned a) { return (x + a - 1) / a * a; }

template <class NF, bool RICHARDS, int HYD, int LPC>
__global__ void __launch_bounds__(TRM_STEP_BLOCK) k_step_wave(View<NF> v_arg, DevParams<NF> p_arg, NF dt, int finalize, int write_kf) {
    // (kernarg layout: the arguments in order, each at its natural alignment)
    constexpr unsigned off_p = round_up_to((unsigned)sizeof(View<NF>), (unsigned)alignof(DevParams<NF>));
    const View<NF>& v = v_arg;
    const DevParams<NF>& p = p_arg;
    constexpr int CPW = 64 / LPC;   // columns per wave
    const int lane = threadIdx.x & 63;
    const int wave = (int)((blockIdx.x * (unsigned)blockDim.x + threadIdx.x) >> 6);
    const int k = lane % LPC, sub = lane / LPC;
    const int Nz = v.Nz, Nh = (int)v.Nh;
    const bool is_bot = k == 0, is_top = k == Nz - 1;
    const LevelGeom<NF> L = level_geom(v, k);
    const bool need_kc = RICHARDS || write_kf;

    const int i = wave * CPW + sub;
    const bool colok = i < Nh;              // uniform within the column's lanes
    const bool act = colok && k < Nz;
    const int ii = colok ? i : Nh - 1;      // safe index for the tail wave
    const unsigned ib0 = (unsigned)ii * (unsigned)sizeof(NF);
    const unsigned cb0 = ((unsigned)ii * (unsigned)v.Nzp + (unsigned)(k < Nz ? k : Nz - 1)) * (unsigned)sizeof(NF), cb = cb0;
    uint32_t viol = 0;

    const NF U = ldg(v.U, cb), sat = ldg(v.sat, cb);
    const NF T = ldg(v.T, cb), liq = ldg(v.liq, cb);
    const NF psi = RICHARDS ? ldg(v.psi, cb) : NF(0);
    const NF U0 = U, sat0 = sat;
    // (read with the other inputs: a load behind a store would hold the wave until that store is acknowledged -- loads and
    // stores retire through the one in-order counter)
    const NF Ts_in = p.seb ? ldg(v.Ts, ib0) : NF(0);

    // (composition bounds of the incoming state were flagged by the launch that produced it)
    uint32_t viol_old = 0;
    const Frac<NF> f = fractions(p, sat, liq, viol_old);
    const NF kap = conductivity(p, f);
    const NF Kc = need_kc ? conductivity_hydraulic<NF, HYD, false>(p, liq, f) : NF(0);

    // ---- neighbours by DPP row shifts (executed by all lanes, never inside a divergent select) -----------
    const NF T_sh = shfl_up1<NF, LPC>(T), kap_sh = shfl_up1<NF, LPC>(kap);
    // halo cells below the bottom / above the top cell
    NF T_m = T_sh, kap_m = kap_sh, T_h = NF(0), kap_h = NF(0), psi_hb = NF(0), psi_ht = NF(0);
    NF flux_U = NF(0), flux_S = NF(0);  // compute_z_bcs! term of this lane's cell (0 in the interior)
    {
    // With the default (no-flux) conditions on liquid fraction and saturation the halo cell has the edge
    // cell's composition, hence bit for bit its conductivity: nothing to recompute.
    const bool same_bot = v.bc.kind[3][0] != 1 && v.bc.kind[3][0] != 3 &&
                          (RICHARDS ? (v.bc.kind[1][0] != 1 && v.bc.kind[1][0] != 3) : p.halo_policy == 1);
    const bool same_top = v.bc.kind[3][1] != 1 && v.bc.kind[3][1] != 3 &&
                          (RICHARDS ? (v.bc.kind[1][1] != 1 && v.bc.kind[1][1] != 3) : p.halo_policy == 1);
    if (is_bot) {
        T_m = halo_bottom(v.bc.kind[2][0], bcval(v, 2, 0), ii, T, v.g);
        kap_m = kap;
        if (!same_bot) {
            NF lh = halo_bottom(v.bc.kind[3][0], bcval(v, 3, 0), ii, liq, v.g);
            NF sh = sat_halo<NF, RICHARDS>(v, p, 0, ii, sat);
            kap_m = conductivity(p, fractions(p, sh, lh, viol));
        }
        if (RICHARDS) psi_hb = halo_bottom(v.bc.kind[4][0], bcval(v, 4, 0), ii, psi, v.g);
        if (v.bc.kind[0][0] == 2) flux_U = flux_term_bottom(bcval(v, 0, 0)[ii], v.g);
        if (RICHARDS && v.bc.kind[1][0] == 2) flux_S = flux_term_bottom(bcval(v, 1, 0)[ii], v.g);
    }
    if (is_top) {
        T_h = halo_top(v.bc.kind[2][1], bcval(v, 2, 1), ii, T, v.g);
        kap_h = kap;
        if (!same_top) {
            NF lh = halo_top(v.bc.kind[3][1], bcval(v, 3, 1), ii, liq, v.g);
            NF sh = sat_halo<NF, RICHARDS>(v, p, 1, ii, sat);
            kap_h = conductivity(p, fractions(p, sh, lh, viol));
        }
        if (RICHARDS) psi_ht = halo_top(v.bc.kind[4][1], bcval(v, 4, 1), ii, psi, v.g);
        // top flux BCs enter with a minus sign; LandModel wires ground_heat_flux / -infiltration
        // (land_model.jl:56-61), produced by k_surface just before this launch
        if (p.seb) {
            flux_U = -flux_term_top(v.ghf[ii], v.g);
            if (RICHARDS) flux_S = -flux_term_top(-v.infil[ii], v.g);
        } else {
            if (v.bc.kind[0][1] == 2) flux_U = -flux_term_top(bcval(v, 0, 1)[ii], v.g);
            if (RICHARDS && v.bc.kind[1][1] == 2) flux_S = -flux_term_top(bcval(v, 1, 1)[ii], v.g);
        }
    }
    }
    // ---- heat: every lane forms its lower face, the top lane also the boundary face -------------------
    const NF qT_lo = -(NF(0.5) * (kap + kap_m)) * ((T - T_m) * L.rdzf_lo);
    const NF qT_sh = shfl_dn1<NF, LPC>(qT_lo);
    const NF qT_hi = is_top ? -(NF(0.5) * (kap_h + kap)) * ((T_h - T) * L.rdzf_hi) : qT_sh;
    NF gU = NF(0) + (-((qT_hi - qT_lo) * L.rdzc));

    // ---- Richards: face conductivities (soil_hydrology.jl:145-163) and Darcy fluxes -------------------
    NF gS = NF(0), Kf_lo = NF(0);
    if (need_kc) {
        const NF Kc_m = shfl_up1<NF, LPC>(Kc);
        const NF Kmin = jl_min(Kc, Kc_m);   // (formed outside the select: keeps it a select, not a branch)
        Kf_lo = (is_bot || is_top) ? Kc : Kmin;
    }
    if (RICHARDS) {
        const NF Kf_up = shfl_up1<NF, LPC>(Kf_lo), Kf_dn = shfl_dn1<NF, LPC>(Kf_lo), psi_sh = shfl_up1<NF, LPC>(psi);
        const NF Kf_m = is_bot ? NF(0) : Kf_up;   // halo face below: never written (0)
        const NF Kf_p = is_top ? Kc : Kf_dn;      // face Nz repeats the top cell's value
        const NF psi_m = is_bot ? psi_hb : psi_sh;
        const NF g_lo = (psi - psi_m) * L.rdzf_lo;
        const NF Ks_lo = upwind_conductivity(g_lo, Kf_m, Kf_lo, Kf_p);
        const NF qW_lo = -Ks_lo * g_lo;
        const NF qW_sh = shfl_dn1<NF, LPC>(qW_lo);
        NF qW_hi = qW_sh;
        {   // boundary face above the top cell (computed by every lane, kept by the top lane)
            const NF g_t = (psi_ht - psi) * L.rdzf_hi;
            const NF Ks_t = upwind_conductivity(g_t, Kf_lo, Kc, NF(0));
            const NF qW_t = -Ks_t * g_t;
            qW_hi = is_top ? qW_t : qW_sh;
        }
        // (+ 0: the evapotranspiration forcing, never passed for bare ground, soil_coupled.jl:86) + user forcing
        // (the per-cell field is served by the generic instance only: its pointer test costs the van Genuchten
        // instances 5 % through register pressure, and a spatially varying user forcing is the rare case)
        const NF F_user = v.Fvwc ? ldg(v.Fvwc, cb) : p.vwc_forcing;
        const NF dtheta = -((qW_hi - qW_lo) * L.rdzc) + NF(0) + F_user;
        gS = NF(0) + div_const(dtheta, p.por, p.rpor);
    }
    // ---- compute_z_bcs!: flux BCs into the boundary cells (x + 0 is exact for the interior lanes) -------
    gU += flux_U;
    if (RICHARDS) gS += flux_S;
    // ---- explicit Euler update ------------------------------------------------------------------------------------
    const NF Unew = U0 + gU * dt;
    bool bad = act && is_nan(Unew);
    NF snew = sat0, z0 = NF(0);
    if (RICHARDS) {
        snew = sat0 + gS * dt;
        bad = bad || (act && is_nan(snew));
        const unsigned long long m_act = wave_ballot(act);
        const NF over = repair_saturation<NF, LPC>(v, snew, k, Nz, m_act, is_bot, is_top, L);
        z0 = water_table<NF, LPC>(snew, m_act, lane, L);
        if (act && is_top) {
            // surface_excess_water: tendency min(0, S) once per column (SURVEY C-3), Euler update, overflow
            const unsigned ib = block_local(ib0);
            NF S = ldg(v.S, ib);
            const NF GS = NF(0) + jl_min(NF(0), S);
            if (finalize) stg(v.G_S, ib, GS);
            S = S + GS * dt;
            stg(v.S, ib, S + over);
            stg(v.wt, ib, z0);
        }
    }
    if (act && is_top && p.seb) {   // zero-tendency prognostic skin_temperature
        const unsigned ib = block_local(ib0);
        stg(v.Ts, ib, Ts_in + NF(0) * dt);
    }
    // ---- closures: (U, sat) -> (T, liq, psi), parameters fetched afresh (see kernarg_reload) ----------------------
    NF ln, Tn;
    const DevParams<NF>& p2 = kernarg_reload<DevParams<NF>>(off_p);
    energy_closure(p2, Unew, snew, ln, Tn, viol);
    const NF psin = RICHARDS ? pressure_head<NF, HYD>(p2, snew, L.zC, L.psiz, z0) : NF(0);
    NF Kf_out = Kf_lo, Kf_out_top = Kc;
    if (finalize && write_kf) {
        const NF Kc_new = conductivity_hydraulic<NF, HYD, false>(p, ln, fractions(p, snew, ln, viol));
        const NF Kc_new_m = shfl_up1<NF, LPC>(Kc_new);
        const NF Kmin_new = jl_min(Kc_new, Kc_new_m);
        Kf_out = (is_bot || is_top) ? Kc_new : Kmin_new;
        Kf_out_top = Kc_new;
    }
    if (act) {
        const unsigned cb = block_local(cb0), ib = block_local(ib0);
        stg(v.U, cb, Unew);
        stg(v.T, cb, Tn);
        stg(v.liq, cb, ln);
        if (RICHARDS) { stg(v.sat, cb, snew); stg(v.psi, cb, psin); }
        if (finalize) {
            // state.tendencies as the reference leaves them after its last step: compute_tendencies! (averaged for
            // Heun) plus the compute_z_bcs! term explicit_step! added.  Only the finalizing launch stores them.
            stg(v.G_U, cb, gU);
            if (RICHARDS) stg(v.G_sat, cb, gS);
        }
        if (is_top && p.seb) {   // the next surface energy balance reads these
            stg(v.top_T, ib, Tn);
            stg(v.top_sat, ib, snew);
            stg(v.top_liq, ib, ln);
        }
        // hydraulic_conductivity of the state: K(old state), K(new state) when finalizing
        if (write_kf) {
            stg(v.Kf, block_local(cb0), Kf_out);
            if (is_top) stg(v.Kf_top, block_local(ib0), Kf_out_top);
        }
        viol |= bad ? 1u : 0u;
    }
    // (only real cells report: the clamped copies that tail lanes carry are not repaired and may be out of bounds)
    if (viol && act) atomicOr(v.status, viol);
}

}  // namespace trm
