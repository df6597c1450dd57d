// trm_kernel_wave.hpp -- fused step, "column per (half-)wavefront" mapping
// (TRM_KERNEL_FUSED_WAVE; the default for Nz <= 64).
//
// Why this mapping: the global N145 grid has only 56 951 columns.  With one lane per
// column that is 890 wavefronts for 1024 SIMDs -- less than one wave per SIMD, so every
// dependent fp64 divide / compensated power runs at its full latency with nothing to
// overlap.  Here the vertical axis is spread over the lanes instead: lane = soil level,
// one column per 32 lanes (Nz <= 32) or per 64 lanes (Nz <= 64): 28 000+ waves at N145.
//
//   phase L  a workgroup owns a tile of 32 consecutive columns; global loads are issued
//            lanes-over-columns (256-byte coalesced row segments of the SoA layout) and
//            written TRANSPOSED into LDS tiles [column][level]; the row stride is odd so
//            the strided writes and the later strided reads are bank-conflict free;
//   phase B  lane = column: everything that exists once per column -- the halo values
//            implied by the boundary conditions, the flux-BC terms (for LandModel the
//            ground heat flux / infiltration produced by k_surface), the surface-excess-
//            water update -- goes to per-column LDS slots, so that phase C carries no
//            boundary-only arithmetic;
//   phase C  lane = level: each wave takes columns of the tile, reads its column from
//            LDS with unit stride, gets the vertical stencil neighbours (k-1, k+1) by
//            wavefront shuffles, the water table by a ballot, the sequential saturation
//            repair by a ballot-guarded lane-serial loop, and writes the closed state back
//            to the column's own LDS slots;
//   phase W  the tile is stored back lanes-over-columns.
//
// Arithmetic is the same device functions as the other kernels (bitwise identical).
#pragma once
#include "trm_kernels.hpp"

namespace trm {

constexpr int WAVE_TILE_COLS = 32;   // columns per workgroup tile (256-byte f64 row segments)
constexpr int WAVE_BLOCK = 512;      // 8 waves per workgroup
constexpr int WAVE_COL_SLOTS = 11;   // per-column LDS scalars

template <class NF, int LPC> TRM_DEV NF shfl_from(NF x, int src_k) { return __shfl(x, src_k, LPC); }
template <class NF, int LPC> TRM_DEV NF shfl_up1(NF x) { return __shfl_up(x, 1, LPC); }
template <class NF, int LPC> TRM_DEV NF shfl_dn1(NF x) { return __shfl_down(x, 1, LPC); }

template <int LPC> TRM_DEV unsigned long long group_mask(int lane) {
    if (LPC == 64) return ~0ull;
    return (lane & 32) ? 0xffffffff00000000ull : 0x00000000ffffffffull;
}

inline size_t wave_lds_bytes(int Nz, size_t esize) {
    const int RS = (Nz + 1) | 1;
    return ((size_t)6 * WAVE_TILE_COLS * RS + (size_t)WAVE_COL_SLOTS * WAVE_TILE_COLS) * esize;
}

template <class NF, bool RICHARDS, int HYD, int LPC>
__global__ void __launch_bounds__(WAVE_BLOCK) k_step_wave(View<NF> v, DevParams<NF> p, NF dt, int finalize, int write_kf) {
    extern __shared__ __align__(16) unsigned char trm_smem[];
    constexpr int TC = WAVE_TILE_COLS;
    constexpr int CPW = 64 / LPC;               // columns per wave at a time
    constexpr int NWAVES = WAVE_BLOCK / 64;
    const int Nz = v.Nz;
    const int RS = (Nz + 1) | 1;                // odd row stride >= Nz + 1
    const long P = v.pitch;
    const long col0 = (long)blockIdx.x * TC;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    NF* tU = reinterpret_cast<NF*>(trm_smem);
    NF* tS = tU + TC * RS;
    NF* tT = tS + TC * RS;
    NF* tL = tT + TC * RS;
    NF* tP = tL + TC * RS;
    NF* tK = tP + TC * RS;
    NF* cc = tK + TC * RS;       // per-column scalars, slot-major
    NF* hTb = cc;                // halo below the bottom cell: T, thermal conductivity, psi
    NF* hKb = cc + 1 * TC;
    NF* hPb = cc + 2 * TC;
    NF* hTt = cc + 3 * TC;       // halo above the top cell
    NF* hKt = cc + 4 * TC;
    NF* hPt = cc + 5 * TC;
    NF* fbU = cc + 6 * TC;       // compute_z_bcs! terms: bottom / top, energy / saturation
    NF* fbS = cc + 7 * TC;
    NF* ftU = cc + 8 * TC;
    NF* ftS = cc + 9 * TC;
    NF* cSn = cc + 10 * TC;      // surface excess water after the Euler update

    uint32_t viol = 0;
    bool bad = false;
    const bool need_kc = RICHARDS || write_kf;

    // ---- phase L: coalesced loads, transposed into LDS ---------------------------------------------
    for (int e = tid; e < TC * Nz; e += WAVE_BLOCK) {
        const int k = e / TC, c = e % TC;
        const long col = col0 + c;
        if (col < v.Nh) {
            const long g = (long)k * P + col;
            const int s = c * RS + k;
            tU[s] = v.U[g];
            tS[s] = v.sat[g];
            tT[s] = v.T[g];
            tL[s] = v.liq[g];
            if (RICHARDS) tP[s] = v.psi[g];
        }
    }
    __syncthreads();

    // ---- phase B: once-per-column work, lane = column -----------------------------------------------------
    if (tid < TC && col0 + tid < v.Nh) {
        const long i = col0 + tid;
        const int sb = tid * RS, st = tid * RS + (Nz - 1);
        // halos of the closure / prognostic fields below the bottom and above the top cell
        {
            NF lh = halo_bottom(v.bc.kind[3][0], bcval(v, 3, 0), i, tL[sb], v.g);
            NF sh = sat_halo<NF, RICHARDS>(v, p, 0, i, tS[sb]);
            hTb[tid] = halo_bottom(v.bc.kind[2][0], bcval(v, 2, 0), i, tT[sb], v.g);
            hKb[tid] = conductivity(p, fractions(p, sh, lh, viol));
            hPb[tid] = RICHARDS ? halo_bottom(v.bc.kind[4][0], bcval(v, 4, 0), i, tP[sb], v.g) : NF(0);
        }
        {
            NF lh = halo_top(v.bc.kind[3][1], bcval(v, 3, 1), i, tL[st], v.g);
            NF sh = sat_halo<NF, RICHARDS>(v, p, 1, i, tS[st]);
            hTt[tid] = halo_top(v.bc.kind[2][1], bcval(v, 2, 1), i, tT[st], v.g);
            hKt[tid] = conductivity(p, fractions(p, sh, lh, viol));
            hPt[tid] = RICHARDS ? halo_top(v.bc.kind[4][1], bcval(v, 4, 1), i, tP[st], v.g) : NF(0);
        }
        NF Sold = RICHARDS ? v.S[i] : NF(0);
        NF tU_term = NF(0), tS_term = NF(0);
        if (p.seb) {  // LandModel wires ground_heat_flux / -infiltration as top flux BCs (land_model.jl:56-61);
                      // k_surface produced them just before this launch
            tU_term = flux_term_top(v.ghf[i], v.g);
            if (RICHARDS) tS_term = flux_term_top(-v.infil[i], v.g);
            v.Ts[i] = v.Ts[i] + NF(0) * dt;  // explicit_step! of the zero-tendency prognostic skin_temperature
        } else {
            if (v.bc.kind[0][1] == 2) tU_term = flux_term_top(bcval(v, 0, 1)[i], v.g);
            if (RICHARDS && v.bc.kind[1][1] == 2) tS_term = flux_term_top(bcval(v, 1, 1)[i], v.g);
        }
        ftU[tid] = tU_term;
        ftS[tid] = tS_term;
        fbU[tid] = (v.bc.kind[0][0] == 2) ? flux_term_bottom(bcval(v, 0, 0)[i], v.g) : NF(0);
        fbS[tid] = (RICHARDS && v.bc.kind[1][0] == 2) ? flux_term_bottom(bcval(v, 1, 0)[i], v.g) : NF(0);
        // surface_excess_water: tendency min(0, S) once per column (SURVEY C-3), Euler update
        cSn[tid] = Sold + (NF(0) + jl_min(NF(0), Sold)) * dt;
    }
    __syncthreads();

    // ---- phase C: lane = level ---------------------------------------------------------------------------------
    const int k = lane % LPC;                  // this lane's level
    const int sub = lane / LPC;                // which of the wave's CPW columns
    const int kk = k < Nz ? k : Nz - 1;        // clamped index for the grid constants
    const bool lvl = k < Nz;
    const bool is_bot = k == 0, is_top = k == Nz - 1;
    const NF zC = v.zC[kk], psiz = v.psiz[kk], zFlo = v.zF[kk], dzc = v.dzc[kk], rdzc = v.rdzc[kk];
    const NF rdzf_lo = v.rdzf[kk], rdzf_hi = v.rdzf[kk + 1];
    const int ku = kk + 1 < Nz ? kk + 1 : kk, kd = kk > 0 ? kk - 1 : 0;
    const NF dzc_up = v.dzc[ku], rdzc_up = v.rdzc[ku], dzc_dn = v.dzc[kd], rdzc_dn = v.rdzc[kd];
    const NF zF_top = v.zF[Nz], dzc_top = v.dzc[Nz - 1];
    const unsigned long long gmask = group_mask<LPC>(lane);

    for (int cb = wave * CPW; cb < TC; cb += NWAVES * CPW) {
        const int c = cb + sub;
        const long i = col0 + c;
        const bool colok = i < v.Nh;           // uniform within the column's lane group
        const bool act = colok && lvl;
        const int s = c * RS + kk;

        uint32_t vi = 0;  // composition flags of this column's cells
        const NF U = tU[s], sat = tS[s], T = tT[s], liq = tL[s];
        const NF psi = RICHARDS ? tP[s] : NF(0);

        const Frac<NF> f = fractions(p, sat, liq, vi);
        const NF kap = conductivity(p, f);
        const NF Kc = need_kc ? conductivity_hydraulic<NF, HYD>(p, liq, f) : NF(0);

        // ---- heat: every lane forms its lower face, the top lane also the boundary face -------------------
        // (shuffles are executed by all lanes -- never inside a divergent select -- and patched after)
        const NF T_sh = shfl_up1<NF, LPC>(T), kap_sh = shfl_up1<NF, LPC>(kap);
        const NF T_m = is_bot ? hTb[c] : T_sh;
        const NF kap_m = is_bot ? hKb[c] : kap_sh;
        const NF qT_lo = -(NF(0.5) * (kap + kap_m)) * ((T - T_m) * rdzf_lo);
        const NF qT_top = -(NF(0.5) * (hKt[c] + kap)) * ((hTt[c] - T) * rdzf_hi);
        const NF qT_sh = shfl_dn1<NF, LPC>(qT_lo);
        const NF qT_hi = is_top ? qT_top : qT_sh;
        NF gU = NF(0) + (-((qT_hi - qT_lo) * rdzc));

        // ---- Richards: face conductivities (soil_hydrology.jl:145-163) and Darcy fluxes -------------------
        NF gS = NF(0), Kf_lo = NF(0);
        if (need_kc) {
            const NF Kc_m = shfl_up1<NF, LPC>(Kc);
            Kf_lo = (is_bot || is_top) ? Kc : jl_min(Kc, Kc_m);
        }
        if (RICHARDS) {
            const NF Kf_up = shfl_up1<NF, LPC>(Kf_lo), Kf_dn = shfl_dn1<NF, LPC>(Kf_lo), psi_sh = shfl_up1<NF, LPC>(psi);
            const NF Kf_m = is_bot ? NF(0) : Kf_up;   // halo face below: never written (0)
            const NF Kf_p = is_top ? Kc : Kf_dn;      // face Nz repeats the top cell's value
            const NF psi_m = is_bot ? hPb[c] : psi_sh;
            const NF g_lo = (psi - psi_m) * rdzf_lo;
            const NF Ks_lo = boolmul(g_lo < NF(0), jl_min(Kf_m, Kf_lo)) + boolmul(g_lo >= NF(0), jl_min(Kf_lo, Kf_p));
            const NF qW_lo = -Ks_lo * g_lo;
            // boundary face above the top cell (only the top lane's value is used)
            const NF g_t = (hPt[c] - psi) * rdzf_hi;
            const NF Ks_t = boolmul(g_t < NF(0), jl_min(Kf_lo, Kc)) + boolmul(g_t >= NF(0), jl_min(Kc, NF(0)));
            const NF qW_top = -Ks_t * g_t;
            const NF qW_sh = shfl_dn1<NF, LPC>(qW_lo);
            const NF qW_hi = is_top ? qW_top : qW_sh;
            const NF dtheta = -((qW_hi - qW_lo) * rdzc) + NF(0) + p.vwc_forcing;
            gS = NF(0) + div_const(dtheta, p.por, p.rpor);
        }
        // ---- compute_z_bcs!: flux BCs into the boundary cells ------------------------------------------------------
        if (is_bot) { gU += fbU[c]; if (RICHARDS) gS += fbS[c]; }
        if (is_top) { gU -= ftU[c]; if (RICHARDS) gS -= ftS[c]; }
        // ---- explicit Euler update ------------------------------------------------------------------------------------
        const NF Unew = U + gU * dt;
        bad = bad || (act && is_nan(Unew));
        NF snew = sat, Snew = NF(0), z0 = NF(0);
        if (RICHARDS) {
            snew = sat + gS * dt;
            bad = bad || (act && is_nan(snew));
            Snew = cSn[c];
            // ---- adjust_saturation_profile! (soil_hydrology.jl:185-219) ----------------------------------------
            // upward pass: sequential in k; the lane-serial loop runs only when some cell is oversaturated
            {
                const bool over = act && !is_top && !(jl_max(snew - NF(1), NF(0)) == NF(0));
                if (__ballot(over) == 0ull) {
                    if (!is_bot) snew = snew + NF(0);  // sat[k+1] += 0 * dz[k] / dz[k+1]
                } else {
                    NF carry = NF(0);
                    for (int q = 0; q < Nz - 1; ++q) {
                        NF cout = NF(0);
                        if (k == q) {
                            if (q > 0) snew = snew + carry;
                            NF e = jl_max(snew - NF(1), NF(0));
                            snew = snew - e;
                            cout = div_const(e * dzc, dzc_up, rdzc_up);
                        }
                        carry = shfl_from<NF, LPC>(cout, q);
                    }
                    if (is_top) snew = snew + carry;
                }
            }
            // downward pass
            {
                const bool under = act && !is_bot && !(jl_max(-snew, NF(0)) == NF(0));
                if (__ballot(under) == 0ull) {
                    if (!is_bot) snew = snew + NF(0);  // sat[k] += deficit (= 0)
                } else {
                    NF pend = NF(0);
                    for (int q = Nz - 1; q >= 1; --q) {
                        NF pout = NF(0);
                        if (k == q) {
                            snew = snew - pend;
                            NF d = jl_max(-snew, NF(0));
                            snew = snew + d;
                            pout = div_const(d * dzc, dzc_dn, rdzc_dn);
                        }
                        pend = shfl_from<NF, LPC>(pout, q);
                    }
                    if (is_bot) snew = snew - pend;
                }
            }
            // surface overflow joins surface_excess_water; bottom clamp
            NF e_top = NF(0);
            if (is_top) {
                e_top = jl_max(snew - NF(1), NF(0));
                snew = snew - e_top;
            }
            e_top = shfl_from<NF, LPC>(e_top, Nz - 1);
            Snew = Snew + e_top * dzc_top;
            if (is_bot) snew = jl_max(snew, NF(0));
            // ---- water table: lower face of the first unsaturated cell from the bottom ------------------
            const unsigned long long unsat = __ballot(act && snew < NF(1)) & gmask;
            const int first = unsat ? (__ffsll((long long)unsat) - 1) % LPC : -1;
            const NF z_first = shfl_from<NF, LPC>(zFlo, first >= 0 ? first : 0);
            z0 = first >= 0 ? z_first : zF_top;
        }
        // ---- closures: (U, sat) -> (T, liq, psi) ------------------------------------------------------------------
        NF ln, Tn;
        energy_closure(p, Unew, snew, ln, Tn, vi);
        const NF psin = RICHARDS ? pressure_head<NF, HYD>(p, snew, zC, psiz, z0) : NF(0);
        NF Kf_out = Kf_lo, Kf_out_top = Kc;
        if (finalize && write_kf) {
            const NF Kc_new = conductivity_hydraulic<NF, HYD>(p, ln, fractions(p, snew, ln, vi));
            const NF Kc_new_m = shfl_up1<NF, LPC>(Kc_new);
            Kf_out = (is_bot || is_top) ? Kc_new : jl_min(Kc_new, Kc_new_m);
            Kf_out_top = Kc_new;
        }
        if (act) {
            tU[s] = Unew;
            tT[s] = Tn;
            tL[s] = ln;
            if (RICHARDS) { tS[s] = snew; tP[s] = psin; }
            tK[s] = Kf_out;
            if (is_top) tK[s + 1] = Kf_out_top;
            if (RICHARDS && is_bot) { cSn[c] = Snew; v.wt[i] = z0; v.S[i] = Snew; }
            viol |= vi;  // padding lanes / columns past Nh carry no information
        }
    }
    __syncthreads();

    // ---- phase W: coalesced stores ---------------------------------------------------------------------------------------
    const bool store_kf = write_kf != 0;
    for (int e = tid; e < TC * (Nz + 1); e += WAVE_BLOCK) {
        const int kr = e / TC, c = e % TC;
        const long col = col0 + c;
        if (col < v.Nh) {
            const long g = (long)kr * P + col;
            const int s = c * RS + kr;
            if (kr < Nz) {
                v.U[g] = tU[s];
                v.T[g] = tT[s];
                v.liq[g] = tL[s];
                if (RICHARDS) { v.sat[g] = tS[s]; v.psi[g] = tP[s]; }
            }
            if (store_kf) v.Kf[g] = tK[s];
        }
    }
    viol |= bad ? 1u : 0u;
    if (viol) atomicOr(v.status, viol);
}

}  // namespace trm
