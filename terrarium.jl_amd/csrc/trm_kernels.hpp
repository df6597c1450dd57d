// trm_kernels.hpp -- HIP kernels of the SoilModel / LandModel(bare ground) step.
//
// Data layout (HBM): one plain buffer per variable, [rows][pitch] with the
// column index fastest (SoA over columns); pitch = Nh rounded up to 64 so every
// level of every wavefront starts on a 512-byte (f64) boundary.  No halos in
// memory: halo values are pure functions of the edge cell and the boundary
// condition and are formed in registers.  Row 0 = bottom layer.
//
// Three implementations sit behind the same C ABI:
//   * k_step_wave (trm_kernel_wave.hpp) -- ONE launch per step, lane = soil level, a column per
//     (half-)wavefront, tiles transposed through LDS, shuffles for the vertical stencil.
//   * k_step_fused (below) -- ONE launch per step, lane = column, rolling stencil in registers.
//   * k_* (unfused)  -- one launch per reference kernel in the reference's order
//     (SURVEY 2.1), used for the stand-alone compute_* entry points and as the
//     A/B comparator for what fusion buys.
#pragma once
#include "trm_device.hpp"

namespace trm {

template <class NF> struct View {
    long Nh, pitch;
    int Nz;
    // 3-D
    NF *U, *sat, *T, *liq, *psi, *Kf, *G_U, *G_sat;
    // 2-D
    NF *S, *G_S, *wt, *Ts, *ghf, *swu, *lwu, *rnet, *Hs, *Hl, *evap, *infil, *runoff;
    const NF *Tair, *pres, *wind, *qair, *rain, *swd, *lwd;
    // grid (device arrays): zC[Nz], zF[Nz+1], dzc[Nz], rdzc[Nz], rdzf[Nz+1] (face f lies below cell f),
    // psiz[Nz] = zC - z_surface (elevation head)
    const NF *zC, *zF, *dzc, *rdzc, *rdzf, *psiz;
    BcGeom<NF> g;
    uint32_t* status;
    BcSet bc;
};

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

// ===========================================================================
// Unfused kernels: one per reference kernel (grid = (ceil(Nh/256), rows))
// ===========================================================================

// compute_hydraulics_kernel! (soil_hydrology.jl:145-163, 297-300)
template <class NF, int HYD> __global__ void k_hydraulics(View<NF> v, DevParams<NF> p) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    int k = blockIdx.y;
    if (i >= v.Nh) return;
    uint32_t viol = 0;
    const int Nz = v.Nz;
    auto Kc = [&](int kk) {
        NF s = v.sat[(long)kk * v.pitch + i], l = v.liq[(long)kk * v.pitch + i];
        return conductivity_hydraulic<NF, HYD>(p, l, fractions(p, s, l, viol));
    };
    if (k <= 0) {
        v.Kf[i] = Kc(0);
    } else if (k >= Nz - 1) {
        NF val = Kc(Nz - 1);
        v.Kf[(long)k * v.pitch + i] = val;
        v.Kf[(long)(k + 1) * v.pitch + i] = val;
    } else {
        v.Kf[(long)k * v.pitch + i] = jl_min(Kc(k), Kc(k - 1));
    }
    if (viol) atomicOr(v.status, viol);
}

// bare-ground evaporation + direct runoff + fused SEB kernel x2 (land_model.jl:79-88), one thread
// per column.  FROM_STATE: take the top-face hydraulic conductivity from (sat, liq) of the top cell
// (what compute_hydraulics! would store there, soil_hydrology.jl:156-158) instead of reading the
// hydraulic_conductivity field -- used in front of the fused step kernels, which do not
// materialise K before the surface processes run.
template <class NF, bool RICHARDS, int HYD, bool FROM_STATE> __global__ void k_surface(View<NF> v, DevParams<NF> p) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= v.Nh) return;
    const long top = (long)(v.Nz - 1) * v.pitch + i;
    SebIn<NF> in = {v.Tair[i], v.pres[i], v.wind[i], v.qair[i], v.rain[i], v.swd[i], v.lwd[i]};
    SebOut<NF> o;
    uint32_t viol = 0;
    NF Kf_top = FROM_STATE ? conductivity_hydraulic<NF, HYD>(p, v.liq[top], fractions(p, v.sat[top], v.liq[top], viol))
                           : v.Kf[top];
    surface_processes(p, in, v.Ts[i], v.T[top], v.sat[top], Kf_top, v.S[i], RICHARDS, v.dzc[v.Nz - 1], o);
    v.Ts[i] = o.Ts; v.ghf[i] = o.ghf; v.swu[i] = o.swu; v.lwu[i] = o.lwu; v.rnet[i] = o.rnet;
    v.Hs[i] = o.Hs; v.Hl[i] = o.Hl; v.evap[i] = o.evap; v.infil[i] = o.infil; v.runoff[i] = o.runoff;
}

// compute_tendencies_kernel! for SoilHydrology{RichardsEq} (soil_hydrology_rre.jl:150-162) and
// SoilEnergyBalance (soil_energy.jl:153-156), hydrology first (soil_coupled.jl:80-90).
template <class NF, bool RICHARDS> __global__ void k_tendencies(View<NF> v, DevParams<NF> p) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    int k = blockIdx.y;
    if (i >= v.Nh) return;
    const int Nz = v.Nz;
    const long P = v.pitch;
    uint32_t viol = 0;
    const long c = (long)k * P + i;
    if (RICHARDS) {
        // psi with halos
        NF psi0 = v.psi[c];
        NF psim = (k > 0) ? v.psi[c - P] : halo_bottom(v.bc.kind[4][0], bcval(v, 4, 0), i, psi0, v.g);
        NF psip = (k < Nz - 1) ? v.psi[c + P] : halo_top(v.bc.kind[4][1], bcval(v, 4, 1), i, psi0, v.g);
        // face conductivities k-1 .. k+2 (halo faces are never written: 0)
        NF Km = (k > 0) ? v.Kf[c - P] : NF(0);
        NF K0 = v.Kf[c];
        NF K1 = v.Kf[c + P];
        NF K2 = (k + 2 <= Nz) ? v.Kf[c + 2 * P] : NF(0);
        NF g_lo = (psi0 - psim) * v.rdzf[k];
        NF g_hi = (psip - psi0) * v.rdzf[k + 1];
        NF Klo = boolmul(g_lo < NF(0), jl_min(Km, K0)) + boolmul(g_lo >= NF(0), jl_min(K0, K1));
        NF Khi = boolmul(g_hi < NF(0), jl_min(K0, K1)) + boolmul(g_hi >= NF(0), jl_min(K1, K2));
        NF q_lo = -Klo * g_lo, q_hi = -Khi * g_hi;
        NF dtheta = -((q_hi - q_lo) * v.rdzc[k]) + NF(0) + p.vwc_forcing;
        v.G_sat[c] += div_const(dtheta, p.por, p.rpor);
        if (k == 0) v.G_S[i] += jl_min(NF(0), v.S[i]);
    }
    {
        NF T0 = v.T[c], s0 = v.sat[c], l0 = v.liq[c];
        NF Tm, sm, lm, Tp, sp, lp;
        if (k > 0) { Tm = v.T[c - P]; sm = v.sat[c - P]; lm = v.liq[c - P]; }
        else {
            Tm = halo_bottom(v.bc.kind[2][0], bcval(v, 2, 0), i, T0, v.g);
            lm = halo_bottom(v.bc.kind[3][0], bcval(v, 3, 0), i, l0, v.g);
            sm = sat_halo<NF, RICHARDS>(v, p, 0, i, s0);
        }
        if (k < Nz - 1) { Tp = v.T[c + P]; sp = v.sat[c + P]; lp = v.liq[c + P]; }
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
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    int k = blockIdx.y;
    if (i >= v.Nh) return;
    const int Nz = v.Nz;
    const long c = (long)k * v.pitch + i;
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

// hydrology closure!, one thread per column (soil_hydraulic_closures.jl:23-44):
// adjust_saturation_profile! (soil_hydrology.jl:185-219), compute_water_table!
// (soil_hydrology.jl:170-175) and, if WITH_PSI, saturation_to_pressure!.
template <class NF, bool WITH_PSI, int HYD> __global__ void k_closure_hydrology(View<NF> v, DevParams<NF> p) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= v.Nh) return;
    const int Nz = v.Nz;
    const long P = v.pitch;
    NF* s = v.sat + i;
    for (int k = 0; k <= Nz - 2; ++k) {
        NF sk = s[(long)k * P];
        NF e = jl_max(sk - NF(1), NF(0));
        s[(long)k * P] = sk - e;
        s[(long)(k + 1) * P] += div_const(e * v.dzc[k], v.dzc[k + 1], v.rdzc[k + 1]);
    }
    for (int k = Nz - 1; k >= 1; --k) {
        NF sk = s[(long)k * P];
        NF d = jl_max(-sk, NF(0));
        s[(long)k * P] = sk + d;
        s[(long)(k - 1) * P] -= div_const(d * v.dzc[k], v.dzc[k - 1], v.rdzc[k - 1]);
    }
    {
        NF st = s[(long)(Nz - 1) * P];
        NF e = jl_max(st - NF(1), NF(0));
        s[(long)(Nz - 1) * P] = st - e;
        v.S[i] += e * v.dzc[Nz - 1];
        s[0] = jl_max(s[0], NF(0));
    }
    int idx = -1;
    for (int k = 0; k < Nz; ++k)
        if (idx < 0 && s[(long)k * P] < NF(1)) idx = k;
    NF z0 = v.zF[idx >= 0 ? idx : Nz];
    v.wt[i] = z0;
    if (WITH_PSI)
        for (int k = 0; k < Nz; ++k) v.psi[(long)k * P + i] = pressure_head<NF, HYD>(p, s[(long)k * P], v.zC[k], v.psiz[k], z0);
}
// compute_water_table! alone (NoFlow initialisation, soil_hydrology.jl:113-117)
template <class NF> __global__ void k_water_table(View<NF> v) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= v.Nh) return;
    int idx = -1;
    for (int k = 0; k < v.Nz; ++k)
        if (idx < 0 && v.sat[(long)k * v.pitch + i] < NF(1)) idx = k;
    // the reference scan also visits the (never filled => 0) halo above the top cell, which
    // maps to the same surface node as "not found"
    v.wt[i] = v.zF[idx >= 0 ? idx : v.Nz];
}
// pressure_to_saturation_kernel! (soil_hydraulic_closures.jl:74-100)
template <class NF> __global__ void k_pressure_to_saturation(View<NF> v, DevParams<NF> p) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    int k = blockIdx.y;
    if (i >= v.Nh) return;
    const long c = (long)k * v.pitch + i;
    NF z = v.zC[k];
    NF psiz = v.psiz[k];
    NF psih = jl_max(NF(0), v.wt[i] - z);
    NF psim = v.psi[c] - psih - psiz;
    v.sat[c] = swrc_theta(p, psim, p.por) / p.por;
}
// energy_to_temperature_kernel! / temperature_to_energy_kernel! (soil_energy_closures.jl:163-171)
template <class NF> __global__ void k_closure_energy(View<NF> v, DevParams<NF> p) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    int k = blockIdx.y;
    if (i >= v.Nh) return;
    const long c = (long)k * v.pitch + i;
    uint32_t viol = 0;
    NF l, t;
    energy_closure(p, v.U[c], v.sat[c], l, t, viol);
    v.liq[c] = l;
    v.T[c] = t;
    if (viol) atomicOr(v.status, viol);
}
template <class NF> __global__ void k_invclosure_energy(View<NF> v, DevParams<NF> p) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    int k = blockIdx.y;
    if (i >= v.Nh) return;
    const long c = (long)k * v.pitch + i;
    uint32_t viol = 0;
    NF l, u;
    energy_invclosure(p, v.T[c], v.sat[c], l, u, viol);
    v.liq[c] = l;
    v.U[c] = u;
    if (viol) atomicOr(v.status, viol);
}
// Heun: state.tendencies .= (state.tendencies + stage.tendencies) / 2 (heun.jl:27-35)
template <class NF> __global__ void k_average(NF* a, const NF* b, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = (a[i] + b[i]) / NF(2);
}

// ===========================================================================
// Fused step, lane = column (TRM_KERNEL_FUSED_LANE): update_state! + explicit_step! + closure!
// (+ finalize) in ONE launch (forward_euler.jl:19-31).  64 columns per workgroup; every global
// access is a coalesced 512-byte row segment; the vertical stencil walks up the column in
// registers; the column's updated (U, sat) wait in LDS ([level][lane], conflict free) across the
// serial saturation repair and the water-table search and are closed to (T, liq, psi, K) on the
// way out.  The stored closure fields T / liq / psi are read, exactly as the reference does.
// ===========================================================================
template <class NF> struct Level { NF U, sat, T, liq, psi, kap, Kc; };
template <class NF> struct Raw { NF U, sat, T, liq, psi; };

constexpr int LANE_BLOCK = 64;

template <class NF, bool RICHARDS, int HYD>
__global__ void __launch_bounds__(LANE_BLOCK) k_step_fused(View<NF> v, DevParams<NF> p, NF dt, int finalize, int write_kf) {
    extern __shared__ __align__(16) unsigned char trm_smem[];
    constexpr int BLOCK = LANE_BLOCK;
    const long i = (long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= v.Nh) return;
    const int Nz = v.Nz;
    const long P = v.pitch;
    NF* lds_sat = reinterpret_cast<NF*>(trm_smem) + threadIdx.x;
    NF* lds_U = lds_sat + (long)Nz * BLOCK;
    uint32_t viol = 0;
    bool bad = false;
    const bool need_kc = RICHARDS || write_kf;

    auto load_raw = [&](int k) {
        Raw<NF> r;
        const long c = (long)k * P + i;
        r.U = v.U[c];
        r.sat = v.sat[c];
        r.T = v.T[c];
        r.liq = v.liq[c];
        r.psi = RICHARDS ? v.psi[c] : NF(0);
        return r;
    };
    auto make_level = [&](const Raw<NF>& r) {
        Level<NF> L;
        L.U = r.U; L.sat = r.sat; L.T = r.T; L.liq = r.liq; L.psi = r.psi;
        Frac<NF> f = fractions(p, L.sat, L.liq, viol);
        L.kap = conductivity(p, f);
        L.Kc = need_kc ? conductivity_hydraulic<NF, HYD>(p, L.liq, f) : NF(0);
        return L;
    };

    // ---- top flux BCs: LandModel wires ground_heat_flux / -infiltration (land_model.jl:56-61); the
    // surface processes that produce them ran in k_surface just before this launch ----------------
    NF S = RICHARDS ? v.S[i] : NF(0);
    NF top_U = NF(0), top_S = NF(0);   // flux-BC terms of the top cell (0 when there is none)
    if (p.seb) {
        top_U = flux_term_top(v.ghf[i], v.g);
        if (RICHARDS) top_S = flux_term_top(-v.infil[i], v.g);
        v.Ts[i] = v.Ts[i] + NF(0) * dt;  // explicit_step! of the zero-tendency prognostic skin_temperature
    } else {
        if (v.bc.kind[0][1] == 2) top_U = flux_term_top(bcval(v, 0, 1)[i], v.g);
        if (RICHARDS && v.bc.kind[1][1] == 2) top_S = flux_term_top(bcval(v, 1, 1)[i], v.g);
    }
    NF bot_U = NF(0), bot_S = NF(0);
    if (v.bc.kind[0][0] == 2) bot_U = flux_term_bottom(bcval(v, 0, 0)[i], v.g);
    if (RICHARDS && v.bc.kind[1][0] == 2) bot_S = flux_term_bottom(bcval(v, 1, 0)[i], v.g);

    // ---- upward sweep: tendencies, Euler update, upward pass of the saturation repair -------
    Level<NF> r0 = make_level(load_raw(0));
    Level<NF> r1 = make_level(load_raw(1));
    Level<NF> r2 = r1;
    if (Nz > 2) r2 = make_level(load_raw(2));
    Raw<NF> nxt = {};
    if (Nz > 3) nxt = load_raw(3);

    // face 0 (bottom boundary): halo cell from the BCs with level 0 as the edge
    NF qT_lo, qW_lo = NF(0);
    NF Kf_a = r0.Kc;                                              // face 0:  k <= 1 branch
    NF Kf_b = (Nz - 1 == 1) ? r1.Kc : jl_min(r1.Kc, r0.Kc);      // face 1
    {
        NF Th = halo_bottom(v.bc.kind[2][0], bcval(v, 2, 0), i, r0.T, v.g);
        NF lh = halo_bottom(v.bc.kind[3][0], bcval(v, 3, 0), i, r0.liq, v.g);
        NF sh = sat_halo<NF, RICHARDS>(v, p, 0, i, r0.sat);
        NF kh = conductivity(p, fractions(p, sh, lh, viol));
        qT_lo = -(NF(0.5) * (r0.kap + kh)) * ((r0.T - Th) * v.rdzf[0]);
        if (RICHARDS) {
            NF ph = halo_bottom(v.bc.kind[4][0], bcval(v, 4, 0), i, r0.psi, v.g);
            NF g = (r0.psi - ph) * v.rdzf[0];
            NF Ks = boolmul(g < NF(0), jl_min(NF(0), Kf_a)) + boolmul(g >= NF(0), jl_min(Kf_a, Kf_b));
            qW_lo = -Ks * g;
        }
    }
    NF carry = NF(0);
    NF Kc_prev_new = NF(0);  // NoFlow + finalize: rolling new-state cell conductivity
    for (int k = 0; k < Nz; ++k) {
        Raw<NF> pre = {};
        if (k + 4 < Nz) pre = load_raw(k + 4);
        // face k+2 conductivity (soil_hydrology.jl:145-163); faces beyond Nz are halo (0)
        NF Kf_c;
        {
            const int f = k + 2;
            if (f > Nz) Kf_c = NF(0);
            else if (f == Nz) Kf_c = r1.Kc;        // face Nz repeats the top cell's value
            else if (f == Nz - 1) Kf_c = r2.Kc;    // face Nz-1 = Kc(top cell)
            else Kf_c = jl_min(r2.Kc, r1.Kc);
        }
        // upper face k+1
        NF Th, kh, ph = NF(0);
        if (k + 1 < Nz) { Th = r1.T; kh = r1.kap; ph = r1.psi; }
        else {
            Th = halo_top(v.bc.kind[2][1], bcval(v, 2, 1), i, r0.T, v.g);
            NF lh = halo_top(v.bc.kind[3][1], bcval(v, 3, 1), i, r0.liq, v.g);
            NF sh = sat_halo<NF, RICHARDS>(v, p, 1, i, r0.sat);
            kh = conductivity(p, fractions(p, sh, lh, viol));
            if (RICHARDS) ph = halo_top(v.bc.kind[4][1], bcval(v, 4, 1), i, r0.psi, v.g);
        }
        NF qT_hi = -(NF(0.5) * (kh + r0.kap)) * ((Th - r0.T) * v.rdzf[k + 1]);
        NF gU = NF(0) + (-((qT_hi - qT_lo) * v.rdzc[k]));
        NF gS = NF(0), qW_hi = NF(0);
        if (RICHARDS) {
            NF g = (ph - r0.psi) * v.rdzf[k + 1];
            NF Ks = boolmul(g < NF(0), jl_min(Kf_a, Kf_b)) + boolmul(g >= NF(0), jl_min(Kf_b, Kf_c));
            qW_hi = -Ks * g;
            NF dtheta = -((qW_hi - qW_lo) * v.rdzc[k]) + NF(0) + p.vwc_forcing;
            gS = NF(0) + div_const(dtheta, p.por, p.rpor);
        }
        // compute_z_bcs!: flux BCs enter the boundary cells' tendencies
        if (k == 0) { gU += bot_U; if (RICHARDS) gS += bot_S; }
        if (k == Nz - 1) { gU -= top_U; if (RICHARDS) gS -= top_S; }
        NF Unew = r0.U + gU * dt;
        bad = bad || is_nan(Unew);
        const long c = (long)k * P + i;
        if (write_kf && !finalize) v.Kf[c] = Kf_a;
        if (RICHARDS) {
            NF snew = r0.sat + gS * dt;
            bad = bad || is_nan(snew);
            // upward pass of adjust_saturation_profile! (soil_hydrology.jl:192-199), fused
            if (k > 0) snew = snew + carry;
            if (k < Nz - 1) {
                NF e = jl_max(snew - NF(1), NF(0));
                snew = snew - e;
                carry = __any(e != NF(0)) ? div_const(e * v.dzc[k], v.dzc[k + 1], v.rdzc[k + 1]) : NF(0);
            }
            lds_sat[(long)k * BLOCK] = snew;
            lds_U[(long)k * BLOCK] = Unew;
        } else {
            // NoFlow: saturation is static, the column closes immediately
            NF ln, Tn;
            energy_closure(p, Unew, r0.sat, ln, Tn, viol);
            v.U[c] = Unew;
            v.liq[c] = ln;
            v.T[c] = Tn;
            if (finalize && write_kf) {
                NF Kc_new = conductivity_hydraulic<NF, HYD>(p, ln, fractions(p, r0.sat, ln, viol));
                NF face = (k == 0 || k == Nz - 1) ? Kc_new : jl_min(Kc_new, Kc_prev_new);
                v.Kf[c] = face;
                if (k == Nz - 1) v.Kf[c + P] = face;
                Kc_prev_new = Kc_new;
            }
        }
        // shift the window
        qT_lo = qT_hi;
        qW_lo = qW_hi;
        Kf_a = Kf_b;
        Kf_b = Kf_c;
        r0 = r1;
        r1 = r2;
        if (k + 3 < Nz) r2 = make_level(nxt);
        nxt = pre;
    }
    if (write_kf && !finalize) v.Kf[(long)Nz * P + i] = Kf_a;  // face Nz

    if (RICHARDS) {
        // surface_excess_water: tendency min(0, S) evaluated once per column (SURVEY C-3), Euler update
        S = S + (NF(0) + jl_min(NF(0), S)) * dt;
        // ---- downward pass of the repair (soil_hydrology.jl:202-208) + top overflow + water table
        NF pend = NF(0);
        int idx = -1;
        for (int k = Nz - 1; k >= 1; --k) {
            NF s = lds_sat[(long)k * BLOCK] - pend;
            NF d = jl_max(-s, NF(0));
            s = s + d;
            pend = __any(d != NF(0)) ? div_const(d * v.dzc[k], v.dzc[k - 1], v.rdzc[k - 1]) : NF(0);
            if (k == Nz - 1) {  // surface overflow joins surface_excess_water (soil_hydrology.jl:211-213)
                NF e = jl_max(s - NF(1), NF(0));
                s = s - e;
                S = S + e * v.dzc[Nz - 1];
            }
            lds_sat[(long)k * BLOCK] = s;
            if (s < NF(1)) idx = k;
        }
        {
            NF s = lds_sat[0] - pend;
            s = jl_max(s, NF(0));
            lds_sat[0] = s;
            if (s < NF(1)) idx = 0;
        }
        const NF z0 = v.zF[idx >= 0 ? idx : Nz];
        v.wt[i] = z0;
        v.S[i] = S;
        // ---- closing sweep: (U, sat) -> (T, liq, psi [, K]) and the stores ---------------------
        NF Kc_prev = NF(0), Tn = NF(0), face = NF(0), sn = NF(0);
        for (int k = 0; k < Nz; ++k) {
            const long c = (long)k * P + i;
            sn = lds_sat[(long)k * BLOCK];
            NF un = lds_U[(long)k * BLOCK];
            NF ln;
            energy_closure(p, un, sn, ln, Tn, viol);
            NF ps = pressure_head<NF, HYD>(p, sn, v.zC[k], v.psiz[k], z0);
            v.U[c] = un;
            v.sat[c] = sn;
            v.T[c] = Tn;
            v.liq[c] = ln;
            v.psi[c] = ps;
            if (finalize && write_kf) {
                NF Kc_new = conductivity_hydraulic<NF, HYD>(p, ln, fractions(p, sn, ln, viol));
                face = (k == 0 || k == Nz - 1) ? Kc_new : jl_min(Kc_new, Kc_prev);
                v.Kf[c] = face;
                if (k == Nz - 1) v.Kf[c + P] = face;
                Kc_prev = Kc_new;
            }
        }
    }
    viol |= bad ? 1u : 0u;
    if (viol) atomicOr(v.status, viol);
}

}  // namespace trm
