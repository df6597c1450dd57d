// trm_launch_column_tail.hip -- k_column_tail (trm_column.hpp): the per-step ForwardEuler program of a bare-ground LandModel (fp64, Richards,
// the LandModel's boundary signature compiled in, per-column outputs staged) with the NEXT step's surface processes evaluated at its
// tail into the pending arrays (TRM_OPT_TAIL_SURFACE; surface_tail in trm_kernels.hpp).
#include "trm_host.hpp"

namespace trmh {

template <int H, int LPC> static int launch_column_tail(trm_ctx* c, double dt, int finalize) {
    using NF = double;
    using P = Policy<NF>;
    const LaunchArgs<NF>& la = launch_args<NF>(c);
    const View<NF>& v = la.state;
    if (!c->tail_counter || !c->pend.f[TRM_FIELD_GROUND_HEAT_FLUX] || !v.top_T)
        return fail(c, TRM_EINVAL, "k_column_tail: the context has no pending arrays / top-cell arrays");
    ColumnArgs<NF> a{};
    a.dt = (NF)dt;
    a.finalize = finalize;
    a.write_kf = (c->opt_write_kf || finalize) ? 1 : 0;
    a.nsteps = 1;
    a.bcT_bot_stage = la.w.bcT_bot;
    a.bcT_top_stage = la.w.bcT_top;
    TailArgs<NF> t{};
    t.counter = c->tail_counter;
    static const int fields[TAIL_COUNT] = {TRM_FIELD_SKIN_TEMPERATURE, TRM_FIELD_GROUND_HEAT_FLUX, TRM_FIELD_SURFACE_SHORTWAVE_UP, TRM_FIELD_SURFACE_LONGWAVE_UP,
                                           TRM_FIELD_SURFACE_NET_RADIATION, TRM_FIELD_SENSIBLE_HEAT_FLUX, TRM_FIELD_LATENT_HEAT_FLUX, TRM_FIELD_EVAPORATION_GROUND,
                                           TRM_FIELD_INFILTRATION, TRM_FIELD_SURFACE_RUNOFF};
    for (int n = 0; n < TAIL_COUNT; ++n) t.out[n] = (NF*)c->pend.f[fields[n]];
    const dim3 grid = column_grid(c, LPC), block(TRM_STEP_BLOCK);
    // one counter per 64-column cluster of the grid (trm_ctx::tail_clusters were allocated for the whole context)
    if ((long)((grid.x * (unsigned)(TRM_STEP_BLOCK / 64) * (64 / LPC) + 63) / 64) > c->tail_clusters) return fail(c, TRM_EINVAL, "k_column_tail: grid beyond the cluster counters");
    const int derive = P::derive_now<true>(c);
    const int scalar_in = derive == DERIVE_T_LIQ ? P::scalar_inputs_now<true>(c) : 1;
    if (derive == DERIVE_T_LIQ && scalar_in) hipLaunchKernelGGL((k_column_tail<NF, true, H, LPC, DERIVE_T_LIQ, true>), grid, block, 0, c->stream, v, la.p, a, t);
    else if (derive == DERIVE_T_LIQ) hipLaunchKernelGGL((k_column_tail<NF, true, H, LPC, DERIVE_T_LIQ, false>), grid, block, 0, c->stream, v, la.p, a, t);
    else if (derive == DERIVE_NONE) hipLaunchKernelGGL((k_column_tail<NF, true, H, LPC, DERIVE_NONE, true>), grid, block, 0, c->stream, v, la.p, a, t);
    else return fail(c, TRM_EINVAL, "k_column_tail: no instance for this derivation mode");
    TRM_HIP(c, hipGetLastError());
    return TRM_OK;
}

int TailLaunch::run(trm_ctx* c, double dt, int finalize) {
    using NF = double;
    const bool deep = c->Nz > 32;
    switch (Policy<NF>::hyd(c)) {
        case HYD_BC_LINEAR: return deep ? launch_column_tail<HYD_BC_LINEAR, 64>(c, dt, finalize) : launch_column_tail<HYD_BC_LINEAR, 32>(c, dt, finalize);
        case HYD_VG_N2: return deep ? launch_column_tail<HYD_VG_N2, 64>(c, dt, finalize) : launch_column_tail<HYD_VG_N2, 32>(c, dt, finalize);
        default: return fail(c, TRM_EINVAL, "k_column_tail: no instance for the generic hydraulics");
    }
}

}  // namespace trmh
