// trm_launch_deep.inl -- columns of 65 ... 128 levels, two levels per lane: the launches of k_column_deep (trm_column_deep.hpp).
// Included by trm_launch_deep_f64.hip / _f32.hip.
#include "trm_host.hpp"
#include "trm_column_deep.hpp"

namespace trmh {

template <class NF, bool RICH, int H, int PROG, bool GENERIC> static int launch_deep(trm_ctx* c, double dt, int finalize, int nsteps) {
    const LaunchArgs<NF>& la = launch_args<NF>(c);
    ColumnArgs<NF> a{};
    a.dt = (NF)dt;
    a.finalize = finalize;
    a.write_kf = (c->opt_write_kf || finalize) ? 1 : 0;
    a.nsteps = nsteps;
    a.bcT_bot_stage = la.w.bcT_bot;      // Heun: the stage's temperature boundary values (evaluated at t + dt)
    a.bcT_top_stage = la.w.bcT_top;
    if (PROG == PROG_HEUN && Policy<NF>::coupled(c)) {   // the stage's soil state is needed by the 0-D processes evaluated at the stage
        a.stage_sat = (NF*)c->stage.f[TRM_FIELD_SATURATION_WATER_ICE];
        a.stage_liq = (NF*)c->stage.f[TRM_FIELD_LIQUID_WATER_FRACTION];
        a.stage_T = (NF*)c->stage.f[TRM_FIELD_TEMPERATURE];
        a.stage_S = (NF*)c->stage.f[TRM_FIELD_SURFACE_EXCESS_WATER];
    }
    const dim3 grid((unsigned)((ncols(c) + (TRM_STEP_BLOCK / 64) - 1) / (TRM_STEP_BLOCK / 64)));
    if constexpr (GENERIC) hipLaunchKernelGGL((k_column_deep<NF, RICH, H, false, PROG, true>), grid, dim3(TRM_STEP_BLOCK), 0, c->stream, state_view<NF>(c), la.p, a, la.stage);
    else if (Policy<NF>::template derive_now<RICH>(c) == DERIVE_T_LIQ) hipLaunchKernelGGL((k_column_deep<NF, RICH, H, true, PROG>), grid, dim3(TRM_STEP_BLOCK), 0, c->stream, state_view<NF>(c), la.p, a, la.stage);
    else hipLaunchKernelGGL((k_column_deep<NF, RICH, H, false, PROG>), grid, dim3(TRM_STEP_BLOCK), 0, c->stream, state_view<NF>(c), la.p, a, la.stage);
    TRM_HIP(c, hipGetLastError());
    // (lanes per column: 64, two levels each; bits 25-26 the program, 27 the generic boundary kinds)
    c->last_program = program_id(TRM_PROGRAM_DEEP, H, 64, (!GENERIC && Policy<NF>::template derive_now<RICH>(c) == DERIVE_T_LIQ) ? DERIVE_T_LIQ : DERIVE_NONE, 0, 1, -1) | (PROG << 25) | ((GENERIC ? 1 : 0) << 27);
    return TRM_OK;
}
template <class NF, int PROG, bool GENERIC> static int deep_by_flow(trm_ctx* c, double dt, int finalize, int nsteps) {
    int rc = TRM_OK;
    if (Policy<NF>::richards(c)) { TRM_BY_HYD(c, rc = (launch_deep<NF, true, H, PROG, GENERIC>(c, dt, finalize, nsteps))); }
    else { TRM_BY_HYD(c, rc = (launch_deep<NF, false, H, PROG, GENERIC>(c, dt, finalize, nsteps))); }
    return rc;
}
template <class NF> int DeepLaunch<NF>::run(trm_ctx* c, int prog, bool generic, double dt, int finalize, int nsteps) {
    if (generic) {
        if (prog == PROG_EULER) return deep_by_flow<NF, PROG_EULER, true>(c, dt, finalize, nsteps);
        if (prog == PROG_HEUN) return deep_by_flow<NF, PROG_HEUN, true>(c, dt, finalize, nsteps);
        return fail(c, TRM_EINVAL, "k_column_deep: the generic boundary kinds run one step per launch");
    }
    switch (prog) {
        case PROG_EULER: return deep_by_flow<NF, PROG_EULER, false>(c, dt, finalize, nsteps);
        case PROG_HEUN: return deep_by_flow<NF, PROG_HEUN, false>(c, dt, finalize, nsteps);
        case PROG_MULTI: return deep_by_flow<NF, PROG_MULTI, false>(c, dt, finalize, nsteps);
        default: return fail(c, TRM_EINVAL, "k_column_deep: unknown program");
    }
}

}  // namespace trmh
