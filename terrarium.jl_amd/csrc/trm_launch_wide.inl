// trm_launch_wide.inl -- columns of 129 ... 256 levels, four levels per lane: the launches of k_column_wide (trm_column_wide.hpp).
// Included by trm_launch_wide_f64.hip / _f32.hip.
#include "trm_host.hpp"
#include "trm_column_wide.hpp"

namespace trmh {

template <class NF, bool RICH, int H, int PROG, bool GENERIC> static int launch_wide(trm_ctx* c, double dt, int finalize) {
    const LaunchArgs<NF>& la = launch_args<NF>(c);
    ColumnArgs<NF> a{};
    a.dt = (NF)dt;
    a.finalize = finalize;
    a.write_kf = (c->opt_write_kf || finalize) ? 1 : 0;
    a.nsteps = 1;
    a.bcT_bot_stage = la.w.bcT_bot;      // Heun: the stage's temperature boundary values (evaluated at t + dt)
    a.bcT_top_stage = la.w.bcT_top;
    if (PROG == PROG_HEUN && Policy<NF>::coupled(c)) {   // the stage's soil state is needed by the 0-D processes evaluated at the stage
        a.stage_sat = (NF*)c->stage.f[TRM_FIELD_SATURATION_WATER_ICE];
        a.stage_liq = (NF*)c->stage.f[TRM_FIELD_LIQUID_WATER_FRACTION];
        a.stage_T = (NF*)c->stage.f[TRM_FIELD_TEMPERATURE];
        a.stage_S = (NF*)c->stage.f[TRM_FIELD_SURFACE_EXCESS_WATER];
    }
    const dim3 grid((unsigned)((ncols(c) + (TRM_STEP_BLOCK / 64) - 1) / (TRM_STEP_BLOCK / 64)));
    hipLaunchKernelGGL((k_column_wide<NF, RICH, H, 4, PROG, GENERIC>), grid, dim3(TRM_STEP_BLOCK), 0, c->stream, state_view<NF>(c), la.p, a, la.stage);
    TRM_HIP(c, hipGetLastError());
    c->last_program = program_id(TRM_PROGRAM_WIDE, H, 64, DERIVE_NONE, 0, 1, -1) | (PROG << 25) | ((GENERIC ? 1 : 0) << 27);
    return TRM_OK;
}
template <class NF, int PROG, bool GENERIC> static int wide_by_flow(trm_ctx* c, double dt, int finalize) {
    int rc = TRM_OK;
    if (Policy<NF>::richards(c)) { TRM_BY_HYD(c, rc = (launch_wide<NF, true, H, PROG, GENERIC>(c, dt, finalize))); }
    else { TRM_BY_HYD(c, rc = (launch_wide<NF, false, H, PROG, GENERIC>(c, dt, finalize))); }
    return rc;
}
template <class NF> int WideLaunch<NF>::run(trm_ctx* c, int prog, bool generic, double dt, int finalize) {
    if (c->Nz > 256 || c->Nz <= 128) return fail(c, TRM_EINVAL, "k_column_wide serves columns of 129 ... 256 levels");
    if (prog == PROG_EULER) return generic ? wide_by_flow<NF, PROG_EULER, true>(c, dt, finalize) : wide_by_flow<NF, PROG_EULER, false>(c, dt, finalize);
    if (prog == PROG_HEUN) return generic ? wide_by_flow<NF, PROG_HEUN, true>(c, dt, finalize) : wide_by_flow<NF, PROG_HEUN, false>(c, dt, finalize);
    return fail(c, TRM_EINVAL, "k_column_wide: one step per launch (ForwardEuler or Heun)");
}

}  // namespace trmh
