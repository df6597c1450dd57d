// trm_launch_generic.inl -- the step kernels that serve every boundary kind (Gradient conditions, Value conditions on liquid
// fraction / saturation / pressure head, a per-cell vwc_forcing field): k_step_wave (ForwardEuler, trm_kernels.hpp) and
// k_heun_generic (Heun in one launch, trm_column.hpp).  Included by trm_launch_generic_f64.hip / _f32.hip.
#include "trm_host.hpp"

namespace trmh {

template <class NF, bool RICH, int H, int LPC> static int launch_wave(trm_ctx* c, double dt, int finalize) {
    const LaunchArgs<NF>& la = launch_args<NF>(c);
    const View<NF>& v = state_view<NF>(c);
    const DevParams<NF>& p = la.p;
    const int wkf = (c->opt_write_kf || finalize) ? 1 : 0;
    hipLaunchKernelGGL((k_step_wave<NF, RICH, H, LPC>), column_grid(c, LPC), dim3(TRM_STEP_BLOCK), 0, c->stream, v, p, (NF)dt, finalize, wkf);
    TRM_HIP(c, hipGetLastError());
    c->last_program = program_id(TRM_PROGRAM_GENERIC_EULER, H, LPC, DERIVE_NONE, 0, 0, -1);
    return TRM_OK;
}
template <class NF> int GenericLaunch<NF>::step(trm_ctx* c, double dt, int finalize) {
    int rc = TRM_OK;
    const bool deep = c->Nz > 32;
    if (Policy<NF>::richards(c)) { TRM_BY_HYD(c, rc = deep ? (launch_wave<NF, true, H, 64>(c, dt, finalize)) : (launch_wave<NF, true, H, 32>(c, dt, finalize))); }
    else { TRM_BY_HYD(c, rc = deep ? (launch_wave<NF, false, H, 64>(c, dt, finalize)) : (launch_wave<NF, false, H, 32>(c, dt, finalize))); }
    return rc;
}

// Heun with the generic boundary kinds: k_heun_generic, one launch per step like k_column<PROG_HEUN>
template <class NF, bool RICH, int H, int LPC> static int launch_heun_generic(trm_ctx* c, double dt, int finalize) {
    const LaunchArgs<NF>& la = launch_args<NF>(c);
    ColumnArgs<NF> a{};
    a.dt = (NF)dt;
    a.finalize = finalize;
    a.write_kf = (c->opt_write_kf || finalize) ? 1 : 0;
    a.nsteps = 1;
    hipLaunchKernelGGL((k_heun_generic<NF, RICH, H, LPC>), column_grid(c, LPC), dim3(TRM_STEP_BLOCK), 0, c->stream, la.state, la.p, la.stage, a);
    TRM_HIP(c, hipGetLastError());
    c->last_program = program_id(TRM_PROGRAM_GENERIC_HEUN, H, LPC, DERIVE_NONE, 0, 0, -1);
    return TRM_OK;
}
template <class NF> int GenericLaunch<NF>::heun(trm_ctx* c, double dt, int finalize) {
    int rc = TRM_OK;
    const bool deep = c->Nz > 32;
    if (Policy<NF>::richards(c)) { TRM_BY_HYD(c, rc = deep ? (launch_heun_generic<NF, true, H, 64>(c, dt, finalize)) : (launch_heun_generic<NF, true, H, 32>(c, dt, finalize))); }
    else { TRM_BY_HYD(c, rc = deep ? (launch_heun_generic<NF, false, H, 64>(c, dt, finalize)) : (launch_heun_generic<NF, false, H, 32>(c, dt, finalize))); }
    return rc;
}

}  // namespace trmh
