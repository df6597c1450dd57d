// trm_launch_land.hip -- the interleaved LandModel launches in fp64 (TRM_OPT_PIPELINE_PARTS = 1): k_land_euler (trm_column.hpp)
// steps the soil columns of one half of the context and evaluates the 0-D surface processes of the other half.
#include "trm_host.hpp"

namespace trmh {

template <int H, int LPC> static int launch_land(trm_ctx* c, int qcol, int qsurf, double dt, int finalize, bool top_arrays) {
    using NF = double;
    constexpr bool RICH = true;
    const LaunchArgs<NF>& la = launch_args<NF>(c);
    const View<NF>&vc = la.part[qcol], &vs = la.part[qsurf];
    if (top_arrays && !vs.top_T) return fail(c, TRM_EINVAL, "LandModel launch: the top-cell arrays were requested on a context that has none");
    const int wkf = (c->opt_write_kf || finalize) ? 1 : 0;
    const unsigned sblocks = (unsigned)((c->part_n[qsurf] + TRM_STEP_BLOCK - 1) / TRM_STEP_BLOCK);
    const dim3 block(TRM_STEP_BLOCK);
    ColumnArgs<NF> a{};
    a.dt = (NF)dt;
    a.finalize = finalize;
    a.write_kf = wkf;
    a.nsteps = 1;
    a.bcT_bot_stage = la.w.bcT_bot;
    a.bcT_top_stage = la.w.bcT_top;
    const long waves = (c->part_n[qcol] + (64 / LPC) - 1) / (64 / LPC);
    const dim3 grid(sblocks + (unsigned)((waves * 64 + TRM_STEP_BLOCK - 1) / TRM_STEP_BLOCK));
    const bool derive = Policy<NF>::derive_now<RICH>(c) == DERIVE_T_LIQ;
#define TRM_LAND(D, T) hipLaunchKernelGGL((k_land_euler<NF, RICH, H, LPC, D, T>), grid, block, 0, c->stream, vc, la.p, a, vs, (int)sblocks)
    if (derive) { if (top_arrays) TRM_LAND(DERIVE_T_LIQ, true); else TRM_LAND(DERIVE_T_LIQ, false); }
    else { if (top_arrays) TRM_LAND(DERIVE_NONE, true); else TRM_LAND(DERIVE_NONE, false); }
#undef TRM_LAND
    TRM_HIP(c, hipGetLastError());
    c->last_program = program_id(TRM_PROGRAM_LAND_INTERLEAVED, H, LPC, derive ? DERIVE_T_LIQ : DERIVE_NONE, 0, 1, -1);
    return TRM_OK;
}
template <> int LandLaunch<double>::run(trm_ctx* c, int qcol, int qsurf, double dt, int finalize, bool top_arrays) {
    using NF = double;
    int rc = TRM_OK;
    const bool deep = c->Nz > 32;
    TRM_BY_HYD(c, rc = deep ? (launch_land<H, 64>(c, qcol, qsurf, dt, finalize, top_arrays)) : (launch_land<H, 32>(c, qcol, qsurf, dt, finalize, top_arrays)));
    return rc;
}

}  // namespace trmh
