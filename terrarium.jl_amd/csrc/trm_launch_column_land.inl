// trm_launch_column_land.inl -- the launch of k_column_land (trm_column.hpp): ONE launch per ForwardEuler step of a bare-ground LandModel
// (fp64, Richards, the LandModel's boundary signature compiled in) with the 0-D surface processes in its first workgroups
// (TRM_OPT_SURFACE_IN_LAUNCH).  Included by trm_launch_column_land_{bc,vg}.hip, one hydraulics instance each.
#include "trm_host.hpp"

namespace trmh {

template <int H, int LPC> static int launch_column_land(trm_ctx* c, double dt, int finalize, bool heun) {
    using NF = double;
    using P = Policy<NF>;
    const LaunchArgs<NF>& la = launch_args<NF>(c);
    const View<NF>& v = la.state;
    if (!v.top_T || !c->top_valid) return fail(c, TRM_EINVAL, "k_column_land: the surface workgroups read the top-cell arrays, which are not current");
    if (int rc = front_epoch_next(c)) return rc;
    ColumnArgs<NF> a{};
    a.dt = (NF)dt;
    a.finalize = finalize;
    a.write_kf = (c->opt_write_kf || finalize) ? 1 : 0;
    a.nsteps = 1;
    a.bcT_bot_stage = la.w.bcT_bot;
    a.bcT_top_stage = la.w.bcT_top;
    FrontArgs fa{};
    fa.gran = c->d_gran;
    fa.epoch = c->front_epoch;
    fa.tag_bias = c->debug_handoff_tag_bias;
    fa.chain_blocks = (int)((c->Nh + TRM_STEP_BLOCK - 1) / TRM_STEP_BLOCK);
    dim3 grid = column_grid(c, LPC);
    grid.x += (unsigned)fa.chain_blocks;
    const dim3 block(TRM_STEP_BLOCK);
    if (heun) {      // the one-launch Heun program: T / liq read as stored, direct stores, scalar inputs (as k_column<..., PROG_HEUN>)
        hipLaunchKernelGGL((k_column_land<NF, true, H, LPC, DERIVE_NONE, false, true, PROG_HEUN>), grid, block, 0, c->stream, v, la.p, a, fa);
        TRM_HIP(c, hipGetLastError());
        c->last_program = program_id(TRM_PROGRAM_COLUMN_LAND, H, LPC, DERIVE_NONE, 0, 1, BCSIG_LAND) | (PROG_HEUN << 25);
        return TRM_OK;
    }
    const int derive = P::derive_now<true>(c);
    int staged = derive == DERIVE_T_LIQ ? P::staged_now<true>(c) : 0, scalar_in = derive == DERIVE_T_LIQ ? P::scalar_inputs_now<true>(c) : 1;
    P::io_paths(true, staged, scalar_in);
#define TRM_LAND1(D, ST, SC) hipLaunchKernelGGL((k_column_land<NF, true, H, LPC, D, ST, SC>), grid, block, 0, c->stream, v, la.p, a, fa)
    if (derive == DERIVE_NONE) TRM_LAND1(DERIVE_NONE, false, true);
    else if (derive != DERIVE_T_LIQ) return fail(c, TRM_EINVAL, "k_column_land: no instance for this derivation mode");
    else if (staged && scalar_in) TRM_LAND1(DERIVE_T_LIQ, true, true);
    else if (staged) TRM_LAND1(DERIVE_T_LIQ, true, false);
    else TRM_LAND1(DERIVE_T_LIQ, false, true);
#undef TRM_LAND1
    TRM_HIP(c, hipGetLastError());
    c->last_program = program_id(TRM_PROGRAM_COLUMN_LAND, H, LPC, derive, staged, scalar_in, BCSIG_LAND);
    return TRM_OK;
}

template <int H> int FrontLaunch::run_hyd(trm_ctx* c, double dt, int finalize, bool heun) {
    return c->Nz > 32 ? launch_column_land<H, 64>(c, dt, finalize, heun) : launch_column_land<H, 32>(c, dt, finalize, heun);
}

}  // namespace trmh
