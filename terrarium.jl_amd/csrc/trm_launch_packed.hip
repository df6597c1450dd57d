// trm_launch_packed.hip -- fp32 with two columns per lane and packed math (trm_packed_f32.hpp): the launches of k_step_pk and
// of k_land_pk (the interleaved LandModel launches in fp32).
#include "trm_host.hpp"
#include "trm_packed_f32.hpp"

namespace trmh {

template <bool RICH, int LPC> static int launch_packed(trm_ctx* c, double dt, int finalize) {
    using NF = float;
    using P = Policy<float>;
    const LaunchArgs<NF>& la = launch_args<NF>(c);
    const int wkf = (c->opt_write_kf || finalize) ? 1 : 0;
    const long pairs = (ncols(c) + 1) / 2;
    const View<NF>& sv = state_view<NF>(c);
    const long waves = (pairs + (64 / LPC) - 1) / (64 / LPC);
    dim3 pg((unsigned)((waves * 64 + TRM_STEP_BLOCK - 1) / TRM_STEP_BLOCK));
    const int derive = P::derive_now<RICH>(c);
    const int staged = P::staged_now<RICH>(c, true);
    const dim3 blk(TRM_STEP_BLOCK);
    // (the boundary kinds compiled in for the signatures that have an instance: TRM_OPT_BC_SIGNATURE, trm_kernels.hpp BCSIG)
    const int sig = c->opt_bc_signature ? bc_signature_of(c) : -1;
#define TRM_LAUNCH_PK(HYDV)                                                                                                                        \
    do {                                                                                                                                           \
        if (derive == DERIVE_T_LIQ) hipLaunchKernelGGL((k_step_pk<RICH, LPC, HYDV, DERIVE_T_LIQ>), pg, blk, 0, c->stream, sv, la.p, (float)dt, finalize, wkf, staged); \
        else if (derive == DERIVE_LIQ && RICH && sig == BCSIG_LAND) hipLaunchKernelGGL((k_step_pk<RICH, LPC, HYDV, DERIVE_LIQ, RICH ? BCSIG_LAND : BCSIG_RUNTIME>), pg, blk, 0, c->stream, sv, la.p, (float)dt, finalize, wkf, staged); \
        else if (derive == DERIVE_LIQ && RICH && sig == BCSIG_T_TOP) hipLaunchKernelGGL((k_step_pk<RICH, LPC, HYDV, DERIVE_LIQ, RICH ? BCSIG_T_TOP : BCSIG_RUNTIME>), pg, blk, 0, c->stream, sv, la.p, (float)dt, finalize, wkf, staged); \
        else if (derive == DERIVE_LIQ) hipLaunchKernelGGL((k_step_pk<RICH, LPC, HYDV, DERIVE_LIQ>), pg, blk, 0, c->stream, sv, la.p, (float)dt, finalize, wkf, staged); \
        else if (derive == DERIVE_LIQ_PSI) hipLaunchKernelGGL((k_step_pk<RICH, LPC, HYDV, DERIVE_LIQ_PSI>), pg, blk, 0, c->stream, sv, la.p, (float)dt, finalize, wkf, staged); \
        else hipLaunchKernelGGL((k_step_pk<RICH, LPC, HYDV, DERIVE_NONE>), pg, blk, 0, c->stream, sv, la.p, (float)dt, finalize, wkf, staged);     \
    } while (0)
    if (P::hyd(c) == HYD_VG_N2) TRM_LAUNCH_PK(HYD_VG_N2);
    else TRM_LAUNCH_PK(HYD_BC_LINEAR);
#undef TRM_LAUNCH_PK
    TRM_HIP(c, hipGetLastError());
    {
        const bool sig_instance = derive == DERIVE_LIQ && RICH && (sig == BCSIG_LAND || sig == BCSIG_T_TOP);
        c->last_program = program_id(TRM_PROGRAM_PACKED_F32, P::hyd(c) == HYD_VG_N2 ? HYD_VG_N2 : HYD_BC_LINEAR, LPC, derive, staged, 1, sig_instance ? sig : -1);
    }
    return TRM_OK;
}
// k_step_pk_land: the packed LandModel step with the surface processes in the first workgroups of the launch
template <int LPC> static int launch_packed_land(trm_ctx* c, double dt, int finalize) {
    using P = Policy<float>;
    const LaunchArgs<float>& la = launch_args<float>(c);
    const View<float>& sv = la.state;
    if (!sv.top_T || !c->top_valid) return fail(c, TRM_EINVAL, "k_step_pk_land: the surface workgroups read the top-cell arrays, which are not current");
    if (int rc = front_epoch_next(c)) return rc;
    FrontArgs fa{};
    fa.gran = c->d_gran;
    fa.epoch = c->front_epoch;
    fa.tag_bias = c->debug_handoff_tag_bias;
    fa.chain_blocks = (int)((c->Nh + TRM_STEP_BLOCK - 1) / TRM_STEP_BLOCK);
    const int wkf = (c->opt_write_kf || finalize) ? 1 : 0;
    const long pairs = (c->Nh + 1) / 2;
    const long waves = (pairs + (64 / LPC) - 1) / (64 / LPC);
    const dim3 pg((unsigned)fa.chain_blocks + (unsigned)((waves * 64 + TRM_STEP_BLOCK - 1) / TRM_STEP_BLOCK)), blk(TRM_STEP_BLOCK);
    const int derive = P::derive_now<true>(c);
    const int staged = P::staged_now<true>(c, true);
    const bool vg = P::hyd(c) == HYD_VG_N2;
#define TRM_PK_LAND(HYDV, D) hipLaunchKernelGGL((k_step_pk_land<LPC, HYDV, D>), pg, blk, 0, c->stream, sv, la.p, (float)dt, finalize, wkf, staged, fa)
    if (derive == DERIVE_LIQ) { if (vg) TRM_PK_LAND(HYD_VG_N2, DERIVE_LIQ); else TRM_PK_LAND(HYD_BC_LINEAR, DERIVE_LIQ); }
    else if (derive == DERIVE_NONE) { if (vg) TRM_PK_LAND(HYD_VG_N2, DERIVE_NONE); else TRM_PK_LAND(HYD_BC_LINEAR, DERIVE_NONE); }
    else return fail(c, TRM_EINVAL, "k_step_pk_land: no instance for this derivation mode");
#undef TRM_PK_LAND
    TRM_HIP(c, hipGetLastError());
    c->last_program = program_id(TRM_PROGRAM_PACKED_LAND, vg ? HYD_VG_N2 : HYD_BC_LINEAR, LPC, derive, staged, 1, BCSIG_LAND);
    return TRM_OK;
}
int PackedLaunch::step_land(trm_ctx* c, double dt, int finalize) {
    return c->Nz > 32 ? launch_packed_land<64>(c, dt, finalize) : launch_packed_land<32>(c, dt, finalize);
}

int PackedLaunch::step(trm_ctx* c, double dt, int finalize) {
    const bool deep = c->Nz > 32;
    if (Policy<float>::richards(c)) return deep ? launch_packed<true, 64>(c, dt, finalize) : launch_packed<true, 32>(c, dt, finalize);
    return deep ? launch_packed<false, 64>(c, dt, finalize) : launch_packed<false, 32>(c, dt, finalize);
}

// columns of part `qcol` step; the surface processes of part `qsurf` run beside them for ITS next column step (k_land_pk)
template <int H, int LPC> static int launch_land_pk(trm_ctx* c, int qcol, int qsurf, double dt, int finalize, bool top_arrays) {
    const LaunchArgs<float>& la = launch_args<float>(c);
    const View<float>&vc = la.part[qcol], &vs = la.part[qsurf];
    if (top_arrays && !vs.top_T) return fail(c, TRM_EINVAL, "LandModel launch: the top-cell arrays were requested on a context that has none");
    const int wkf = (c->opt_write_kf || finalize) ? 1 : 0;
    const unsigned sblocks = (unsigned)((c->part_n[qsurf] + TRM_STEP_BLOCK - 1) / TRM_STEP_BLOCK);
    const dim3 block(TRM_STEP_BLOCK);
    const long pairs = (c->part_n[qcol] + 1) / 2;
    const long waves = (pairs + (64 / LPC) - 1) / (64 / LPC);
    const dim3 grid(sblocks + (unsigned)((waves * 64 + TRM_STEP_BLOCK - 1) / TRM_STEP_BLOCK));
    if (top_arrays) hipLaunchKernelGGL((k_land_pk<true, LPC, H, true>), grid, block, 0, c->stream, vc, la.p, (float)dt, finalize, wkf, vs, (int)sblocks);
    else hipLaunchKernelGGL((k_land_pk<true, LPC, H, false>), grid, block, 0, c->stream, vc, la.p, (float)dt, finalize, wkf, vs, (int)sblocks);
    TRM_HIP(c, hipGetLastError());
    c->last_program = program_id(TRM_PROGRAM_LAND_INTERLEAVED, H, LPC, DERIVE_NONE, 0, 1, -1);
    return TRM_OK;
}
template <> int LandLaunch<float>::run(trm_ctx* c, int qcol, int qsurf, double dt, int finalize, bool top_arrays) {
    const bool deep = c->Nz > 32;
    if (Policy<float>::hyd(c) == HYD_VG_N2) return deep ? launch_land_pk<HYD_VG_N2, 64>(c, qcol, qsurf, dt, finalize, top_arrays) : launch_land_pk<HYD_VG_N2, 32>(c, qcol, qsurf, dt, finalize, top_arrays);
    return deep ? launch_land_pk<HYD_BC_LINEAR, 64>(c, qcol, qsurf, dt, finalize, top_arrays) : launch_land_pk<HYD_BC_LINEAR, 32>(c, qcol, qsurf, dt, finalize, top_arrays);
}

}  // namespace trmh
