// trm_launch_column.inl -- the launch of k_column<NF, RICH, ., ., ., PROG, ...> (trm_column.hpp) for one (precision, flow scheme,
// program): included by the trm_launch_column_*.hip files, each of which instantiates its share of ColumnLaunch<NF, RICH, PROG>.
#include "trm_host.hpp"

namespace trmh {

template <class NF, bool RICH, int H, int LPC, int PROG> static int launch_column(trm_ctx* c, double dt, int finalize, int nsteps) {
    using P = Policy<NF>;
    const LaunchArgs<NF>& la = launch_args<NF>(c);
    const View<NF>& v = state_view<NF>(c);
    const DevParams<NF>& p = la.p;
    ColumnArgs<NF> a;
    a.dt = (NF)dt;
    a.finalize = finalize;
    a.write_kf = (c->opt_write_kf || finalize) ? 1 : 0;
    a.nsteps = nsteps;
    a.bcT_bot_stage = la.w.bcT_bot;
    a.bcT_top_stage = la.w.bcT_top;
    a.series = (const SeriesTable<NF>*)c->d_series_table;
    a.series_rows = (const SeriesRow*)c->d_series_rows;
    a.nseries = (int)c->series.size();
    a.stage_sat = a.stage_liq = a.stage_T = a.stage_S = nullptr;
    if (PROG == PROG_HEUN && P::coupled(c)) {   // the stage's soil state is needed by the 0-D processes evaluated at the stage
        a.stage_sat = (NF*)c->stage.f[TRM_FIELD_SATURATION_WATER_ICE];
        a.stage_liq = (NF*)c->stage.f[TRM_FIELD_LIQUID_WATER_FRACTION];
        a.stage_T = (NF*)c->stage.f[TRM_FIELD_TEMPERATURE];
        a.stage_S = (NF*)c->stage.f[TRM_FIELD_SURFACE_EXCESS_WATER];
    }
    const dim3 grid = column_grid(c, LPC), block(TRM_STEP_BLOCK);
    const int derive = P::template derive_now<RICH>(c);     // (for this kernel: DERIVE_NONE or DERIVE_T_LIQ)
    if (derive != DERIVE_NONE && derive != DERIVE_T_LIQ) return fail(c, TRM_EINVAL, "k_column: no instance for this derivation mode");
    int pid = 0;    // TRM_INFO_LAST_PROGRAM of the instance that is launched below
    if constexpr (PROG == PROG_MULTI) {
        const bool series = !c->series.empty();
        pid = program_id(TRM_PROGRAM_COLUMN_MULTI, H, LPC, DERIVE_NONE, 0, 1, -1) | (c->params.seb ? 1 << 25 : 0) | (series ? 1 << 26 : 0);
        if (c->params.seb && series) hipLaunchKernelGGL((k_column<NF, RICH, H, LPC, DERIVE_NONE, PROG_MULTI, true, true>), grid, block, 0, c->stream, v, p, a);
        else if (c->params.seb) hipLaunchKernelGGL((k_column<NF, RICH, H, LPC, DERIVE_NONE, PROG_MULTI, true, false>), grid, block, 0, c->stream, v, p, a);
        else if (series) hipLaunchKernelGGL((k_column<NF, RICH, H, LPC, DERIVE_NONE, PROG_MULTI, false, true>), grid, block, 0, c->stream, v, p, a);
        else hipLaunchKernelGGL((k_column<NF, RICH, H, LPC, DERIVE_NONE, PROG_MULTI, false, false>), grid, block, 0, c->stream, v, p, a);
    } else if constexpr (PROG == PROG_EULER) {
        // the context's boundary kinds as a signature; the instantiated ones take the program with the kinds compiled in (fp64; with
        // the derivation of T / liq or without it)
        const int sig = (c->opt_bc_signature && H != HYD_GENERIC) ? bc_signature_of(c) : -1;
        int staged = derive == DERIVE_T_LIQ ? P::template staged_now<RICH>(c) : 0, scalar_in = derive == DERIVE_T_LIQ ? P::template scalar_inputs_now<RICH>(c) : 1;
        const bool has_instance = sig == 0 || sig == BCSIG_T_TOP || sig == (BCSIG_T_TOP | BCSIG_FU_BOT) || (RICH && (sig == BCSIG_LAND || sig == (BCSIG_T_TOP | BCSIG_FS_TOP)));
        P::io_paths(!has_instance || sig == BCSIG_LAND, staged, scalar_in);
        if (launch_by_signature<ColumnSigLaunch, NF, RICH>(sig, c, v, p, a, grid, block, LPC, derive, staged, scalar_in)) {
            // (trm_launch_column_sig.inl: without the derivation the signature instances store directly and take the scalar path)
            pid = derive == DERIVE_T_LIQ ? program_id(TRM_PROGRAM_COLUMN_EULER, H, LPC, DERIVE_T_LIQ, staged, scalar_in, sig) : program_id(TRM_PROGRAM_COLUMN_EULER, H, LPC, DERIVE_NONE, 0, 1, sig);
        }
        // with the derivation (every large or HBM-resident fp64 state): how the per-column outputs leave / inputs arrive
        else if (derive == DERIVE_T_LIQ) {
            pid = std::is_same<NF, double>::value ? program_id(TRM_PROGRAM_COLUMN_EULER, H, LPC, DERIVE_T_LIQ, staged, scalar_in, -1) : program_id(TRM_PROGRAM_COLUMN_EULER, H, LPC, DERIVE_T_LIQ, 0, 1, -1);
            if constexpr (!std::is_same<NF, double>::value) {
                // (fp32 off the packed kernel derives only on request: one instance)
                hipLaunchKernelGGL((k_column<NF, RICH, H, LPC, DERIVE_T_LIQ, PROG_EULER, false>), grid, block, 0, c->stream, v, p, a);
            } else {
                if (staged && scalar_in) hipLaunchKernelGGL((k_column<NF, RICH, H, LPC, DERIVE_T_LIQ, PROG_EULER, false, false, true, true>), grid, block, 0, c->stream, v, p, a);
                else if (staged) hipLaunchKernelGGL((k_column<NF, RICH, H, LPC, DERIVE_T_LIQ, PROG_EULER, false, false, true, false>), grid, block, 0, c->stream, v, p, a);
                else hipLaunchKernelGGL((k_column<NF, RICH, H, LPC, DERIVE_T_LIQ, PROG_EULER, false, false, false, true>), grid, block, 0, c->stream, v, p, a);     // (io_paths: never (0, 0))
            }
        }
        else { pid = program_id(TRM_PROGRAM_COLUMN_EULER, H, LPC, DERIVE_NONE, 0, 1, -1); hipLaunchKernelGGL((k_column<NF, RICH, H, LPC, DERIVE_NONE, PROG_EULER, false>), grid, block, 0, c->stream, v, p, a); }
    } else {
        // (Heun: the same signatures)
        const int hsig = (c->opt_bc_signature && H != HYD_GENERIC) ? bc_signature_of(c) : -1;
        const bool launched = launch_by_signature<ColumnSigHeunLaunch, NF, RICH>(hsig, c, v, p, a, grid, block, LPC);
        pid = program_id(TRM_PROGRAM_COLUMN_HEUN, H, LPC, DERIVE_NONE, 0, 1, launched ? hsig : -1);
        if (!launched) hipLaunchKernelGGL((k_column<NF, RICH, H, LPC, DERIVE_NONE, PROG, false>), grid, block, 0, c->stream, v, p, a);
    }
    TRM_HIP(c, hipGetLastError());
    c->last_program = pid;
    return TRM_OK;
}

template <class NF, bool RICH, int PROG> int ColumnLaunch<NF, RICH, PROG>::run(trm_ctx* c, double dt, int finalize, int nsteps) {
    int rc = TRM_OK;
    const bool deep = c->Nz > 32;
    TRM_BY_HYD(c, rc = deep ? (launch_column<NF, RICH, H, 64, PROG>(c, dt, finalize, nsteps)) : (launch_column<NF, RICH, H, 32, PROG>(c, dt, finalize, nsteps)));
    return rc;
}

}  // namespace trmh
