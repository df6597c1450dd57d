// trm_launch_column_sig.inl -- k_column<NF, RICH, ., ., DERIVE_T_LIQ | DERIVE_NONE, PROG_EULER, ..., STAGED, SCALAR_IN, BCSIG> for ONE boundary-condition
// signature (trm_kernels.hpp: BCSIG): included by the trm_launch_column_sig_*.hip files, each of which instantiates its signatures.
#include "trm_host.hpp"

namespace trmh {

template <class NF, bool RICH, int SIG, int H, int LPC>
static void launch_column_sig(trm_ctx* c, const View<NF>& v, const DevParams<NF>& p, const ColumnArgs<NF>& a, dim3 grid, dim3 block, int derive, int staged, int scalar_in) {
    // (T and liq read as stored -- small grids, the vegetation-coupled LandModel, the first step after an upload: direct stores, scalar inputs)
    // (Policy::io_paths has reduced (staged, scalar_in) to the combinations instantiated here: (0, 1), (1, 0), and (1, 1) for the LandModel)
    if (derive != DERIVE_T_LIQ) hipLaunchKernelGGL((k_column<NF, RICH, H, LPC, DERIVE_NONE, PROG_EULER, false, false, false, true, SIG>), grid, block, 0, c->stream, v, p, a);
    else if (staged && !scalar_in) hipLaunchKernelGGL((k_column<NF, RICH, H, LPC, DERIVE_T_LIQ, PROG_EULER, false, false, true, false, SIG>), grid, block, 0, c->stream, v, p, a);
    else if (!staged) hipLaunchKernelGGL((k_column<NF, RICH, H, LPC, DERIVE_T_LIQ, PROG_EULER, false, false, false, true, SIG>), grid, block, 0, c->stream, v, p, a);
    else if constexpr (SIG == BCSIG_LAND) hipLaunchKernelGGL((k_column<NF, RICH, H, LPC, DERIVE_T_LIQ, PROG_EULER, false, false, true, true, SIG>), grid, block, 0, c->stream, v, p, a);
}

template <class NF, bool RICH, int SIG>
void ColumnSigLaunch<NF, RICH, SIG>::run(trm_ctx* c, const View<NF>& v, const DevParams<NF>& p, const ColumnArgs<NF>& a, dim3 grid, dim3 block, int lpc, int derive, int staged, int scalar_in) {
    TRM_BY_COMPILED_HYD(c, (lpc == 64 ? (launch_column_sig<NF, RICH, SIG, H, 64>(c, v, p, a, grid, block, derive, staged, scalar_in))
                             : (launch_column_sig<NF, RICH, SIG, H, 32>(c, v, p, a, grid, block, derive, staged, scalar_in))));
}

}  // namespace trmh
