// trm_launch_column_sig_heun.inl -- k_column<NF, RICH, ., ., DERIVE_NONE, PROG_HEUN, ..., BCSIG>: the one-launch Heun step with the
// boundary-condition signature compiled in (trm_kernels.hpp: BCSIG); included by the trm_launch_column_sig_heun_*.hip files.
#include "trm_host.hpp"

namespace trmh {

template <class NF, bool RICH, int SIG, int H, int LPC>
static void launch_column_sig_heun(trm_ctx* c, const View<NF>& v, const DevParams<NF>& p, const ColumnArgs<NF>& a, dim3 grid, dim3 block) {
    hipLaunchKernelGGL((k_column<NF, RICH, H, LPC, DERIVE_NONE, PROG_HEUN, false, false, false, true, SIG>), grid, block, 0, c->stream, v, p, a);
}
template <class NF, bool RICH, int SIG>
void ColumnSigHeunLaunch<NF, RICH, SIG>::run(trm_ctx* c, const View<NF>& v, const DevParams<NF>& p, const ColumnArgs<NF>& a, dim3 grid, dim3 block, int lpc) {
    TRM_BY_COMPILED_HYD(c, (lpc == 64 ? (launch_column_sig_heun<NF, RICH, SIG, H, 64>(c, v, p, a, grid, block)) : (launch_column_sig_heun<NF, RICH, SIG, H, 32>(c, v, p, a, grid, block))));
}

}  // namespace trmh
