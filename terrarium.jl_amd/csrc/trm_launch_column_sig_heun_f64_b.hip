// trm_launch_column_sig_heun_f64_b.hip -- the one-launch Heun program with a compile-time boundary-condition signature (see trm_launch_column_sig_heun.inl)
#include "trm_launch_column_sig_heun.inl"
namespace trmh {
template struct ColumnSigHeunLaunch<double, true, BCSIG_T_TOP | BCSIG_FU_BOT>;
template struct ColumnSigHeunLaunch<double, true, BCSIG_LAND>;
template struct ColumnSigHeunLaunch<double, false, 0>;
template struct ColumnSigHeunLaunch<double, false, BCSIG_T_TOP>;
template struct ColumnSigHeunLaunch<double, false, BCSIG_T_TOP | BCSIG_FU_BOT>;
}  // namespace trmh
