// trm_launch_column_sig_heun_f64_a.hip -- the one-launch Heun program with a compile-time boundary-condition signature (see trm_launch_column_sig_heun.inl)
#include "trm_launch_column_sig_heun.inl"
namespace trmh {
template struct ColumnSigHeunLaunch<double, true, 0>;
template struct ColumnSigHeunLaunch<double, true, BCSIG_T_TOP>;
}  // namespace trmh
