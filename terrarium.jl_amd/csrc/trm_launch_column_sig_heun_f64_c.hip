// trm_launch_column_sig_heun_f64_c.hip -- the one-launch Heun program with a compile-time boundary-condition signature (see trm_launch_column_sig_heun.inl)
#include "trm_launch_column_sig_heun.inl"
namespace trmh {
template struct ColumnSigHeunLaunch<double, true, BCSIG_T_TOP | BCSIG_FS_TOP>;
}  // namespace trmh
