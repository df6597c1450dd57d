// trm_launch_column_sig_f64_rich_c.hip -- the ForwardEuler column program with a compile-time boundary-condition signature (see trm_launch_column_sig.inl)
#include "trm_launch_column_sig.inl"
namespace trmh {
template struct ColumnSigLaunch<double, true, BCSIG_T_TOP | BCSIG_FS_TOP>;
}  // namespace trmh
