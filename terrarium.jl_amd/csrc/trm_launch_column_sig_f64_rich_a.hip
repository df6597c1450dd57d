// trm_launch_column_sig_f64_rich_a.hip -- the ForwardEuler column program with a compile-time boundary-condition signature (see trm_launch_column_sig.inl)
#include "trm_launch_column_sig.inl"
namespace trmh {
template struct ColumnSigLaunch<double, true, 0>;
template struct ColumnSigLaunch<double, true, BCSIG_T_TOP>;
}  // namespace trmh
