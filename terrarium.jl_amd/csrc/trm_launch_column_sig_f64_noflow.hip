// trm_launch_column_sig_f64_noflow.hip -- the ForwardEuler column program with a compile-time boundary-condition signature (see trm_launch_column_sig.inl)
#include "trm_launch_column_sig.inl"
namespace trmh {
template struct ColumnSigLaunch<double, false, 0>;
template struct ColumnSigLaunch<double, false, BCSIG_T_TOP>;
template struct ColumnSigLaunch<double, false, BCSIG_T_TOP | BCSIG_FU_BOT>;
}  // namespace trmh
