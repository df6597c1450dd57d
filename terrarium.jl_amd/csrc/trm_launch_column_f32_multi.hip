// trm_launch_column_f32_multi.hip -- k_column instantiations: float, PROG_MULTI (see trm_launch_column.inl)
#include "trm_launch_column.inl"
namespace trmh {
template struct ColumnLaunch<float, true, PROG_MULTI>;
template struct ColumnLaunch<float, false, PROG_MULTI>;
}  // namespace trmh
