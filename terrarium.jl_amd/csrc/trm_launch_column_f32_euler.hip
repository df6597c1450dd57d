// trm_launch_column_f32_euler.hip -- k_column instantiations: float, PROG_EULER (see trm_launch_column.inl)
#include "trm_launch_column.inl"
namespace trmh {
template struct ColumnLaunch<float, true, PROG_EULER>;
template struct ColumnLaunch<float, false, PROG_EULER>;
}  // namespace trmh
