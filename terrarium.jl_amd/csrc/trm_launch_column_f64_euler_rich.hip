// trm_launch_column_f64_euler_rich.hip -- k_column instantiations: double, PROG_EULER (see trm_launch_column.inl)
#include "trm_launch_column.inl"
namespace trmh {
template struct ColumnLaunch<double, true, PROG_EULER>;
}  // namespace trmh
