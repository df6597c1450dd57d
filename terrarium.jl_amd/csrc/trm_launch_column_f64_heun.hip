// trm_launch_column_f64_heun.hip -- k_column instantiations: double, PROG_HEUN (see trm_launch_column.inl)
#include "trm_launch_column.inl"
namespace trmh {
template struct ColumnLaunch<double, true, PROG_HEUN>;
template struct ColumnLaunch<double, false, PROG_HEUN>;
}  // namespace trmh
