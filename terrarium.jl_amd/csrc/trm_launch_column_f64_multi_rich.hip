// trm_launch_column_f64_multi_rich.hip -- k_column instantiations: double, PROG_MULTI (see trm_launch_column.inl)
#include "trm_launch_column.inl"
namespace trmh {
template struct ColumnLaunch<double, true, PROG_MULTI>;
}  // namespace trmh
