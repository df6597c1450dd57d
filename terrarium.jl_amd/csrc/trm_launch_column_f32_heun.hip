// trm_launch_column_f32_heun.hip -- k_column instantiations: float, PROG_HEUN (see trm_launch_column.inl)
#include "trm_launch_column.inl"
namespace trmh {
template struct ColumnLaunch<float, true, PROG_HEUN>;
template struct ColumnLaunch<float, false, PROG_HEUN>;
}  // namespace trmh
