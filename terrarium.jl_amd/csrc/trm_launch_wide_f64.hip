// trm_launch_wide_f64.hip -- k_column_wide instantiations, double (see trm_launch_wide.inl)
#include "trm_launch_wide.inl"
namespace trmh {
template struct WideLaunch<double>;
}  // namespace trmh
