// trm_launch_wide_f32.hip -- k_column_wide instantiations, float (see trm_launch_wide.inl)
#include "trm_launch_wide.inl"
namespace trmh {
template struct WideLaunch<float>;
}  // namespace trmh
