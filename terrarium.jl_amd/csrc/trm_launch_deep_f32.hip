// trm_launch_deep_f32.hip -- k_column_deep instantiations, float (see trm_launch_deep.inl)
#include "trm_launch_deep.inl"
namespace trmh {
template struct DeepLaunch<float>;
}  // namespace trmh
