// trm_launch_deep_f64.hip -- k_column_deep instantiations, double (see trm_launch_deep.inl)
#include "trm_launch_deep.inl"
namespace trmh {
template struct DeepLaunch<double>;
}  // namespace trmh
