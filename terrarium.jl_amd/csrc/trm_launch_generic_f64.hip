// trm_launch_generic_f64.hip -- k_step_wave / k_heun_generic instantiations, double (see trm_launch_generic.inl)
#include "trm_launch_generic.inl"
namespace trmh {
template struct GenericLaunch<double>;
}  // namespace trmh
