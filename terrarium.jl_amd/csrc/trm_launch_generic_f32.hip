// trm_launch_generic_f32.hip -- k_step_wave / k_heun_generic instantiations, float (see trm_launch_generic.inl)
#include "trm_launch_generic.inl"
namespace trmh {
template struct GenericLaunch<float>;
}  // namespace trmh
