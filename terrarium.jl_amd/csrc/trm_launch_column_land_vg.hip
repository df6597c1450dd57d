// trm_launch_column_land_vg.hip -- k_column_land with van Genuchten (n = 2) + Mualem (see trm_launch_column_land.inl)
#include "trm_launch_column_land.inl"
namespace trmh {
template int FrontLaunch::run_hyd<HYD_VG_N2>(trm_ctx*, double, int, bool);
}  // namespace trmh
