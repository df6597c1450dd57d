// trm_launch_column_land_bc.hip -- k_column_land with the reference-default hydraulics (see trm_launch_column_land.inl)
#include "trm_launch_column_land.inl"
namespace trmh {
extern template int FrontLaunch::run_hyd<HYD_VG_N2>(trm_ctx*, double, int, bool);     // (trm_launch_column_land_vg.hip)
template int FrontLaunch::run_hyd<HYD_BC_LINEAR>(trm_ctx*, double, int, bool);
int FrontLaunch::run(trm_ctx* c, double dt, int finalize, bool heun) {
    switch (Policy<double>::hyd(c)) {
        case HYD_BC_LINEAR: return run_hyd<HYD_BC_LINEAR>(c, dt, finalize, heun);
        case HYD_VG_N2: return run_hyd<HYD_VG_N2>(c, dt, finalize, heun);
        default: return fail(c, TRM_EINVAL, "k_column_land: no instance for the generic hydraulics");
    }
}
}  // namespace trmh
