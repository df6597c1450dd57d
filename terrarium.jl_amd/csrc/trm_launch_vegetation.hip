// trm_launch_vegetation.hip -- the launches of the vegetation kernels (trm_vegetation.hpp): the standalone VegetationModel
// (k_vegetation), the 0-D part of the vegetation-coupled LandModel (k_surface_veg), plant available water, and the averaging
// step of the coupled Heun (k_heun_average_0d).
#include "trm_host.hpp"

namespace trmh {

// the 0-D part of the coupled LandModel's compute_auxiliary! (+ tendencies and explicit step of the 0-D prognostics)
template <class NF> int Veg<NF>::surface_veg(trm_ctx* c, const FieldSet& s, bool from_state, bool advance, double dt, bool store_paw) {
    SurfaceVegArgs<NF> a{};
    a.dt = (NF)dt;
    a.richards = Policy<NF>::richards(c) ? 1 : 0;
    a.from_state = from_state ? 1 : 0;
    a.top_arrays = (from_state && c->top_valid && c->d_top3 && &s == &c->state) ? 1 : 0;
    a.advance = advance ? 1 : 0;
    a.store_paw = store_paw ? 1 : 0;
    return surface_veg_launch(c, cached_view<NF>(c, s), Policy<NF>::veg_view(c, s), a);
}
template <class NF> int Veg<NF>::surface_veg_launch(trm_ctx* c, const View<NF>& v, const VegView<NF>& vv, const SurfaceVegArgs<NF>& a) {
    if (a.top_arrays && !v.top_T) return fail(c, TRM_EINVAL, "k_surface_veg: the top-cell arrays were requested on a context that has none");
    const DevParams<NF>& p = launch_args<NF>(c).p;
    const VegDev<NF> vp = Policy<NF>::veg_dev(c);
    const dim3 blocks((unsigned)((ncols(c) + 63) / 64));   // 64 columns per 256-thread workgroup
    // (the kernel is bound by cold instruction fetch: the hydraulics of the top-face conductivity are compiled in)
    if (c->Nzp == 32) { TRM_BY_HYD(c, hipLaunchKernelGGL((k_surface_veg<NF, 32, H>), blocks, dim3(256), 0, c->stream, v, p, vv, vp, a)); }
    else if (c->Nzp == 64) { TRM_BY_HYD(c, hipLaunchKernelGGL((k_surface_veg<NF, 64, H>), blocks, dim3(256), 0, c->stream, v, p, vv, vp, a)); }
    else hipLaunchKernelGGL((k_surface_veg<NF, 0, HYD_GENERIC>), col_grid(c), dim3(256), 0, c->stream, v, p, vv, vp, a);
    TRM_HIP(c, hipGetLastError());
    return TRM_OK;
}
// k_vegetation<MODE> on the vegetation fields of field set `s`
template <class NF> int Veg<NF>::vegetation(trm_ctx* c, const FieldSet& s, int mode, double dt, int nsteps, int finalize) {
    const VegView<NF> vv = Policy<NF>::veg_view(c, s);
    const VegDev<NF> vp = Policy<NF>::veg_dev(c);
#define TRM_VEG(MODE) hipLaunchKernelGGL((k_vegetation<NF, MODE>), col_grid(c), dim3(256), 0, c->stream, vv, vp, (NF)dt, nsteps, finalize)
    switch (mode) {
        case VEG_AUX: TRM_VEG(VEG_AUX); break;
        case VEG_TEND: TRM_VEG(VEG_TEND); break;
        case VEG_UPDATE: TRM_VEG(VEG_UPDATE); break;
        case VEG_EXPLICIT: TRM_VEG(VEG_EXPLICIT); break;
        case VEG_EULER: TRM_VEG(VEG_EULER); break;
        case VEG_HEUN: TRM_VEG(VEG_HEUN); break;
        default: return fail(c, TRM_EINVAL, "k_vegetation: unknown mode");
    }
#undef TRM_VEG
    TRM_HIP(c, hipGetLastError());
    return TRM_OK;
}
template <class NF> int Veg<NF>::plant_available_water(trm_ctx* c, const FieldSet& s, bool store_paw) {
    const DevParams<NF>& p = launch_args<NF>(c).p;
    const NF *sat = (const NF*)s.f[TRM_FIELD_SATURATION_WATER_ICE], *liq = (const NF*)s.f[TRM_FIELD_LIQUID_WATER_FRACTION];
    NF *paw = (NF*)s.f[TRM_FIELD_PLANT_AVAILABLE_WATER], *smlf = (NF*)s.f[TRM_FIELD_SOIL_MOISTURE_LIMITING_FACTOR];
    const NF *rootf = (const NF*)c->d_rootf, *dzc = (const NF*)c->d_dzc, *rdzc = (const NF*)c->d_rdzc;
    const dim3 blocks((unsigned)((c->Nh + 63) / 64));
    if (c->Nzp == 32)
        hipLaunchKernelGGL((k_plant_available_water_block<NF, 32>), blocks, dim3(256), 0, c->stream, sat, liq, rootf, dzc, rdzc, store_paw ? paw : nullptr, smlf,
                           c->Nh, c->Nz, p.por, Policy<NF>::veg_dev(c));
    else if (c->Nzp == 64)
        hipLaunchKernelGGL((k_plant_available_water_block<NF, 64>), blocks, dim3(256), 0, c->stream, sat, liq, rootf, dzc, rdzc, store_paw ? paw : nullptr, smlf,
                           c->Nh, c->Nz, p.por, Policy<NF>::veg_dev(c));
    else
        hipLaunchKernelGGL((k_plant_available_water<NF>), col_grid(c), dim3(256), 0, c->stream, sat, liq, (const NF*)c->state.f[TRM_FIELD_ROOT_FRACTION], paw, smlf,
                           c->Nh, c->Nz, c->Nzp, p.por, Policy<NF>::veg_dev(c), dzc);
    TRM_HIP(c, hipGetLastError());
    return TRM_OK;
}
template <class NF> int Veg<NF>::heun_average_0d(trm_ctx* c, const VegView<NF>& vs, const VegView<NF>& vg, double dt) {
    hipLaunchKernelGGL((k_heun_average_0d<NF>), col_grid(c), dim3(256), 0, c->stream, vs, vg, (NF)dt, c->Nh);
    TRM_HIP(c, hipGetLastError());
    return TRM_OK;
}

template struct Veg<double>;
template struct Veg<float>;

}  // namespace trmh
