// trm_launch_unfused.hip -- one launch per reference kernel, in the reference's order: the stand-alone
// compute_* / closure entry points of the C ABI, columns deeper than the fused kernels serve, and the bit-for-bit comparator of
// every fused path; plus LandModel's 0-D surface kernel (k_surface) and update_inputs! of the time series (k_interp_series).
#include "trm_host.hpp"

namespace trmh {

// update_inputs!(state, clock) for the time series sources: evaluates every series at `time` into the input
// field / boundary value array of field set `s` (the Heun stage has its own copies)
// Levels appended by trm_series_append travel on the side stream; a step that reads one of them first makes the context
// stream wait for the copy (steps that stay within the older levels run under it).
template <class NF> int Unfused<NF>::await_levels(trm_ctx* c, trm_ctx::Series& sr, int last_level) {
    if (sr.pending_from < 0 || last_level < sr.pending_from) return TRM_OK;
    TRM_HIP(c, hipStreamWaitEvent(c->stream, c->copy_done, 0));
    for (auto& o : c->series) o.pending_from = -1;     // (one event covers every copy issued so far)
    return TRM_OK;
}
template <class NF> int Unfused<NF>::update_inputs(trm_ctx* c, const FieldSet& s, double time) {
    if (c->series.empty()) return TRM_OK;
    const bool stage = &s == &c->stage;
    SeriesJobs<NF> jobs;
    int nj = 0;
    auto flush = [&]() -> int {
        if (nj == 0) return TRM_OK;
        hipLaunchKernelGGL(k_interp_series<NF>, dim3((unsigned)((ncols(c) + 255) / 256), (unsigned)nj), dim3(256), 0, c->stream, jobs, ncols(c));
        TRM_HIP(c, hipGetLastError());
        nj = 0;
        return TRM_OK;
    };
    for (auto& sr : c->series) {
        int n1, n2;
        double f, g;
        if (sr.trimmed && time < sr.trimmed_before)
            return fail(c, TRM_EINVAL, "a windowed time series was asked for a time before the levels it still holds (trm_series_trim_before released them): "
                                       "re-create the series from the record's head before stepping from an earlier clock");
        series_time_indices(sr.times, sr.indexing, time, n1, n2, f, g);
        if (int rw = await_levels(c, sr, std::max(n1, n2))) return rw;
        if (!sr.is_bc && stage && !c->has_stage) continue;   // (fused Heun: the stage's surface processes are never evaluated)
        NF* dst;
        if (sr.is_bc) {
            void*& slot = stage ? c->bc_value_stage[sr.var][sr.side] : c->bc_value[sr.var][sr.side];
            if (!slot) {
                TRM_HIP(c, hipMalloc(&slot, (size_t)c->Nh * sizeof(NF)));
                c->args_valid = false;
            }
            dst = (NF*)slot;
        } else {
            dst = (NF*)s.f[sr.field];
        }
        const NF* base = (const NF*)sr.d_values + first_col(c);   // (a pipeline part evaluates its own columns)
        dst += first_col(c);
        jobs.job[nj++] = SeriesJob<NF>{dst, base + sr.slot(n1) * c->Nh, base + sr.slot(n2) * c->Nh, f, g, sr.indexing == TRM_TIME_RASTER ? 1 : 0};
        if (nj == 16) { int rc = flush(); if (rc) return rc; }
    }
    return flush();
}
template <class NF> int Unfused<NF>::hydraulics(trm_ctx* c, const FieldSet& s) {
    const View<NF>& v = cached_view<NF>(c, s);
    const DevParams<NF>& p = launch_args<NF>(c).p;
    TRM_BY_HYD(c, hipLaunchKernelGGL((k_hydraulics<NF, H>), cell_grid(c, c->Nz), dim3(256), 0, c->stream, v, p));
    TRM_HIP(c, hipGetLastError());
    return TRM_OK;
}
template <class NF> int Unfused<NF>::surface(trm_ctx* c, const FieldSet& s, bool from_state) {
    const View<NF>& v = cached_view<NF>(c, s);
    const DevParams<NF>& p = launch_args<NF>(c).p;
    if (from_state && c->top_valid && c->d_top3 && &s == &c->state) {   // (the top-cell arrays: only where they exist)
        if (Policy<NF>::richards(c)) { TRM_BY_HYD(c, hipLaunchKernelGGL((k_surface<NF, true, H, true, true>), col_grid(c), dim3(256), 0, c->stream, v, p)); }
        else { TRM_BY_HYD(c, hipLaunchKernelGGL((k_surface<NF, false, H, true, true>), col_grid(c), dim3(256), 0, c->stream, v, p)); }
    } else if (from_state) {
        if (Policy<NF>::richards(c)) { TRM_BY_HYD(c, hipLaunchKernelGGL((k_surface<NF, true, H, true, false>), col_grid(c), dim3(256), 0, c->stream, v, p)); }
        else { TRM_BY_HYD(c, hipLaunchKernelGGL((k_surface<NF, false, H, true, false>), col_grid(c), dim3(256), 0, c->stream, v, p)); }
    } else {
        if (Policy<NF>::richards(c)) hipLaunchKernelGGL((k_surface<NF, true, HYD_GENERIC, false, false>), col_grid(c), dim3(256), 0, c->stream, v, p);
        else hipLaunchKernelGGL((k_surface<NF, false, HYD_GENERIC, false, false>), col_grid(c), dim3(256), 0, c->stream, v, p);
    }
    TRM_HIP(c, hipGetLastError());
    return TRM_OK;
}
template <class NF> int Unfused<NF>::compute_auxiliary(trm_ctx* c, const FieldSet& s) {
    int rc = hydraulics(c, s);
    if (rc) return rc;
    if (Policy<NF>::coupled(c)) rc = Veg<NF>::surface_veg(c, s, false, false, 0.0);
    else if (c->params.seb) rc = surface(c, s);
    return rc;
}
template <class NF> int Unfused<NF>::compute_tendencies(trm_ctx* c, const FieldSet& s) {
    const View<NF>& v = cached_view<NF>(c, s);
    const DevParams<NF>& p = launch_args<NF>(c).p;
    if (Policy<NF>::richards(c)) hipLaunchKernelGGL((k_tendencies<NF, true>), cell_grid(c, c->Nz), dim3(256), 0, c->stream, v, p);
    else hipLaunchKernelGGL((k_tendencies<NF, false>), cell_grid(c, c->Nz), dim3(256), 0, c->stream, v, p);
    TRM_HIP(c, hipGetLastError());
    if (Policy<NF>::coupled(c)) {   // surface hydrology (canopy water) and vegetation tendencies (land_model.jl:90-97)
        if (int rv = Veg<NF>::vegetation(c, s, VEG_TEND, 0.0, 1, 0)) return rv;
    }
    return TRM_OK;
}
template <class NF> int Unfused<NF>::reset_tendencies(trm_ctx* c, const FieldSet& s) {
    for (int f : {TRM_FIELD_TEND_INTERNAL_ENERGY, TRM_FIELD_TEND_SATURATION_WATER_ICE, TRM_FIELD_TEND_SURFACE_EXCESS_WATER,
                  TRM_FIELD_TEND_CARBON_VEGETATION, TRM_FIELD_TEND_VEGETATION_AREA_FRACTION, TRM_FIELD_TEND_CANOPY_WATER})
        TRM_HIP(c, hipMemsetAsync(s.f[f], 0, field_elems(c, f) * sizeof(NF), c->stream));
    return TRM_OK;
}
template <class NF> int Unfused<NF>::update_state(trm_ctx* c, const FieldSet& s, bool tendencies) {
    int rc = reset_tendencies(c, s);
    if (!rc) rc = compute_auxiliary(c, s);
    if (!rc && tendencies) rc = compute_tendencies(c, s);
    return rc;
}
template <class NF> int Unfused<NF>::explicit_step(trm_ctx* c, const FieldSet& s, double dt) {
    const View<NF>& v = cached_view<NF>(c, s);
    const DevParams<NF>& p = launch_args<NF>(c).p;
    if (Policy<NF>::richards(c)) hipLaunchKernelGGL((k_explicit_step<NF, true>), cell_grid(c, c->Nz), dim3(256), 0, c->stream, v, p, (NF)dt);
    else hipLaunchKernelGGL((k_explicit_step<NF, false>), cell_grid(c, c->Nz), dim3(256), 0, c->stream, v, p, (NF)dt);
    TRM_HIP(c, hipGetLastError());
    if (Policy<NF>::coupled(c)) {
        if (int rv = Veg<NF>::vegetation(c, s, VEG_EXPLICIT, dt, 1, 0)) return rv;
    }
    return TRM_OK;
}
// hydrology closure: adjust_saturation_profile! + compute_water_table! (+ saturation_to_pressure!)
template <class NF, bool PSI, bool ADJ, int H> static void launch_closure_hydrology(trm_ctx* c, View<NF> v, DevParams<NF> p) {
    if (c->Nz <= 32) hipLaunchKernelGGL((k_closure_hydrology_wave<NF, PSI, H, ADJ, 32>), wave_grid(c, 32), dim3(256), 0, c->stream, v, p);
    else if (c->Nz <= 64) hipLaunchKernelGGL((k_closure_hydrology_wave<NF, PSI, H, ADJ, 64>), wave_grid(c, 64), dim3(256), 0, c->stream, v, p);
    else hipLaunchKernelGGL((k_closure_hydrology_seq<NF, PSI, H, ADJ>), col_grid(c), dim3(256), 0, c->stream, v, p);
}
template <class NF> int Unfused<NF>::closure_hydrology(trm_ctx* c, const FieldSet& s, bool with_psi, bool with_adjust) {
    const View<NF>& v = cached_view<NF>(c, s);
    const DevParams<NF>& p = launch_args<NF>(c).p;
    if (with_psi) {
        TRM_BY_HYD(c, (launch_closure_hydrology<NF, true, true, H>(c, v, p)));
    } else if (with_adjust) {
        launch_closure_hydrology<NF, false, true, HYD_GENERIC>(c, v, p);
    } else {
        launch_closure_hydrology<NF, false, false, HYD_GENERIC>(c, v, p);
    }
    TRM_HIP(c, hipGetLastError());
    return TRM_OK;
}
template <class NF> int Unfused<NF>::closure(trm_ctx* c, const FieldSet& s) {
    const View<NF>& v = cached_view<NF>(c, s);
    const DevParams<NF>& p = launch_args<NF>(c).p;
    if (Policy<NF>::richards(c)) {
        int rc = closure_hydrology(c, s, true);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_closure_energy<NF>, cell_grid(c, c->Nz), dim3(256), 0, c->stream, v, p);
    TRM_HIP(c, hipGetLastError());
    return TRM_OK;
}
template <class NF> int Unfused<NF>::invclosure(trm_ctx* c, const FieldSet& s) {
    const View<NF>& v = cached_view<NF>(c, s);
    const DevParams<NF>& p = launch_args<NF>(c).p;
    if (Policy<NF>::richards(c)) {
        hipLaunchKernelGGL(k_pressure_to_saturation<NF>, cell_grid(c, c->Nz), dim3(256), 0, c->stream, v, p);
        TRM_HIP(c, hipGetLastError());
        int rc = closure_hydrology(c, s, false);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_invclosure_energy<NF>, cell_grid(c, c->Nz), dim3(256), 0, c->stream, v, p);
    TRM_HIP(c, hipGetLastError());
    return TRM_OK;
}
template <class NF> int Unfused<NF>::initialize(trm_ctx* c) {
    const View<NF>& v = cached_view<NF>(c, c->state);
    const DevParams<NF>& p = launch_args<NF>(c).p;
    if (Policy<NF>::richards(c)) {  // soil_hydrology_rre.jl:33-47
        int rc = closure_hydrology(c, c->state, true);
        if (!rc) rc = hydraulics(c, c->state);
        if (rc) return rc;
    } else {  // soil_hydrology.jl:113-117
        int rc = hydraulics(c, c->state);
        if (rc) return rc;
        rc = closure_hydrology(c, c->state, false, false);  // compute_water_table! only
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_invclosure_energy<NF>, cell_grid(c, c->Nz), dim3(256), 0, c->stream, v, p);  // soil_energy.jl:64-77
    TRM_HIP(c, hipGetLastError());
    return TRM_OK;
}
template <class NF> int Unfused<NF>::average(trm_ctx* c, int field) {
    long n = (long)field_elems(c, field);
    hipLaunchKernelGGL(k_average<NF>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream,
                       (NF*)c->state.f[field], (const NF*)c->stage.f[field], n);
    TRM_HIP(c, hipGetLastError());
    return TRM_OK;
}

template struct Unfused<double>;
template struct Unfused<float>;

}  // namespace trmh
