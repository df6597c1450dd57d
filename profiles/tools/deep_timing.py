"""Per-cell time of the fused Euler step on deep columns (two levels per lane, Nz = 65 ... 128) against Nz = 64 (one level per
lane) and against the reference-order kernels: python profiles/tools/deep_timing.py"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
import workloads as W
mask = sys.argv[1] if len(sys.argv) > 1 else "N145"   # N72: 14 017 columns -- every state of the comparison fits the Infinity Cache
lat, lon = W.columns_from_mask(mask)
out = {"mask": mask, "columns": int(lat.size)}
for config in ("heat", "richards"):
    for Nz, kernel in ((64, "fused"), (100, "fused"), (128, "fused"), (100, "unfused")):
        w = W.make_workload(config, lat, lon, Nz)
        d = W.setup_device(w)
        d.set_option("step_kernel", kernel)
        d.step(w["dt"], 10, finalize=False)
        d.save_state()
        ts = []
        for _ in range(5):
            d.restore_state()
            d.step_timed(w["dt"], 50, finalize=False)
            d.restore_state()
            ts.append(d.step_timed(w["dt"], 50, finalize=False) * 1e3 / 50)
        us = float(np.median(ts))
        out[f"{config}_Nz{Nz}_{kernel}"] = {"us_per_step": round(us, 2), "ps_per_cell": round(us * 1e6 / (lat.size * Nz), 2), "status": d.status()}
        d.close()
# Nz = 100: the resident multi-step program (the library default for nsteps > 1) and Heun in one launch against the staged Heun
for config in ("heat", "richards"):
    w = W.make_workload(config, lat, lon, 100)
    for name, spl, kernel, heun in (("multistep", 0, "fused", False), ("heun_fused", 1, "fused", True), ("heun_unfused", 1, "unfused", True)):
        d = W.setup_device(w, steps_per_launch=spl)
        d.set_option("step_kernel", kernel)
        step = d.step_heun_timed if heun else d.step_timed
        d.step(w["dt"], 10, finalize=False)
        d.save_state()
        ts = []
        for _ in range(5):
            d.restore_state()
            step(w["dt"], 50, finalize=False)
            d.restore_state()
            ts.append(step(w["dt"], 50, finalize=False) * 1e3 / 50)
        out[f"{config}_Nz100_{name}"] = {"us_per_step": round(float(np.median(ts)), 2), "status": d.status()}
        d.close()
# round 4: the coverage holes closed -- Heun with the generic boundary kinds (FreeDrainage: soil_model_bcs.jl:40) on Nz = 100, the
# vegetation-coupled LandModel on Nz = 100, columns of 129 ... 256 levels (four levels per lane) -- each against the reference-order kernels
def timed(d, step, dt, n=20):
    d.save_state()
    ts = []
    for _ in range(5):
        d.restore_state()
        step(dt, n, finalize=False)
        d.restore_state()
        ts.append(step(dt, n, finalize=False) * 1e3 / n)
    return round(float(np.median(ts)), 2)
# FreeDrainage(): since round 4 on the branch-free program (TRM_OPT_ZERO_GRADIENT_FAST); with the option off: the GENERIC instance
w = W.make_workload("richards", lat, lon, 100)
w["bcs"][("pressure_head", "bottom")] = ("gradient", 0.0)
for kernel, fast in (("fused", 1), ("fused", 0), ("unfused", 1)):
    d = W.setup_device(w)
    d.set_option("step_kernel", kernel)
    d.set_option("zero_gradient_fast", fast)
    d.step_heun(w["dt"], 5, finalize=False)
    name = kernel if fast else "fused_generic_instance"
    out[f"richards_Nz100_free_drainage_heun_{name}"] = {"us_per_step": timed(d, d.step_heun_timed, w["dt"]), "status": d.status()}
    d.close()
ncol = min(lat.size, 14017)
wv = W.make_workload("landveg", lat[:ncol], lon[:ncol], 100, hydraulics="vg")
for heun in (False, True):
    for kernel in ("fused", "unfused"):
        d = W.setup_device(wv)
        d.set_option("step_kernel", kernel)
        (d.step_heun if heun else d.step)(wv["dt"], 5, finalize=False)
        out[f"landveg_Nz100_{'heun' if heun else 'euler'}_{kernel}"] = {"columns": ncol, "us_per_step": timed(d, d.step_heun_timed if heun else d.step_timed, wv["dt"]), "status": d.status()}
        d.close()
for config in ("heat", "richards"):
    for Nz in (160, 256):
        w = W.make_workload(config, lat[:ncol], lon[:ncol], Nz)
        for heun in (False, True):
            for kernel in ("fused", "unfused"):
                d = W.setup_device(w)
                d.set_option("step_kernel", kernel)
                (d.step_heun if heun else d.step)(w["dt"], 5, finalize=False)
                out[f"{config}_Nz{Nz}_{'heun' if heun else 'euler'}_{kernel}"] = {"columns": ncol, "us_per_step": timed(d, d.step_heun_timed if heun else d.step_timed, w["dt"]), "status": d.status()}
                d.close()
for config in ("heat", "richards"):
    out[f"{config}_per_cell_ratio_100_vs_64"] = round(out[f"{config}_Nz100_fused"]["ps_per_cell"] / out[f"{config}_Nz64_fused"]["ps_per_cell"], 3)
print(json.dumps(out))
