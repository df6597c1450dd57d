"""Per-cell time of the fused per-step launches on columns of 65 ... 128 levels -- two levels per lane and one column per wave
(k_column_deep, TRM_OPT_GROUPED_COLUMNS = 0) against five levels per lane on groups of lanes (k_column_wide<M = 5>, = 1) -- and
against Nz = 64 on one level per lane, every state of a comparison on the same side of the 256 MiB Infinity Cache:
    python profiles/tools/grouped_timing.py [copies of the N145 columns, default 2: every state HBM-resident] [mask]"""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
import workloads as W
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 2
mask = sys.argv[2] if len(sys.argv) > 2 else "N145"
lat, lon = W.columns_from_mask(mask)
lat, lon = np.tile(lat, copies), np.tile(lon, copies)
out = {"mask": mask, "copies": copies, "columns": int(lat.size)}


def timed(w, grouped, heun=False, nsteps=40):
    d = W.setup_device(w)
    try:
        d.set_option("grouped_columns", grouped)      # (only in a library built with profiles/r05/exp7_grouped_columns.patch)
    except KeyError:
        if grouped:
            d.close()
            return None
    step = d.step_heun_timed if heun else d.step_timed
    (d.step_heun if heun else d.step)(w["dt"], 6, finalize=False)
    d.save_state()
    ts = []
    for _ in range(5):
        d.restore_state()
        step(w["dt"], nsteps, finalize=False)
        d.restore_state()
        ts.append(step(w["dt"], nsteps, finalize=False) * 1e3 / nsteps)
    p = d.last_program()
    st = d.status()
    d.close()
    return float(np.median(ts)), p, st


for config in ("heat", "richards"):
    base = None
    for Nz in (64, 100, 80, 65, 120):
        w = W.make_workload(config, lat, lon, Nz)
        state_mb = (6 if config == "richards" else 4) * lat.size * (64 if Nz <= 64 else 128) * 8 / 2 ** 20
        for grouped in ((0,) if Nz <= 64 else (0, 1)):
            r = timed(w, grouped)
            if r is None:
                continue
            us, p, st = r
            ps = us * 1e6 / (lat.size * Nz)
            if Nz == 64:
                base = ps
            key = f"{config}_Nz{Nz}_" + ("one_level_per_lane" if Nz <= 64 else ("grouped" if grouped else "two_levels_per_lane"))
            out[key] = {"us_per_step": round(us, 2), "ps_per_cell": round(ps, 3), "vs_Nz64_per_cell": round(ps / base, 3), "family": p["family"],
                        "levels_per_lane": p.get("levels_per_lane", 1), "derive": p["derive"], "allocated_state_MB": round(state_mb), "status": st}
            print(key, out[key], flush=True)
    w = W.make_workload(config, lat, lon, 100)
    for grouped in (0, 1):
        r = timed(w, grouped, heun=True, nsteps=20)
        if r is None:
            continue
        us, p, st = r
        out[f"{config}_Nz100_heun_" + ("grouped" if grouped else "two_levels_per_lane")] = {"us_per_step": round(us, 2), "status": st}
print(json.dumps(out))
