"""Cost of the vegetation-coupled LandModel step against the bare-ground one (C4-VG columns, N145 mask x 32 levels, fp64).
Usage: python profiles/tools/veg_timing.py [steps]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import terrarium_jl_amd as trm  # noqa: E402
import workloads as W  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
lat, lon = W.columns_from_mask("N145")
w = W.make_workload("land", lat, lon, 32, dtype=np.float64, hydraulics="vg")
dt = 0.05     # (the reference's per-year carbon turnover applied per second only survives short steps)
out = {}
for name in ("bare", "coupled"):
    d = W.setup_device(w)
    if name == "coupled":
        d.set_vegetation(trm.flatten_vegetation(trm.VegetationCarbon(), surface_hydrology=trm.SurfaceHydrology.canopy()), "coupled")
        d.set("carbon_vegetation", 1.5)
        d.set("vegetation_area_fraction", 0.5)
        d.set("canopy_water", 5.0e-5)
        d.set_forcing("SAI", 0.5)
    d.step(dt, 20, finalize=False)
    d.save_state()
    best = 1e30
    for _ in range(5):
        d.restore_state()
        best = min(best, d.step_timed(dt, steps, finalize=False) / steps * 1e3)
    if name == "coupled":
        for _ in range(50):
            d.compute_plant_available_water()      # (the cooperative PAW phases alone, for the kernel trace)
    print(name, "status", d.status(), "finite T", bool(np.all(np.isfinite(d.get("temperature")))), file=sys.stderr)
    if name == "coupled":
        for f in ("carbon_vegetation", "canopy_water", "skin_temperature", "transpiration", "soil_moisture_limiting_factor"):
            a = d.get(f)
            print(f, float(np.nanmin(a)), float(np.nanmax(a)), int(np.isnan(a).sum()), file=sys.stderr)
    out[name] = round(best, 2)
out["unit"] = "us/step (device time, best of 5)"
print(json.dumps(out))
