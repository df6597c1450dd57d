import sys, os
sys.path[:0] = ["/root/repo", "/root/repo/tests", "/root/repo/oracle"]
import numpy as np
import workloads as W
lat, lon = W.columns_from_mask("N72")
sel = np.linspace(0, lat.size - 1, 130).astype(int)
w = W.make_workload("land", lat[sel], lon[sel], 32, hydraulics="vg")
a, orc = W.setup_device(w), W.setup_oracle(w)
for n in range(1, 8):
    a.step(w["dt"], 1, finalize=True)
    orc.timestep(w["dt"], True)
    for name in ("tend_internal_energy", "tend_saturation_water_ice", "temperature", "saturation_water_ice", "ground_heat_flux", "infiltration"):
        x, y = a.get(name), orc.get(name)
        err = np.abs(x - y) / np.maximum(1, np.abs(y))
        idx = np.unravel_index(np.argmax(err), err.shape)
        print(n, name, "max scaled err %.3e at %s: dev %.17g orc %.17g" % (err.max(), idx, x[idx], y[idx]))
