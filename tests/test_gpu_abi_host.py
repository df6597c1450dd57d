"""The C ABI driven by a plain C host (tests/abi_host.c: gcc, dlopen, no Python in the loop on the library side): the
closest stand-in for a Julia `ccall` this image allows.  The C program runs trm_create -> trm_upload -> trm_set_bc ->
trm_initialize -> trm_step -> trm_download on a small soil case written here; its numbers are compared with the oracle's."""
import os
import subprocess

import numpy as np
import pytest

import terrarium_jl_amd as trm
from terrarium_jl_amd import _capi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("richards", [0, 1])
def test_c_host_steps_a_soil_case_and_matches_the_oracle(tmp_path, richards):
    import oracle
    exe = str(tmp_path / "abi_host")
    subprocess.check_call(["gcc", "-O1", "-Wall", "-Werror", "-std=c99", "-o", exe, os.path.join(ROOT, "tests", "abi_host.c"), "-ldl"])
    Nh, Nz, nsteps, dt = 64, 20, 25, 300.0 if not richards else 60.0
    thickness = trm.ExponentialSpacing(dz_min=0.05, dz_max=100.0, N=Nz, sig=3).get_spacing()
    rng = np.random.default_rng(11)
    zc = trm.ColumnGrid(trm.PrescribedSpacing(dz=list(thickness)), Nh).z_centers()
    T0 = rng.uniform(-8.0, 12.0, Nh)                          # frozen and thawed columns
    T = T0[None, :] - 0.05 * zc[:, None]
    sat = np.clip(np.minimum(1.0, 0.8 - 0.05 * zc)[:, None] * (1.0 + 0.05 * rng.uniform(-1, 1, Nh))[None, :], 0.05, 1.0) if richards else np.ones((Nz, Nh))
    Ttop = T0 + 4.0
    case, result = tmp_path / "case.bin", tmp_path / "result.bin"
    with open(case, "wb") as f:
        np.array([Nh, Nz, nsteps, richards], dtype=np.int64).tofile(f)
        np.array([dt], dtype=np.float64).tofile(f)
        for a in (thickness, T, sat, Ttop):
            np.ascontiguousarray(a, dtype=np.float64).tofile(f)
    subprocess.check_call([exe, "run", _capi.LIB_PATH, str(case), str(result)])
    out = np.fromfile(result, dtype=np.float64)
    U, Td, liq = (out[n * Nz * Nh:(n + 1) * Nz * Nh].reshape(Nz, Nh) for n in range(3))
    t, it, flags = out[3 * Nz * Nh:]
    o = oracle.Oracle(Nh, thickness, oracle.default_params(flow=richards))
    o.set("temperature", T)
    o.set("saturation_water_ice", sat)
    o.set_bc("temperature", "top", "value", Ttop)
    o.initialize()
    o.run(dt, nsteps)
    assert (t, it, flags) == (o.clock()[0], nsteps, 0)
    assert np.array_equal(U, o.get("internal_energy")) and np.array_equal(Td, o.get("temperature")) and np.array_equal(liq, o.get("liquid_water_fraction"))
