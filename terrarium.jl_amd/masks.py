"""Land-sea masks on full Gaussian grids (the reference's `inputs/*.nc`).

The two masks the reference ships are packaged as packed bits (terrarium.jl_amd/data/era5_land_mask_N*.npz, made by
data/make_mask_data.py from the reference's NetCDF inputs with the `> 0.5` threshold of
examples/simulations/soil_heat_global.jl:37); any other mask is read straight from its NetCDF-4 file
(`land_mask_from_netcdf`, through the package's own reader in io.py).
Ring order = row-major flatten of [lat N->S][lon 0->360).
"""
import os

import numpy as np

from . import io as _io

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def land_mask_from_netcdf(path: str, variable: str = "lsm", threshold: float = 0.5) -> np.ndarray:
    """`Raster(path; name = :lsm)[Ti(1)] .> 0.5` (examples/simulations/soil_heat_global.jl:30-37): boolean [nlat][nlon]."""
    lsm = _io.Hdf5File(path)[variable].read_cf()    # (packed int16 masks unpack to 0..1; missing values -> NaN -> not land)
    while lsm.ndim > 2:
        lsm = lsm[0]            # the leading time axis has one entry
    return lsm > threshold


def load_land_mask(name: str = "N145", path: str = None) -> np.ndarray:
    """Boolean land mask [nlat][nlon]; N72 -> 14 017 land points, N145 -> 56 951.  `path` may name a NetCDF-4 file
    (read directly) or a packed .npz."""
    if path is not None and not path.endswith(".npz"):
        return land_mask_from_netcdf(path)
    path = path or os.path.join(DATA_DIR, f"era5_land_mask_{name}.npz")
    with np.load(path) as f:
        nlat, nlon = (int(v) for v in f["shape"])
        mask = np.unpackbits(f["packed"])[: nlat * nlon].astype(bool).reshape(nlat, nlon)
        assert int(mask.sum()) == int(f["land_count"])
    return mask


def gaussian_latlon(nlat: int, nlon: int):
    """Latitudes (radians, north -> south, Gauss-Legendre nodes) and longitudes
    (radians, 0 -> 2pi) of a full Gaussian grid."""
    x, _ = np.polynomial.legendre.leggauss(nlat)
    lat = np.arcsin(x[::-1])
    lon = 2.0 * np.pi * np.arange(nlon) / nlon
    return lat, lon


def masked_latlon(mask: np.ndarray):
    """(lat_i, lon_i) in radians of every land column, in ring order."""
    nlat, nlon = mask.shape
    lat, lon = gaussian_latlon(nlat, nlon)
    la = np.repeat(lat[:, None], nlon, axis=1)[mask]
    lo = np.repeat(lon[None, :], nlat, axis=0)[mask]
    return la, lo


def synthetic_latlon(num_columns: int):
    """Quasi-uniform synthetic sphere sampling for the 0.1-degree configuration
    (SURVEY 8(d)): lat_i = asin(2 (i + 1/2) / Nh - 1), lon_i = 2 pi frac(i / phi)."""
    i = np.arange(num_columns, dtype=np.float64)
    lat = np.arcsin(2.0 * (i + 0.5) / num_columns - 1.0)
    phi_inv = 2.0 / (1.0 + np.sqrt(5.0))
    lon = 2.0 * np.pi * np.mod(i * phi_inv, 1.0)
    return lat, lon
