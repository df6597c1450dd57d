"""terrarium.jl_amd/io.py: the dependency-free NetCDF-4 / HDF5 reader and the raster input source.
The reader is checked against files the test writes itself byte by byte (tests/hdf5_writer.py, written from the HDF5
format specification) and, where the reference tree is present, against the two land masks the reference ships."""
import os

import numpy as np
import pytest

import terrarium_jl_amd as trm
from terrarium_jl_amd import io as tio
from hdf5_writer import Writer

REF_INPUTS = "/root/reference/inputs"


def test_lookup3_known_answers():
    # self-test vectors of Bob Jenkins' lookup3.c (driver5)
    assert tio.lookup3(b"Four score and seven years ago", 0) == 0x17770551
    assert tio.lookup3(b"Four score and seven years ago", 1) == 0xCD628161
    assert tio.lookup3(b"", 0) == 0xDEADBEEF


def make_file(rng):
    nt, ny, nx = 5, 24, 32
    w = Writer()
    data = dict(
        time=np.arange(nt, dtype=np.int32),
        x=np.arange(1, nx + 1, dtype=np.int64), y=np.arange(1, ny + 1, dtype=np.float64),
        forcing=rng.random((nt, ny, nx)).astype(np.float32),
        static=rng.standard_normal((ny, nx)),
        small=np.array([1.5, -2.5, 3.25], dtype=np.float32),
        packed=rng.integers(-1000, 1000, (7, 50)).astype(np.int16))
    w.dataset("time", data["time"], attrs={"units": "hours since 2020-01-01 00:00:00", "calendar": "standard"})
    w.dataset("x", data["x"])
    w.dataset("y", data["y"])
    w.dataset("forcing", data["forcing"], layout="chunked", chunks=(2, 10, 32), deflate=True, shuffle=True, attrs={"units": "W/m^2"})
    w.dataset("static", data["static"], layout="chunked", chunks=(24, 32))          # one uncompressed chunk, as the reference masks
    w.dataset("small", data["small"], layout="compact")
    w.dataset("packed", data["packed"], layout="chunked", chunks=(4, 16), deflate=True)
    return w.tobytes(), data


def test_reader_against_byte_by_byte_writer(tmp_path):
    blob, data = make_file(np.random.default_rng(3))
    path = tmp_path / "forcing.nc"
    path.write_bytes(blob)
    f = tio.Hdf5File(str(path))
    assert sorted(f.keys()) == sorted(data)
    for name, a in data.items():
        ds = f[name]
        assert ds.shape == a.shape and ds.dtype == a.dtype, name
        assert np.array_equal(ds.read(), a), name
    assert f["forcing"].layout[0] == "chunked" and [fid for fid, _ in f["forcing"].filters] == [2, 1]
    assert f["time"].attrs == {"units": "hours since 2020-01-01 00:00:00", "calendar": "standard"}
    t, epoch = tio.decode_time_axis(f["time"].read(), f["time"].attrs["units"])
    assert np.array_equal(t, 3600.0 * np.arange(5)) and epoch.year == 2020
    with pytest.raises(KeyError):
        f["nope"]
    # a flipped bit in the metadata is caught by the checksums
    bad = bytearray(blob)
    bad[-20] ^= 0x40
    with pytest.raises(tio.Hdf5FormatError):
        tio.Hdf5File(bytes(bad))["time"]
    with pytest.raises(tio.Hdf5FormatError):
        tio.Hdf5File(b"not an hdf5 file at all" * 10)


@pytest.mark.skipif(not os.path.isdir(REF_INPUTS), reason="the reference tree is not present on this machine")
@pytest.mark.parametrize("name,count,shape", [("N72", 14017, (144, 288)), ("N145", 56951, (290, 580))])
def test_reference_masks_through_the_reader(name, count, shape):
    path = os.path.join(REF_INPUTS, f"era5-land_land_sea_mask_{name}.nc")
    f = tio.Hdf5File(path)
    assert set(f.keys()) == {"time", "lat", "lon", "lsm"}
    lsm = f["lsm"].read()
    assert lsm.shape == (1,) + shape and lsm.dtype == np.float64
    mask = trm.masks.land_mask_from_netcdf(path)
    assert mask.shape == shape and int(mask.sum()) == count
    assert np.array_equal(mask, trm.masks.load_land_mask(name))            # the packaged bits are this file's mask
    assert np.array_equal(mask, trm.masks.load_land_mask(path=path))
    # the file's coordinates are the Gaussian grid the package derives for the columns
    lat, lon = trm.masks.gaussian_latlon(*shape)
    assert np.allclose(np.degrees(lat), f["lat"].read(), atol=1e-6) and np.allclose(np.degrees(lon), f["lon"].read(), atol=1e-9)


def test_packaged_masks_load_without_the_test_tree():
    assert trm.masks.DATA_DIR.endswith(os.path.join("terrarium.jl_amd", "data"))
    assert int(trm.masks.load_land_mask("N72").sum()) == 14017 and int(trm.masks.load_land_mask("N145").sum()) == 56951


# test/inputs/raster_inputs.jl:22-141 (RasterInputSource): static raster, time-indexed raster, index map
def test_raster_input_source_index_map_and_times(tmp_path):
    rng = np.random.default_rng(4)
    blob, data = make_file(rng)
    path = tmp_path / "forcing.nc"
    path.write_bytes(blob)
    mask = rng.random((24, 32)) > 0.6
    grid = trm.ColumnRingGrid(trm.UniformSpacing(dz=0.5, N=5), mask)
    static = tio.RasterInputSource.from_netcdf(grid, str(path), "static", name="air_pressure")
    assert static.static and static.name == "air_pressure"
    assert np.array_equal(static.columns(), data["static"].ravel()[np.flatnonzero(mask.ravel())])      # vec(data)[idxmap]
    src = tio.RasterInputSource.from_netcdf(grid, str(path), "forcing", name="surface_shortwave_down")
    assert not src.static and np.array_equal(src.times, 3600.0 * np.arange(5))                       # reftime = first time
    cols = src.columns()
    assert cols.shape == (5, grid.num_columns)
    for n in range(5):
        assert np.array_equal(cols[n], data["forcing"][n].ravel()[grid.mask_index])
    shifted = tio.RasterInputSource(grid, data["forcing"], "rainfall", times=3600.0 * np.arange(5), reftime=-7200.0)
    assert np.array_equal(shifted.times, 3600.0 * np.arange(5) + 7200.0)
    with pytest.raises(ValueError):
        tio.RasterInputSource(grid, data["forcing"][:, :10], "rainfall", times=np.arange(5.0))


def test_oracle_raster_interpolation_rule():
    """update_from_raster! (TerrariumRastersExt.jl:96-121) as restated in the oracle: nodes, midpoints, flat ends."""
    import oracle
    o = oracle.Oracle(3, trm.UniformSpacing(dz=0.5, N=5).get_spacing(), oracle.default_params(seb=1, flow=1))
    times = np.array([0.0, 3600.0, 7200.0, 10800.0])
    vals = np.array([[1.0, 2.0, 3.0], [3.0, 2.0, -1.0], [5.0, 2.5, 0.0], [4.0, 2.0, 8.0]])
    o.set_forcing_series("air_temperature", times, vals, "raster")
    for t, expect in ((0.0, vals[0]), (3600.0, vals[1]), (1800.0, (vals[0] + vals[1]) / 2), (-50.0, vals[0]), (20000.0, vals[3]),
                      (9000.0, vals[2] + 1800.0 * (vals[3] - vals[2]) / 3600.0)):
        o.set_clock(t)
        o.update_inputs()
        assert np.array_equal(o.get("air_temperature"), expect), t


def test_cf_packing_is_applied_like_rasters_does(tmp_path):
    """Classic ERA5 downloads are int16-packed (scale_factor / add_offset / _FillValue / missing_value).  Rasters.jl /
    NCDatasets.jl unpack by default on the reference side (and the ERA5 example calls replace_missing(..., NaN)); the reader's
    read_cf() does the same, RasterInputSource.from_netcdf and land_mask_from_netcdf go through it.  Chunks that were never
    allocated read as the fill value, i.e. missing."""
    from hdf5_writer import Writer
    rng = np.random.default_rng(9)
    scale, offset, fill = 0.0125, 273.15, np.int16(-32767)
    raw = rng.integers(-3000, 3000, size=(3, 12, 16)).astype(np.int16)
    raw[1, 4, 5] = fill
    raw[2, 0, 0] = fill
    w = Writer()
    w.dataset("t2m", raw, layout="chunked", chunks=(1, 8, 8), deflate=True, shuffle=True,
              attrs=dict(scale_factor=np.float64(scale), add_offset=np.float64(offset), _FillValue=fill, missing_value=fill, units="K"))
    w.dataset("time", np.arange(3, dtype=np.float64), attrs=dict(units="hours since 2020-01-01 00:00:00"))
    lsm = (rng.random((1, 12, 16)) * 65534 - 32767).astype(np.int16)
    w.dataset("lsm", lsm, attrs=dict(scale_factor=np.float64(1 / 65534), add_offset=np.float64(0.5)))
    w.dataset("plain", np.arange(6, dtype=np.float32).reshape(2, 3))
    path = tmp_path / "era5_packed.nc"
    path.write_bytes(w.tobytes())
    f = tio.Hdf5File(str(path))
    got = f["t2m"].read_cf()
    expect = raw.astype(np.float64) * scale + offset
    expect[raw == fill] = np.nan
    assert got.dtype == np.float64 and np.array_equal(got, expect, equal_nan=True) and np.isnan(got[1, 4, 5]) and np.isnan(got[2, 0, 0])
    assert np.array_equal(f["t2m"].read(), raw)                              # the stored values stay available
    assert f["plain"].read_cf().dtype == np.float32                          # nothing to unpack: returned as stored
    mask = rng.random((12, 16)) > 0.5
    grid = trm.ColumnRingGrid(trm.UniformSpacing(dz=0.5, N=5), mask)
    src = tio.RasterInputSource.from_netcdf(grid, str(path), "t2m", name="air_temperature")
    assert np.array_equal(src.times, 3600.0 * np.arange(3)) and np.array_equal(src.columns(), grid.gather(expect), equal_nan=True)
    land = trm.masks.land_mask_from_netcdf(str(path))
    assert np.array_equal(land, (lsm[0].astype(np.float64) / 65534 + 0.5) > 0.5)
