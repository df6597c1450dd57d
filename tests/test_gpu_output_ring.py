"""Output path on the device: rows of a field (trm_download_rows) and ColumnRingGrid scatter / gather
(src/grids/column_ring_grid.jl:102-149: RingGrids.Field(field, grid; fill_value) and Oceananigans.Field(ring_field, grid))
against the independent CPU restatement oracle/ring_oracle.py (explicit loops over the grid points; the product's host mirror
terrarium.jl_amd/grids.py is checked against the same restatement on the CPU in tests/test_oracle_known_answers.py) and the
reference's own scatter / gather identity (test/grids.jl:44-139)."""
import numpy as np
import pytest

import workloads as W
import terrarium_jl_amd as trm
import ring_oracle as R

pytestmark = pytest.mark.gpu


def _ring_state(dtype=np.float64, config="richards", Nz=20):
    mask = trm.masks.load_land_mask("N72")
    lat, lon = trm.masks.masked_latlon(mask)
    grid = trm.ColumnRingGrid(trm.ExponentialSpacing(N=Nz), mask, dtype=dtype)
    w = W.make_workload(config, lat, lon, Nz, dtype=dtype)
    d = W.setup_device(w)
    d.set_ring_grid(mask.size, grid.mask_index)
    d.step(w["dt"], 7, finalize=True)
    return grid, mask, d


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_rows_of_a_field(dtype):
    grid, mask, d = _ring_state(dtype)
    T = d.get("temperature")
    assert np.array_equal(d.get_rows("temperature", 0, 20), T)
    assert np.array_equal(d.get_rows("temperature", 19, 1)[0], T[-1]) and np.array_equal(d.get("ground_temperature"), T[-1])
    assert np.array_equal(d.get_rows("temperature", 3, 9), T[3:12])
    K = d.get("hydraulic_conductivity")                       # Face field: Nz + 1 rows, the top face lives in its own buffer
    assert np.array_equal(d.get_rows("hydraulic_conductivity", 0, 21), K)
    assert np.array_equal(d.get_rows("hydraulic_conductivity", 18, 3), K[18:]) and np.array_equal(d.get_rows("hydraulic_conductivity", 20, 1)[0], K[20])
    assert np.array_equal(d.get_rows("water_table", 0, 1)[0], d.get("water_table"))
    with pytest.raises(trm.TerrariumHipError):
        d.get_rows("temperature", 15, 6)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_ring_scatter_and_gather_on_the_device(dtype):
    grid, mask, d = _ring_state(dtype)
    T, wt = d.get("temperature"), d.get("water_table")
    full = d.get_ring("temperature")                          # [Nz][P]
    assert full.shape == (20, mask.size) and full.dtype == np.dtype(dtype)
    assert np.array_equal(full, R.ring_field_from_columns(T, mask), equal_nan=True)
    assert np.array_equal(d.get_ring("water_table", fill=-999.0), R.ring_field_from_columns(wt, mask, -999.0))
    g = d.get_ring("ground_temperature", fill=0.0)            # a view of the top row: one row scattered
    assert g.shape == (mask.size,) and np.array_equal(g, R.ring_field_from_columns(T[-1], mask, 0.0))
    assert np.array_equal(d.get_ring("hydraulic_conductivity", row0=19, nrows=2), R.ring_field_from_columns(d.get("hydraulic_conductivity")[19:], mask), equal_nan=True)
    # gather: a full-grid array becomes the field (test/grids.jl: scatter then gather is the identity)
    rng = np.random.default_rng(5)
    new_full = rng.normal(size=(20, mask.size)).astype(dtype)
    d.set_ring("temperature", new_full)
    assert np.array_equal(d.get("temperature"), R.columns_from_ring_field(new_full, mask))
    d.set_ring("surface_excess_water", new_full[0])
    assert np.array_equal(d.get("surface_excess_water"), R.columns_from_ring_field(new_full[0], mask))
    d.set_ring("temperature", d.get_ring("temperature", fill=0.0))
    assert np.array_equal(d.get("temperature"), R.columns_from_ring_field(new_full, mask))


def test_ring_scatter_of_one_shard_and_into_a_device_buffer():
    """A context that holds one block of the columns (parallel.shard_range) scatters to its own points of the full grid; a
    coupled model on the same device receives the field without a host copy (trm_scatter_ring_device / trm_gather_ring_device)."""
    import torch
    from terrarium_jl_amd import parallel
    mask = trm.masks.load_land_mask("N72")
    lat, lon = trm.masks.masked_latlon(mask)
    grid = trm.ColumnRingGrid(trm.ExponentialSpacing(N=20), mask)
    lo, hi = parallel.shard_range(lat.size, 3, 1)
    w = W.make_workload("heat", lat[lo:hi], lon[lo:hi], 20)
    d = W.setup_device(w)
    d.set_ring_grid(mask.size, grid.mask_index[lo:hi])
    d.step(w["dt"], 3, finalize=True)
    T = d.get("temperature")
    expect = np.full((20, mask.size), np.nan)
    expect[:, grid.mask_index[lo:hi]] = T
    assert np.array_equal(d.get_ring("temperature"), expect, equal_nan=True)
    buf = torch.empty((2, mask.size), dtype=torch.float64, device="cuda")
    d.scatter_ring_to("temperature", buf.data_ptr(), fill=-1.0, row0=18, nrows=2)
    torch.cuda.synchronize()
    assert np.array_equal(buf.cpu().numpy(), np.where(np.isnan(expect[18:]), -1.0, expect[18:]))
    src = torch.arange(mask.size, dtype=torch.float64, device="cuda")
    d.gather_ring_from("air_temperature", src.data_ptr())
    assert np.array_equal(d.get("air_temperature"), grid.mask_index[lo:hi].astype(np.float64))
    with pytest.raises(trm.TerrariumHipError):
        d.set_ring_grid(mask.size, grid.mask_index[lo:hi][::-1])
