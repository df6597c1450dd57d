"""CPU restatement of the ColumnRingGrid <-> full ring grid conversions (src/grids/column_ring_grid.jl:102-149).

TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Only tests/ may import this module; nothing under terrarium.jl_amd/ does.  It is
deliberately independent of terrarium.jl_amd/grids.py (the product's host mirror): index arithmetic written as explicit loops
over the grid points, from the reference's text alone, so that the device kernels k_scatter_ring / k_gather_ring are checked
against something the product did not compute.  Pinned by the reference's scatter / gather identities (test/grids.jl:44-139;
tests/test_oracle_known_answers.py).

Conventions (column_ring_grid.jl:37-59): `mask` is the Bool vector over the P points of the ring grid in ring order (here: any
array, flattened in C order -- the order in which terrarium.jl_amd/masks.py reads the reference's NetCDF masks); the columns of a
ColumnRingGrid are the true points of the mask, in that order (Nh = sum(mask), column_ring_grid.jl:45-47).
"""
import numpy as np


def ring_field_from_columns(columns, mask, fill_value=np.nan):
    """RingGrids.Field(field, grid; fill_value) (column_ring_grid.jl:102-115): `ring_field = fill(fill_value)`, then
    `ring_field.data[grid.mask.data, :] .= field`.  columns: [Nh] or [rows][Nh]; returns [P] or [rows][P]."""
    m = np.asarray(mask).reshape(-1)
    cols = np.asarray(columns)
    two_d = cols.ndim == 1
    rows = cols.reshape(1, -1) if two_d else cols
    out = np.empty((rows.shape[0], m.size), dtype=rows.dtype)
    column = 0
    for point in range(m.size):                 # logical indexing walks the points in order and consumes one column per true point
        if m[point]:
            for r in range(rows.shape[0]):
                out[r, point] = rows[r, column]
            column += 1
        else:
            for r in range(rows.shape[0]):
                out[r, point] = fill_value
    assert column == rows.shape[1], "one column per true point of the mask"
    return out[0] if two_d else out


def columns_from_ring_field(ring_field, mask):
    """Oceananigans.Field(ring_field, grid) (column_ring_grid.jl:124-149): `interior(field)[:, 1, :] .= ring_field.data[grid.mask.data, :]`
    (only masked points are copied).  ring_field: [P] or [rows][P]; returns [Nh] or [rows][Nh]."""
    m = np.asarray(mask).reshape(-1)
    full = np.asarray(ring_field)
    two_d = full.ndim == 1
    rows = full.reshape(1, -1) if two_d else full.reshape(full.shape[0], -1)
    assert rows.shape[1] == m.size
    n = int(np.count_nonzero(m))
    out = np.empty((rows.shape[0], n), dtype=rows.dtype)
    column = 0
    for point in range(m.size):
        if m[point]:
            for r in range(rows.shape[0]):
                out[r, column] = rows[r, point]
            column += 1
    return out[0] if two_d else out
