"""Column grids and vertical discretisations.

Host-side mirror of the reference's `src/grids/` (ColumnGrid, ColumnRingGrid,
UniformSpacing / ExponentialSpacing / PrescribedSpacing).  Only the metadata
lives here: the z coordinates themselves are derived inside the HIP library
(`trm_create`, csrc/trm_grid.hpp) from the layer thicknesses, following
`ColumnGrid` (src/grids/column_grid.jl:20-34).
"""
from dataclasses import dataclass, field
import math
from typing import Optional, Sequence

import numpy as np


class AbstractVerticalSpacing:
    """src/grids/vertical_discretization.jl:6"""

    def num_layers(self) -> int:
        return self.N

    def get_spacing(self) -> np.ndarray:
        """Layer thicknesses, index 0 = surface layer (vertical_discretization.jl:20)."""
        return np.array([self(i) for i in range(1, self.num_layers() + 1)], dtype=np.float64)


@dataclass
class UniformSpacing(AbstractVerticalSpacing):
    """vertical_discretization.jl:30-35"""
    dz: float = 0.1
    N: int = 100

    def __call__(self, i: int) -> float:
        return self.dz


def _round_sigdigits(x: float, sig: int) -> float:
    # Julia Base: round(x; sigdigits) = _round_digits(x, sig - hidigit(x, 10))
    h = math.floor(math.log10(abs(x))) + 1
    digits = sig - h
    if digits >= 0:
        sc = 10.0 ** digits
        return float(np.rint(x * sc) / sc)
    sc = 10.0 ** (-digits)
    return float(np.rint(x / sc) * sc)


@dataclass
class ExponentialSpacing(AbstractVerticalSpacing):
    """vertical_discretization.jl:47-76: quasi-exponential thickness from dz_min
    at the surface to dz_max at the bottom, rounded to `sig` significant digits."""
    dz_min: float = 0.05
    dz_max: float = 100.0
    N: int = 50
    sig: Optional[int] = 3

    def __post_init__(self):
        assert self.N > 1, "number of grid points for exponential spacing must be > 1"

    def __call__(self, i: int) -> float:
        assert 0 < i <= self.N, f"index {i} out of range"
        l0 = math.log2(self.dz_min)
        ln = math.log2(self.dz_max)
        li = l0 + (i - 1) * (ln - l0) / (self.N - 1)
        v = 2.0 ** li
        return v if self.sig is None else _round_sigdigits(v, self.sig)


@dataclass
class PrescribedSpacing(AbstractVerticalSpacing):
    """vertical_discretization.jl:87-93"""
    dz: Sequence[float] = field(default_factory=list)

    def num_layers(self) -> int:
        return len(self.dz)

    def __call__(self, i: int) -> float:
        return float(self.dz[i - 1])


class ColumnGrid:
    """Set of laterally independent vertical columns (src/grids/column_grid.jl:9-39).

    `dtype` is the number format NF; `device` the HIP device ordinal the state
    lives on.  x spans (0, 1) as in the reference, so dx = 1 / num_columns.
    """

    def __init__(self, vert: AbstractVerticalSpacing, num_columns: int = 1, dtype=np.float64, device: int = 0):
        self.vert = vert
        self.dtype = np.dtype(dtype)
        assert self.dtype in (np.dtype(np.float64), np.dtype(np.float32))
        self.num_columns = int(num_columns)
        self.device = int(device)
        self.thickness = np.ascontiguousarray(vert.get_spacing(), dtype=np.float64)
        self.Nz = int(self.thickness.size)
        self.dx = 1.0 / self.num_columns

    @property
    def Nh(self) -> int:
        return self.num_columns

    def z_faces(self) -> np.ndarray:
        """z of the Nz+1 faces, bottom first (column_grid.jl:31)."""
        cs = np.cumsum(self.thickness)  # sequential, as Julia's cumsum for n < 128
        return np.concatenate([-cs[::-1], [0.0]]).astype(self.dtype)

    def z_centers(self) -> np.ndarray:
        zf = self.z_faces()
        return ((zf[1:] + zf[:-1]) / 2).astype(self.dtype)

    def __repr__(self):
        return f"ColumnGrid{{{self.dtype.name}}}(Nh={self.Nh}, Nz={self.Nz}, device={self.device})"


class ColumnRingGrid(ColumnGrid):
    """Columns = `True` points of a land mask in ring order
    (src/grids/column_ring_grid.jl:37-59).  Ring order is the row-major flatten
    of the [lat N->S][lon 0->360) mask; x spans (1, Nh)."""

    def __init__(self, vert: AbstractVerticalSpacing, mask: np.ndarray, dtype=np.float64, device: int = 0):
        mask = np.asarray(mask, dtype=bool)
        self.mask = mask
        self.mask_index = np.flatnonzero(mask.ravel())
        super().__init__(vert, int(self.mask_index.size), dtype=dtype, device=device)
        self.dx = (self.num_columns - 1) / self.num_columns if self.num_columns > 1 else 1.0

    def scatter(self, columns: np.ndarray, fill=np.nan) -> np.ndarray:
        """`RingGrids.Field(field, grid; fill_value)` (column_ring_grid.jl:102-115): column data `[Nh]` or
        `[rows][Nh]` (the layout of `DeviceState.get`) -> the full ring grid `[nlat][nlon]` / `[rows][nlat][nlon]`,
        `fill` at the unmasked points."""
        columns = np.asarray(columns)
        if columns.shape[-1] != self.num_columns:
            raise ValueError(f"last axis must hold the {self.num_columns} columns of the mask")
        dtype = columns.dtype if np.issubdtype(columns.dtype, np.floating) or not np.isnan(fill) else np.float64
        out = np.full(columns.shape[:-1] + (self.mask.size,), fill, dtype=dtype)
        out[..., self.mask_index] = columns
        return out.reshape(columns.shape[:-1] + self.mask.shape)

    def gather(self, full: np.ndarray) -> np.ndarray:
        """`Oceananigans.Field(ring_field, grid)` (column_ring_grid.jl:117-149): full ring grid `[nlat][nlon]` or
        `[rows][nlat][nlon]` -> the masked points in ring order, `[Nh]` / `[rows][Nh]` (what `DeviceState.set`,
        `set_forcing` and the series calls take)."""
        full = np.asarray(full)
        nd = self.mask.ndim
        if full.shape[-nd:] != self.mask.shape:
            raise ValueError(f"trailing axes must match the mask shape {self.mask.shape}")
        flat = full.reshape(full.shape[:-nd] + (self.mask.size,))
        return np.ascontiguousarray(flat[..., self.mask_index])
