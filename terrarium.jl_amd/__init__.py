"""MI355X-native SoilModel / LandModel(bare ground) time-stepping core.

Host-side mirror of the Terrarium.jl interface for the one hot path this
repository accelerates: the explicit time step of
`SoilModel` -- heat conduction with freeze/thaw, Richards water transport and
the bare-ground surface energy balance -- over laterally independent columns.
All arithmetic runs in hand-written HIP kernels (csrc/) behind the C ABI of
include/terrarium_hip.h; there is no CPU fallback.
"""
from .grids import (AbstractVerticalSpacing, UniformSpacing, ExponentialSpacing, PrescribedSpacing, ColumnGrid,
                    ColumnRingGrid)
from . import masks

__all__ = ["AbstractVerticalSpacing", "UniformSpacing", "ExponentialSpacing", "PrescribedSpacing", "ColumnGrid",
           "ColumnRingGrid", "masks"]
