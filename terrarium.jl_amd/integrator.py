"""Time steppers and the model integrator -- host-side mirror of the reference's
`src/timesteppers/` (ForwardEuler, Heun, ModelIntegrator, run!, timestep!) and of
the state-variable interface, driving libterrarium_hip.so through its C ABI.

What stays on the host (as in the reference, where the host issues launches):
evaluation of time-dependent boundary-condition / forcing *functions* to
per-column arrays once per step (the library then reads arrays), bookkeeping,
and sharding.  All field arithmetic runs on the GPU.
"""
import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _capi
from . import models as M
from .io import RasterInputSource
from .grids import ColumnGrid


# ---- time steppers ------------------------------------------------------------
@dataclass
class ForwardEuler:
    """src/timesteppers/forward_euler.jl:6-9"""
    dt: float = 300.0


@dataclass
class Heun:
    """src/timesteppers/heun.jl:8-14"""
    dt: float = 300.0


# ---- boundary-condition aliases (src/models/soil/soil_model_bcs.jl) -----------
def _bc(var, side, kind, value):
    return {(var, side): (kind, value)}


def PrescribedSurfaceTemperature(name, value):
    """ValueBoundaryCondition on top `temperature` (soil_model_bcs.jl:17)."""
    return _bc("temperature", "top", "value", value)


def PrescribedBottomTemperature(name, value):
    return _bc("temperature", "bottom", "value", value)


def GroundHeatFlux(value):
    """FluxBoundaryCondition on top `internal_energy` (soil_model_bcs.jl:6)."""
    return _bc("internal_energy", "top", "flux", value)


def GeothermalHeatFlux(value):
    return _bc("internal_energy", "bottom", "flux", value)


def InfiltrationFlux(value):
    """FluxBoundaryCondition on top `saturation_water_ice` (soil_model_bcs.jl:29); positive = upward."""
    return _bc("saturation_water_ice", "top", "flux", value)


def ImpermeableBoundary():
    return _bc("saturation_water_ice", "bottom", "noflux", 0.0)


def FreeDrainage():
    """GradientBoundaryCondition(0) on bottom `pressure_head` (soil_model_bcs.jl:40)."""
    return _bc("pressure_head", "bottom", "gradient", 0.0)


def merge_boundary_conditions(*bcs):
    """src/boundary_conditions.jl:19"""
    out = {}
    for b in bcs:
        out.update(b or {})
    return out


# ---- device context -------------------------------------------------------------
class FieldTimeSeries:
    """Stand-in for an Oceananigans `FieldTimeSeries` of a 2-D field: `values[nt][Nh]` (or `[nt]`, broadcast over the
    columns) at strictly increasing `times[nt]` (seconds), with `time_indexing` in {"linear", "clamp", "cyclical"}.
    Given as an input (`InputSource(fts; name)`, input_sources.jl:142-171) or as a boundary value it is uploaded
    once and evaluated on the device at the integrator's clock by every step, so `run!` stays ONE library call."""

    def __init__(self, times, values, time_indexing="linear"):
        self.times = np.asarray(times, dtype=np.float64)
        self.values = np.asarray(values)
        if time_indexing not in _capi.TIME_INDEXING:
            raise ValueError(f"time_indexing must be one of {sorted(_capi.TIME_INDEXING)}")
        if self.values.shape[0] != self.times.size:
            raise ValueError("values must have one row per time")
        self.time_indexing = time_indexing
        self.window = None          # levels of the device window (None: the whole record is resident)

    def windowed(self, levels):
        """The same series streamed through a device window of `levels` time levels (Oceananigans' `InMemory(chunk)` backend
        of a FieldTimeSeries on disk): `run` / `Simulation` then append and trim between library calls
        (trm_series_append / trm_series_trim_before) instead of holding the whole record in HBM.  `values` may be any
        array-like that supports slicing along the first axis (a memory-mapped file)."""
        if self.time_indexing == "cyclical":
            raise ValueError("a cyclical series is periodic over its whole record and cannot be windowed")
        self.window = int(levels)
        if self.window < 3:
            raise ValueError("the window must hold at least 3 time levels")
        return self

    @classmethod
    def from_function(cls, f, times, time_indexing="linear"):
        """Samples f(t) -> number | per-column array at `times`."""
        times = np.asarray(times, dtype=np.float64)
        return cls(times, np.stack([np.asarray(f(float(t)), dtype=np.float64) for t in times]), time_indexing)


def InputSource(source, name=None):
    """InputSource(field | fts | raster; name) (input_sources.jl:104-160, ext/TerrariumRastersExt:45-52): a (name, source)
    pair for `initialize(...; inputs)`.  A `RasterInputSource` carries its own name."""
    return (name or source.name, source)


def InputSources(*sources):
    """InputSources(sources...) (input_sources.jl:32-50) -> dict for `initialize(...; inputs)`."""
    return dict(sources)


class DeviceArray:
    """A view of library-owned device memory (CUDA array interface v2; HIP pointers on ROCm builds of the consumers)."""

    def __init__(self, ptr, shape, dtype, owner):
        self._owner = owner      # keeps the context alive
        self.__cuda_array_interface__ = dict(shape=tuple(shape), typestr=np.dtype(dtype).str, data=(int(ptr), False), version=2, strides=None)


class DeviceState:
    """Owns one `trm_ctx` (one device's shard of columns) and exposes the state
    variables by the reference's names; reading an attribute downloads the field
    as an array [rows][Nh] (row 0 = bottom layer) -- `Array(interior(field))`."""

    def __init__(self, grid: ColumnGrid, params: "_capi.TrmParams"):
        object.__setattr__(self, "_ctx", None)
        self._lib = _capi.lib()
        self.grid = grid
        self.dtype = grid.dtype
        self.params = params
        th = np.ascontiguousarray(grid.thickness, dtype=np.float64)
        g = _capi.TrmGrid(_capi.TRM_F64 if grid.dtype == np.float64 else _capi.TRM_F32, grid.Nz, grid.Nh,
                          th.ctypes.data_as(C.POINTER(C.c_double)), float(grid.dx), int(grid.device), 0)
        h = C.c_void_p()
        rc = self._lib.trm_create(C.byref(g), C.byref(params), C.byref(h))
        if rc != 0:
            raise _capi.TerrariumHipError(f"trm_create failed (code {rc}): {self._lib.trm_last_error(None).decode()}")
        self._ctx = h

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.trm_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        _capi.check(self._ctx, rc, what)

    # -- fields -----------------------------------------------------------------
    def rows(self, name):
        r = C.c_int64()
        self._check(self._lib.trm_field_rows(self._ctx, _capi.FIELD[name], C.byref(r)), "trm_field_rows")
        return int(r.value)

    def get(self, name) -> np.ndarray:
        if name == "ground_temperature":
            if getattr(self, "vegetation_mode", "off") == "standalone":   # an input of the VegetationModel (no soil)
                name = "vegetation_ground_temperature"
            else:                                                          # view of the top soil layer (soil_energy.jl:52-57): one row
                return self.get_rows("temperature", self.grid.Nz - 1, 1)[0]
        if name == "rainfall_ground" and getattr(self, "vegetation_mode", "off") != "coupled":
            name = "rainfall"          # NoCanopyInterception: alias of rainfall (canopy_interception.jl:11-15)
        rows = self.rows(name)
        a = np.empty((rows, self.grid.Nh), dtype=self.dtype)
        self._check(self._lib.trm_download(self._ctx, _capi.FIELD[name], a.ctypes.data), "trm_download")
        return a[0] if rows == 1 else a

    def _alias(self, name):
        """(field, first row, rows) of a reference variable name that is a view or an alias of another field"""
        if name == "ground_temperature" and getattr(self, "vegetation_mode", "off") != "standalone":
            return "temperature", self.grid.Nz - 1, 1
        if name == "ground_temperature":
            name = "vegetation_ground_temperature"
        if name == "rainfall_ground" and getattr(self, "vegetation_mode", "off") != "coupled":
            name = "rainfall"
        return name, 0, self.rows(name)

    def get_rows(self, name, row0, nrows) -> np.ndarray:
        """Rows [row0, row0 + nrows) of a field as `[nrows][Nh]` (trm_download_rows): only these rows cross PCIe."""
        a = np.empty((nrows, self.grid.Nh), dtype=self.dtype)
        self._check(self._lib.trm_download_rows(self._ctx, _capi.FIELD[name], int(row0), int(nrows), a.ctypes.data), "trm_download_rows")
        return a

    # -- ring grid: scatter / gather on the device (column_ring_grid.jl:102-149) ---------------------------------------
    def set_ring_grid(self, num_points, mask_index):
        """The columns are the points `mask_index` of a full grid of `num_points` points (trm_set_ring_grid)."""
        idx = np.ascontiguousarray(mask_index, dtype=np.int64)
        assert idx.shape == (self.grid.Nh,)
        self._check(self._lib.trm_set_ring_grid(self._ctx, int(num_points), idx.ctypes.data), "trm_set_ring_grid")
        self.ring_points = int(num_points)

    def get_ring(self, name, fill=np.nan, row0=None, nrows=None) -> np.ndarray:
        """`RingGrids.Field(field, grid; fill_value)`: the variable on the full grid, `[num_points]` for one row or
        `[nrows][num_points]`, scattered on the device (trm_download_ring)."""
        field, r0, nr = self._alias(name)
        if row0 is not None:
            r0, nr = r0 + int(row0), int(nrows if nrows is not None else nr - row0)
        a = np.empty((nr, self.ring_points), dtype=self.dtype)
        self._check(self._lib.trm_download_ring(self._ctx, _capi.FIELD[field], r0, nr, float(fill), a.ctypes.data), "trm_download_ring")
        return a[0] if nr == 1 else a

    def set_ring(self, name, full):
        """`Oceananigans.Field(ring_field, grid)`: the masked points of a full-grid array `[num_points]` / `[rows][num_points]`
        become the field, gathered on the device (trm_upload_ring)."""
        rows = self.rows(name)
        a = np.empty((rows, self.ring_points), dtype=self.dtype)
        a[...] = np.asarray(full).reshape(rows, self.ring_points)
        self._check(self._lib.trm_upload_ring(self._ctx, _capi.FIELD[name], a.ctypes.data), "trm_upload_ring")

    def scatter_ring_to(self, name, device_ptr, fill=np.nan, row0=0, nrows=None):
        """Scatter rows of a field into a device buffer `[nrows][num_points]` of a coupled model (trm_scatter_ring_device)."""
        field, r0, nr = self._alias(name)
        nr = int(nrows) if nrows is not None else nr - int(row0)
        self._check(self._lib.trm_scatter_ring_device(self._ctx, _capi.FIELD[field], r0 + int(row0), nr, float(fill), C.c_void_p(int(device_ptr))), "trm_scatter_ring_device")

    def gather_ring_from(self, name, device_ptr):
        """The field from a full-grid device array `[rows][num_points]` of a coupled model (trm_gather_ring_device)."""
        self._check(self._lib.trm_gather_ring_device(self._ctx, _capi.FIELD[name], C.c_void_p(int(device_ptr))), "trm_gather_ring_device")

    def reset(self):
        """reset!(state) + reset!(clock) (trm_reset)."""
        self._check(self._lib.trm_reset(self._ctx), "trm_reset")

    def set(self, name, value):
        """set!(field, number | array | function(x, z))"""
        rows = self.rows(name)
        Nh = self.grid.Nh
        if callable(value):
            zc = self.z_centers() if rows == self.grid.Nz else self.z_faces()
            x = np.arange(1, Nh + 1, dtype=np.float64)
            value = np.array([[value(x[i], zc[k]) for i in range(Nh)] for k in range(rows)], dtype=np.float64) \
                if rows > 1 else np.array([value(x[i]) for i in range(Nh)], dtype=np.float64)
        a = np.empty((rows, Nh), dtype=self.dtype)
        v = np.asarray(value)
        if v.ndim == 1 and rows > 1 and v.size == rows and rows != Nh:
            v = v.reshape(rows, 1)  # a vertical profile shared by every column
        a[...] = v
        self._check(self._lib.trm_upload(self._ctx, _capi.FIELD[name], a.ctypes.data), "trm_upload")

    def __getattr__(self, name):
        if name in _capi.FIELD or name in ("ground_temperature", "rainfall_ground"):
            return self.get(name)
        raise AttributeError(name)

    def set_bc(self, var, side, kind, value=0.0):
        if np.ndim(value) == 0:
            ptr, scalar = None, float(value)
        else:
            arr = np.ascontiguousarray(value, dtype=self.dtype)
            assert arr.shape == (self.grid.Nh,), arr.shape
            ptr, scalar = arr.ctypes.data, 0.0
        self._check(self._lib.trm_set_bc(self._ctx, _capi.BC_VAR[var], _capi.SIDE[side], _capi.BC_KIND[kind], ptr,
                                         scalar), "trm_set_bc")

    # -- zero-copy device views (coupling with a model on the same device; speedy_dry_land.jl:45-66) -----------------------
    def device_array(self, name) -> "DeviceArray":
        """The device buffer of a field as a `__cuda_array_interface__` object (`torch.as_tensor(a, device="cuda")`, CuPy, ...):
        `[num_columns]` for 2-D fields, `[num_columns][pitch]` for 3-D fields -- the z-FASTEST device layout, level k of
        column i at [i, k], k = 0 the bottom cell, entries k >= rows are padding."""
        ptr, pitch = C.c_void_p(), C.c_int64()
        self._check(self._lib.trm_field_device_ptr(self._ctx, _capi.FIELD[name], C.byref(ptr), C.byref(pitch)), "trm_field_device_ptr")
        shape = (self.grid.Nh,) if self.rows(name) == 1 else (self.grid.Nh, int(pitch.value))
        return DeviceArray(ptr.value, shape, self.dtype, self)

    def bc_device_array(self, var, side) -> "DeviceArray":
        """The device array `[num_columns]` of a boundary condition's values (trm_bc_device_ptr)."""
        ptr = C.c_void_p()
        self._check(self._lib.trm_bc_device_ptr(self._ctx, _capi.BC_VAR[var], _capi.SIDE[side], C.byref(ptr)), "trm_bc_device_ptr")
        return DeviceArray(ptr.value, (self.grid.Nh,), self.dtype, self)

    def set_vegetation(self, veg_params, mode="standalone"):
        """Enables the vegetation processes (trm_set_vegetation)."""
        self._check(self._lib.trm_set_vegetation(self._ctx, C.byref(veg_params), _capi.VEGETATION[mode]), "trm_set_vegetation")
        self.vegetation_mode = mode

    def compute_plant_available_water(self):
        self._check(self._lib.trm_compute_plant_available_water(self._ctx), "trm_compute_plant_available_water")

    def set_forcing(self, name, value):
        if name == "ground_temperature":
            name = "vegetation_ground_temperature"
        a = np.empty(self.grid.Nh, dtype=self.dtype)
        a[...] = value
        self._check(self._lib.trm_set_forcing(self._ctx, _capi.FIELD[name], a.ctypes.data), "trm_set_forcing")

    # -- time series input sources (input_sources.jl:142-171) ------------------------
    def _series_args(self, times, values, time_indexing):
        times = np.ascontiguousarray(times, dtype=np.float64)
        vals = np.ascontiguousarray(values, dtype=self.dtype)
        if vals.ndim == 1:      # one value per time, broadcast over the columns
            vals = np.ascontiguousarray(np.broadcast_to(vals[:, None], (vals.size, self.grid.Nh)))
        assert vals.shape == (times.size, self.grid.Nh), (vals.shape, times.size, self.grid.Nh)
        return times, vals, _capi.TIME_INDEXING[time_indexing]

    def set_forcing_series(self, name, times, values, time_indexing="linear"):
        """FieldTimeSeriesInputSource: `values[nt][Nh]` at `times[nt]`, resident on the device and evaluated at
        the clock by every step (trm_set_forcing_series)."""
        t, v, ti = self._series_args(times, values, time_indexing)
        self._check(self._lib.trm_set_forcing_series(self._ctx, _capi.FIELD[name], t.size, t.ctypes.data, v.ctypes.data, ti),
                    "trm_set_forcing_series")

    def set_bc_series(self, var, side, kind, times, values, time_indexing="linear"):
        """Time-dependent boundary values (the reference's functional boundary conditions sampled at `times`)."""
        t, v, ti = self._series_args(times, values, time_indexing)
        self._check(self._lib.trm_set_bc_series(self._ctx, _capi.BC_VAR[var], _capi.SIDE[side], _capi.BC_KIND[kind],
                                                t.size, t.ctypes.data, v.ctypes.data, ti), "trm_set_bc_series")

    def _series_id(self, target):
        """`target`: an input field name, or a (bc variable, side) pair"""
        if isinstance(target, tuple):
            return 1, _capi.BC_VAR[target[0]], _capi.SIDE[target[1]]
        return 0, _capi.FIELD["vegetation_ground_temperature" if target == "ground_temperature" else target], 0

    def series_append(self, target, times, values):
        """Continues a series with further time levels through the side stream (trm_series_append)."""
        is_bc, sid, side = self._series_id(target)
        t, v, _ = self._series_args(times, values, "linear")
        self._check(self._lib.trm_series_append(self._ctx, is_bc, sid, side, t.size, t.ctypes.data, v.ctypes.data), "trm_series_append")

    def series_window(self, target, levels=0):
        """Declares a series as windowed: trimmed by `series_trim_before`, strict about times before its head (trm_series_window)."""
        is_bc, sid, side = self._series_id(target)
        self._check(self._lib.trm_series_window(self._ctx, is_bc, sid, side, int(levels)), "trm_series_window")

    def series_trim_before(self, t):
        """Releases the time levels no evaluation at a time >= t can touch (trm_series_trim_before)."""
        self._check(self._lib.trm_series_trim_before(self._ctx, float(t)), "trm_series_trim_before")

    def series_info(self, target):
        is_bc, sid, side = self._series_id(target)
        n, cap, t0, t1 = C.c_int64(), C.c_int64(), C.c_double(), C.c_double()
        self._check(self._lib.trm_series_info(self._ctx, is_bc, sid, side, C.byref(n), C.byref(cap), C.byref(t0), C.byref(t1)), "trm_series_info")
        return dict(levels=int(n.value), capacity=int(cap.value), t_first=t0.value, t_last=t1.value)

    def save_state(self): self._check(self._lib.trm_save_state(self._ctx), "trm_save_state")
    def restore_state(self): self._check(self._lib.trm_restore_state(self._ctx), "trm_restore_state")
    def clear_series(self): self._check(self._lib.trm_clear_series(self._ctx), "trm_clear_series")
    def update_inputs(self): self._check(self._lib.trm_update_inputs(self._ctx), "trm_update_inputs")

    # -- grid ---------------------------------------------------------------------
    def _grid_arrays(self):
        Nz = self.grid.Nz
        zf, dzf = np.zeros(Nz + 1), np.zeros(Nz + 1)
        zc, dzc = np.zeros(Nz), np.zeros(Nz)
        self._check(self._lib.trm_get_grid(self._ctx, zf.ctypes.data, zc.ctypes.data, dzc.ctypes.data, dzf.ctypes.data),
                    "trm_get_grid")
        return dict(zF=zf, zC=zc, dzc=dzc, dzf=dzf)

    def z_centers(self):
        return self._grid_arrays()["zC"]

    def z_faces(self):
        return self._grid_arrays()["zF"]

    # -- reference process interface (abstract_model.jl:52-95,175-215) -------------
    def initialize(self): self._check(self._lib.trm_initialize(self._ctx), "trm_initialize")
    def update_state(self, compute_tendencies=True):
        self._check(self._lib.trm_update_state(self._ctx, int(compute_tendencies)), "trm_update_state")
    def compute_auxiliary(self): self._check(self._lib.trm_compute_auxiliary(self._ctx), "trm_compute_auxiliary")
    def compute_tendencies(self): self._check(self._lib.trm_compute_tendencies(self._ctx), "trm_compute_tendencies")
    def reset_tendencies(self): self._check(self._lib.trm_reset_tendencies(self._ctx), "trm_reset_tendencies")
    def explicit_step(self, dt): self._check(self._lib.trm_explicit_step(self._ctx, float(dt)), "trm_explicit_step")
    def closure(self): self._check(self._lib.trm_closure(self._ctx), "trm_closure")
    def invclosure(self): self._check(self._lib.trm_invclosure(self._ctx), "trm_invclosure")

    def step(self, dt, nsteps=1, finalize=True):
        self._check(self._lib.trm_step(self._ctx, float(dt), int(nsteps), int(finalize)), "trm_step")

    def step_heun(self, dt, nsteps=1, finalize=True):
        self._check(self._lib.trm_step_heun(self._ctx, float(dt), int(nsteps), int(finalize)), "trm_step_heun")

    def step_timed(self, dt, nsteps=1, finalize=False) -> float:
        ms = C.c_float()
        self._check(self._lib.trm_step_timed(self._ctx, float(dt), int(nsteps), int(finalize), C.byref(ms)),
                    "trm_step_timed")
        return float(ms.value)

    def step_heun_timed(self, dt, nsteps=1, finalize=False) -> float:
        ms = C.c_float()
        self._check(self._lib.trm_step_heun_timed(self._ctx, float(dt), int(nsteps), int(finalize), C.byref(ms)), "trm_step_heun_timed")
        return float(ms.value)

    # -- Heun in two calls (state-dependent forcings / boundary values evaluated at the stage in between) ------------------
    def heun_predict(self, dt):
        """heun.jl:41-52: update_state!(state), stage := state, explicit_step!(stage), closure!(stage) (trm_heun_predict)."""
        self._check(self._lib.trm_heun_predict(self._ctx, float(dt)), "trm_heun_predict")

    def heun_stage_auxiliary(self):
        """reset tendencies + compute_auxiliary!(stage): the first half of update_state!(stage) (trm_heun_stage_auxiliary) -- after
        it the stage's auxiliary fields are its own, as a forcing function inside the tendency kernel finds them."""
        self._check(self._lib.trm_heun_stage_auxiliary(self._ctx), "trm_heun_stage_auxiliary")

    def heun_correct(self, dt, finalize=True):
        """heun.jl:54-71: update_state!(stage), average_tendencies!, explicit_step!(state), closure!(state) (trm_heun_correct)."""
        self._check(self._lib.trm_heun_correct(self._ctx, float(dt), int(finalize)), "trm_heun_correct")

    def stage_device_array(self, name) -> "DeviceArray":
        """The device buffer of a field of the Heun STAGE (layout as `device_array`; trm_stage_field_device_ptr)."""
        ptr, pitch = C.c_void_p(), C.c_int64()
        self._check(self._lib.trm_stage_field_device_ptr(self._ctx, _capi.FIELD[name], C.byref(ptr), C.byref(pitch)), "trm_stage_field_device_ptr")
        shape = (self.grid.Nh,) if self.rows(name) == 1 else (self.grid.Nh, int(pitch.value))
        return DeviceArray(ptr.value, shape, self.dtype, self)

    def stage_bc_device_array(self, var, side) -> "DeviceArray":
        """The device array `[num_columns]` of the STAGE's boundary values (trm_stage_bc_device_ptr)."""
        ptr = C.c_void_p()
        self._check(self._lib.trm_stage_bc_device_ptr(self._ctx, _capi.BC_VAR[var], _capi.SIDE[side], C.byref(ptr)), "trm_stage_bc_device_ptr")
        return DeviceArray(ptr.value, (self.grid.Nh,), self.dtype, self)

    def set_forcing_device(self, name, device_ptr):
        """An input field from DEVICE memory `[num_columns]`, stream-ordered, the host does not wait (trm_set_forcing_device)."""
        if name == "ground_temperature":
            name = "vegetation_ground_temperature"
        self._check(self._lib.trm_set_forcing_device(self._ctx, _capi.FIELD[name], C.c_void_p(int(device_ptr))), "trm_set_forcing_device")

    def clock(self):
        t, it = C.c_double(), C.c_int64()
        self._check(self._lib.trm_clock(self._ctx, C.byref(t), C.byref(it)), "trm_clock")
        return t.value, int(it.value)

    def set_clock(self, time, iteration):
        self._check(self._lib.trm_set_clock(self._ctx, float(time), int(iteration)), "trm_set_clock")

    def reduce(self, name, op) -> np.ndarray:
        rows = 1 if op == "volume_integral_z" else self.rows(name)
        out = np.zeros(rows, dtype=np.float64)
        self._check(self._lib.trm_reduce(self._ctx, _capi.FIELD[name], _capi.REDUCE[op], out.ctypes.data), "trm_reduce")
        return out

    # -- multi-device diagnostics (RCCL inside the library) ---------------------------------------------------
    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        _capi.check(None, self._lib.trm_comm_unique_id(buf), "trm_comm_unique_id")
        return buf.raw

    def comm_init(self, rank: int, world_size: int, unique_id: bytes):
        assert len(unique_id) == 128
        self._check(self._lib.trm_comm_init(self._ctx, int(rank), int(world_size), C.c_char_p(unique_id)), "trm_comm_init")

    def comm_world(self) -> int:
        r, w = C.c_int(), C.c_int()
        self._check(self._lib.trm_comm_info(self._ctx, C.byref(r), C.byref(w)), "trm_comm_info")
        return int(w.value)

    def comm_rank(self) -> int:
        r, w = C.c_int(), C.c_int()
        self._check(self._lib.trm_comm_info(self._ctx, C.byref(r), C.byref(w)), "trm_comm_info")
        return int(r.value)

    def comm_destroy(self):
        self._check(self._lib.trm_comm_destroy(self._ctx), "trm_comm_destroy")

    def reduce_global(self, name, op) -> np.ndarray:
        rows = 1 if op == "volume_integral_z" else self.rows(name)
        out = np.zeros(rows, dtype=np.float64)
        self._check(self._lib.trm_reduce_global(self._ctx, _capi.FIELD[name], _capi.REDUCE[op], out.ctypes.data), "trm_reduce_global")
        return out

    def status_global(self) -> int:
        f = C.c_uint32()
        self._check(self._lib.trm_status_global(self._ctx, C.byref(f)), "trm_status_global")
        return int(f.value)

    def status(self) -> int:
        f = C.c_uint32()
        self._check(self._lib.trm_status(self._ctx, C.byref(f)), "trm_status")
        return int(f.value)

    def set_status(self, flags: int):
        """Puts a status word back (trm_set_status): what a restart does with the flags of its checkpoint."""
        self._check(self._lib.trm_set_status(self._ctx, C.c_uint32(int(flags))), "trm_set_status")

    def set_option(self, option, value):
        if option == "step_kernel" and isinstance(value, str):
            value = _capi.KERNEL[value]
        self._check(self._lib.trm_set_option(self._ctx, _capi.OPTION[option], int(value)), "trm_set_option")

    def get_option(self, option) -> int:
        v = C.c_int()
        self._check(self._lib.trm_get_option(self._ctx, _capi.OPTION[option], C.byref(v)), "trm_get_option")
        return int(v.value)

    def last_program(self) -> dict:
        """Which kernel instance the last step launch of this context selected (TRM_INFO_LAST_PROGRAM), decoded: the selection
        rules of the library are pure host logic on sizes, kinds and options -- a wrong rule costs speed, never correctness, so
        nothing but a look at this id would notice."""
        return _capi.decode_program(self.get_option("info_last_program"))

    def set_stream(self, hip_stream_handle):
        self._check(self._lib.trm_set_stream(self._ctx, C.c_void_p(hip_stream_handle)), "trm_set_stream")

    def synchronize(self):
        self._check(self._lib.trm_synchronize(self._ctx), "trm_synchronize")


# ---- state-dependent forcings and boundary values ------------------------------------
class StateFunction:
    """A boundary value, input or `vwc_forcing` that depends on the model STATE: the mirror of the reference's user
    `Forcing` in discrete form (src/forcings.jl:13-15, `forcing(i, j, k, grid, clock, fields)`) and of boundary conditions
    that read `fields` (src/boundary_conditions.jl:25-28, `getbc(..., clock, fields)`).

    `func(fields, clock, parameters)` is evaluated ON THE DEVICE before every step, vectorised over all columns (the
    reference evaluates the scalar form per cell inside the tendency kernel: the same values at the same state):
      fields.<name>   a torch view of the library's own buffer, zero-copy -- `[num_columns, rows]` for 3-D fields (level k of
                      column i at [i, k], k = 0 the bottom cell), `[num_columns]` for 2-D fields
      clock           `.time`, `.iteration`
    and returns a tensor (or number) that broadcasts to the target: `[num_columns, Nz]` for `vwc_forcing`, `[num_columns]`
    for a boundary value or an input.  Evaluation and the step share one stream, so nothing waits on the host.
    Under Heun the function is evaluated twice per step, as the reference does (heun.jl:37-71): at the state with the clock at
    t, and at the STAGE -- `fields` are then the predicted state's buffers -- with the clock at t + dt, between the library's
    trm_heun_predict and trm_heun_correct."""

    def __init__(self, func, parameters=None):
        self.func = func
        self.parameters = parameters


class _Clock:
    def __init__(self, time, iteration):
        self.time, self.iteration = time, iteration


class _DeviceFields:
    """`fields` of a StateFunction: attribute access -> torch view of the field's device buffer (cached)."""

    def __init__(self, state, torch, stage=False):
        object.__setattr__(self, "_state", state)
        object.__setattr__(self, "_torch", torch)
        object.__setattr__(self, "_views", {})
        object.__setattr__(self, "_stage", stage)      # the Heun stage's buffers instead of the state's

    def __getattr__(self, name):
        if name not in self._views:
            field, row0, rows = self._state._alias(name)      # (e.g. ground_temperature = the top row of temperature)
            arr = self._state.stage_device_array(field) if self._stage else self._state.device_array(field)
            t = self._torch.as_tensor(arr, device=f"cuda:{int(self._state.grid.device)}")
            if t.dim() == 2:
                t = t[:, row0] if (field != name and rows == 1) else t[:, row0:row0 + rows]
            self._views[name] = t
        return self._views[name]


# ---- integrator -------------------------------------------------------------------
class ModelIntegrator:
    """src/timesteppers/model_integrator.jl:10-37"""

    def __init__(self, model, timestepper, state: DeviceState, boundary_conditions, initializers, inputs):
        self.model = model
        self.timestepper = timestepper
        self.state = state
        self.boundary_conditions = boundary_conditions
        self.initializers = initializers
        self.inputs = inputs  # name -> number | array | f(t) returning an array/number
        self._next_level = {}  # windowed FieldTimeSeries: the first level of the record that is not on the device yet
        self._sf = None        # StateFunction targets: [(function, destination view)], bound by initialize_integrator

    @property
    def clock(self):
        return self.state.clock()

    def _apply_time_dependent(self, t, dt=None):
        """update_inputs! + BC functions, evaluated at the pre-tick time (the reference fills halos
        before tick!, forward_euler.jl:19-31).

        Heun evaluates its stage at the ticked stage clock (heun.jl:52-59: tick!(stage.clock) then update_state!(stage)),
        so under Heun a function f is handed over as the two-node series [(t, f(t)), (t + dt, f(t + dt))]: the library
        evaluates it at t for the state and at t + dt for the stage (both nodes are hit exactly)."""
        dyn = self._apply_state_functions(t)
        heun = isinstance(self.timestepper, Heun) and dt is not None
        for (var, side), (kind, value) in self.boundary_conditions.items():
            if callable(value):
                if heun:
                    self.state.set_bc_series(var, side, kind, [t, t + dt], np.stack([np.broadcast_to(np.asarray(value(tt), dtype=np.float64), (self.state.grid.Nh,)) for tt in (t, t + dt)]))
                else:
                    self.state.set_bc(var, side, kind, value(t))
                dyn = True
        for name, value in self.inputs.items():
            if callable(value):
                if heun:
                    self.state.set_forcing_series(name, [t, t + dt], np.stack([np.broadcast_to(np.asarray(value(tt), dtype=np.float64), (self.state.grid.Nh,)) for tt in (t, t + dt)]))
                else:
                    self.state.set_forcing(name, value(t))
                dyn = True
        return dyn

    # -- state-dependent functions: evaluated by torch on the library's stream, written into the library's buffers ----------
    def _state_functions(self):
        out = [(("bc", var, side, kind), v) for (var, side), (kind, v) in self.boundary_conditions.items() if isinstance(v, StateFunction)]
        out += [(("input", name), v) for name, v in self.inputs.items() if isinstance(v, StateFunction)]
        forcing = getattr(getattr(getattr(self.model, "soil", None), "hydrology", None), "vwc_forcing", None)
        if isinstance(forcing, StateFunction):
            out.append((("vwc_forcing",), forcing))
        return out

    def _bind_state_functions(self):
        """Allocates the targets and puts the library on a torch stream (called by initialize_integrator)."""
        fns = self._state_functions()
        self._sf = []
        if not fns:
            return
        import torch  # device memory / stream plumbing of the host mirror
        st = self.state
        dev = f"cuda:{int(st.grid.device)}"
        if getattr(self, "_sf_stream", None) is None:
            self._sf_stream = torch.cuda.Stream(device=dev)
            st.set_stream(self._sf_stream.cuda_stream)
        self._sf_fields = _DeviceFields(st, torch)
        for target, fn in fns:
            if target[0] == "bc":
                _, var, side, kind = target
                st.set_bc(var, side, kind, np.zeros(st.grid.Nh, dtype=st.dtype))
                dest = torch.as_tensor(st.bc_device_array(var, side), device=dev)
            elif target[0] == "input":
                st.set_forcing(target[1], 0.0)
                dest = getattr(self._sf_fields, target[1])
            else:
                st.set("vwc_forcing", np.zeros((st.grid.Nz, st.grid.Nh), dtype=st.dtype))
                dest = self._sf_fields.vwc_forcing
            self._sf.append((fn, dest, target))
        self._sf_stage = None      # the stage's destinations (Heun), bound at the first Heun step

    def _bind_stage_functions(self):
        """The Heun stage's `fields` and the stage's copies of the targets (trm_stage_*_device_ptr)."""
        import torch
        st = self.state
        dev = f"cuda:{int(st.grid.device)}"
        self._sf_stage_fields = _DeviceFields(st, torch, stage=True)
        self._sf_stage = []
        for fn, _, target in self._sf:
            if target[0] == "bc":
                dest = torch.as_tensor(st.stage_bc_device_array(target[1], target[2]), device=dev)
            elif target[0] == "input":
                dest = getattr(self._sf_stage_fields, target[1])
            else:
                dest = self._sf_stage_fields.vwc_forcing
            self._sf_stage.append((fn, dest))

    def _evaluate(self, pairs, fields, clock):
        import torch
        with torch.cuda.stream(self._sf_stream):
            for fn, dest in pairs:
                out = fn.func(fields, clock, fn.parameters)
                dest.copy_(torch.as_tensor(out, device=dest.device, dtype=dest.dtype).expand_as(dest))

    def _apply_state_functions(self, t):
        if not self._sf:
            return False
        self._evaluate([(fn, dest) for fn, dest, _ in self._sf], self._sf_fields, _Clock(t, self.state.clock()[1]))
        return True

    def _heun_step_with_state_functions(self, dt, finalize):
        """One Heun step with the functions evaluated at both stages (heun.jl:37-71): the state's values have been written by
        _apply_time_dependent; predict, evaluate at the stage (clock t + dt), correct."""
        if self._sf_stage is None:
            self._bind_stage_functions()
        t, it = self.state.clock()
        self.state.heun_predict(dt)
        # The stage's clock has ticked (heun.jl:52: time t + dt, iteration + 1).  Boundary values and inputs are evaluated where the
        # reference fills halos / updates inputs -- BEFORE compute_auxiliary!(stage): the stage's auxiliary fields still hold the
        # state's copies (copyto!, heun.jl:45) --, the forcing where the reference evaluates it -- inside compute_tendencies!(stage),
        # AFTER compute_auxiliary!(stage) (state_variables.jl:72-80).
        clock = _Clock(t + dt, it + 1)
        early = [(fn, dest) for (fn, dest), (_, _, target) in zip(self._sf_stage, self._sf) if target[0] != "vwc_forcing"]
        late = [(fn, dest) for (fn, dest), (_, _, target) in zip(self._sf_stage, self._sf) if target[0] == "vwc_forcing"]
        if early:
            self._evaluate(early, self._sf_stage_fields, clock)
        if late:
            self.state.heun_stage_auxiliary()
            self._evaluate(late, self._sf_stage_fields, clock)
        self.state.heun_correct(dt, finalize)

    # -- windowed series: a record streamed through a fixed device window ------------------------------------------------
    def _windowed(self):
        out = [((var, side), v) for (var, side), (kind, v) in self.boundary_conditions.items() if isinstance(v, FieldTimeSeries) and v.window]
        out += [(name, v) for name, v in self.inputs.items() if isinstance(v, FieldTimeSeries) and v.window]
        return out

    def _feed(self, t, dt, nsteps):
        """Moves the device windows forward to time `t` (trim, then append as many levels as the window has room for --
        the copy runs on a side stream under the steps that follow) and returns how many of the next `nsteps` steps the
        windows cover.  The last of k steps evaluates its inputs at t + (k - 1) dt, a Heun stage at t + k dt."""
        ok = nsteps
        series = self._windowed()
        if not series:
            return ok
        self.state.series_trim_before(t)
        for target, fts in series:
            info = self.state.series_info(target)
            nxt = self._next_level[id(fts)]
            room = fts.window - info["levels"]
            if room > 0 and nxt < fts.times.size:
                hi = min(nxt + room, fts.times.size)
                self.state.series_append(target, fts.times[nxt:hi], np.asarray(fts.values[nxt:hi]))
                self._next_level[id(fts)] = hi
                info = self.state.series_info(target)
            if self._next_level[id(fts)] < fts.times.size:      # (beyond the end of the record the window IS the record's end)
                k = int(np.floor((info["t_last"] - t) / dt + 1e-9))
                if k < 1:
                    raise ValueError(f"the window of {fts.window} levels of {target} does not cover one step of {dt} s at t = {t}")
                ok = min(ok, k)
        return ok

    def _rewind_windows(self, t):
        """A restore that goes BACK in time in a used integrator: a device window whose head has been trimmed past `t` no longer
        holds the bracket of `t` -- the series starts over from the head of its record (`_seek_windows` then moves it forward)."""
        for target, fts in self._windowed():
            info = self.state.series_info(target)
            if info["t_first"] <= t or fts.times[0] > t and info["t_first"] == fts.times[0]:
                continue
            n = min(fts.window, fts.times.size)
            if isinstance(target, tuple):
                var, side = target
                kind = self.boundary_conditions[(var, side)][0]
                self.state.set_bc_series(var, side, kind, fts.times[:n], np.asarray(fts.values[:n]), fts.time_indexing)
            else:
                self.state.set_forcing_series(target, fts.times[:n], np.asarray(fts.values[:n]), fts.time_indexing)
            self.state.series_window(target, fts.window)
            self._next_level[id(fts)] = n

    def _seek_windows(self, t):
        """Moves the device windows forward until they hold the bracket of time `t` (a restart into a fresh integrator, whose
        windows start at the head of their records)."""
        for target, fts in self._windowed():
            while True:
                info = self.state.series_info(target)
                nxt = self._next_level[id(fts)]
                if info["t_last"] > t or nxt >= fts.times.size:
                    break
                self.state.series_trim_before(min(t, info["t_last"]))
                info = self.state.series_info(target)
                hi = min(nxt + max(1, fts.window - info["levels"]), fts.times.size)
                self.state.series_append(target, fts.times[nxt:hi], np.asarray(fts.values[nxt:hi]))
                self._next_level[id(fts)] = hi

    def _step(self, dt, nsteps, finalize, heun=None):
        """`nsteps` steps in as few library calls as the device windows of the time series allow (one, without windows)."""
        heun = isinstance(self.timestepper, Heun) if heun is None else heun
        if heun and self._sf:
            assert nsteps == 1, "state-dependent functions are evaluated before every step"
            if self._windowed():
                self._feed(self.state.clock()[0], dt, 1)
            return self._heun_step_with_state_functions(dt, finalize)
        stepper = self.state.step_heun if heun else self.state.step
        if not self._windowed():
            return stepper(dt, nsteps, finalize=finalize)
        done = 0
        while done < nsteps:
            k = self._feed(self.state.clock()[0], dt, nsteps - done)
            stepper(dt, k, finalize=finalize and done + k == nsteps)
            done += k

    def _has_time_dependence(self):
        return bool(self._sf) or any(callable(v) for _, v in self.boundary_conditions.values()) or \
            any(callable(v) for v in self.inputs.values())


def current_time(integrator: ModelIntegrator) -> float:
    return integrator.state.clock()[0]


def _apply_model_initializer(state: DeviceState, model):
    init = model.initializer
    if isinstance(init, M.DefaultInitializer) or init is None:
        return
    zc = state.z_centers()
    hyd = init.hydrology if isinstance(init, M.SoilInitializer) else None
    en = init.energy if isinstance(init, M.SoilInitializer) else init
    if isinstance(hyd, M.ConstantSaturation):
        state.set("saturation_water_ice", hyd.sat)
    elif isinstance(hyd, M.SaturationWaterTable):
        # soil_model_init.jl:149-152 compares z <= +depth, i.e. always true (SURVEY C-2): reproduced
        state.set("saturation_water_ice", np.where(zc <= hyd.water_table_depth, 1.0, hyd.vadose_zone_saturation))
    if isinstance(en, M.ConstantSoilTemperature):
        state.set("temperature", en.T0)
    elif isinstance(en, M.QuasiThermalSteadyState):
        state.set("temperature", en.T0 - en.Qgeo / en.k_eff * zc)
    elif isinstance(en, M.PiecewiseLinearInitialSoilTemperature):
        state.set("temperature", M.piecewise_linear(*en.knots)(zc))


def initialize(model, timestepper=None, boundary_conditions=None, initializers=None, inputs=None) -> ModelIntegrator:
    """initialize(model, timestepper; boundary_conditions, initializers) (model_integrator.jl:145-161).

    `boundary_conditions`: dict {(variable, side): (kind, value)} as built by the aliases above; a value may
    be a number, a per-column array, a `FieldTimeSeries` (evaluated on the device every step) or a function of
    time returning a number / array (evaluated on the host before every step: one library call per step).
    `initializers`: dict {field name: number | array | function(x, z)} (`set!` arguments).
    `inputs`: dict {input field: number | array | FieldTimeSeries | function(t)} (`InputSources(InputSource(...))`)."""
    timestepper = timestepper or ForwardEuler()
    bcs = dict(boundary_conditions or {})
    inits = dict(initializers or {})
    inputs = dict(inputs or {})
    state = DeviceState(model.grid, M.flatten(model))
    if isinstance(model, M.VegetationModel):
        state.set_vegetation(M.flatten_vegetation(model.vegetation, model.constants), "standalone")
    elif getattr(model, "vegetation", None) is not None:    # LandModel(grid; soil, vegetation) (land_model.jl:24-34)
        state.set_vegetation(M.flatten_vegetation(model.vegetation, model.constants, model.soil.hydrology.hydraulic_properties,
                                                  model.soil.strat.texture, model.surface_hydrology), "coupled")
    integ = ModelIntegrator(model, timestepper, state, bcs, inits, inputs)
    initialize_integrator(integ)
    return integ


def initialize_integrator(integ: ModelIntegrator):
    """initialize!(integrator) (model_integrator.jl:96-109): reset, inputs, user initializers,
    model initializer, process initializers."""
    st = integ.state
    st.reset()           # reset!(state), reset!(clock): every prognostic / auxiliary / tendency field to zero, clock to 0
    st.clear_series()
    grid = getattr(integ.model, "grid", None)
    if hasattr(grid, "mask_index") and not hasattr(st, "ring_points"):
        st.set_ring_grid(grid.mask.size, grid.mask_index)   # ColumnRingGrid: scatter / gather run on the device

    def head(fts):      # a windowed series starts with the first `window` levels of its record
        n = fts.times.size if not fts.window else min(fts.window, fts.times.size)
        integ._next_level[id(fts)] = n
        return fts.times[:n], np.asarray(fts.values[:n])

    for (var, side), (kind, value) in integ.boundary_conditions.items():
        if isinstance(value, RasterInputSource):
            value.attach_boundary(st, var, side, kind)     # a named input variable as boundary value (soil_heat_global_era5.jl:31-44)
        elif isinstance(value, FieldTimeSeries):
            st.set_bc_series(var, side, kind, *head(value), value.time_indexing)
            if value.window:
                st.series_window((var, side), value.window)
        elif not isinstance(value, StateFunction):
            st.set_bc(var, side, kind, value(0.0) if callable(value) else value)
    for name, value in integ.inputs.items():
        if isinstance(value, RasterInputSource):
            value.name = name
            value.attach(st)      # static raster: set once; time-indexed: device-resident series (ext/TerrariumRastersExt)
        elif isinstance(value, FieldTimeSeries):
            st.set_forcing_series(name, *head(value), value.time_indexing)
            if value.window:
                st.series_window(name, value.window)
        elif not isinstance(value, StateFunction):
            st.set_forcing(name, value(0.0) if callable(value) else value)
    integ._bind_state_functions()
    st.update_inputs()   # initialize!(fields, source, clock) = update_inputs! at the start time
    soil = getattr(integ.model, "soil", None)
    forcing = getattr(getattr(soil, "hydrology", None), "vwc_forcing", None)
    if forcing is not None and not isinstance(forcing, (int, float, StateFunction)):
        st.set("vwc_forcing", forcing)   # per-cell user forcing (soil_hydrology.jl:37-38)
    for name, value in integ.initializers.items():
        st.set(name, value)
    if isinstance(integ.model, M.VegetationModel):
        return integ                 # (no process initialisers: initialize!(state, model::VegetationModel) is the default no-op)
    _apply_model_initializer(st, integ.model)
    st.initialize()
    return integ


def timestep(integ: ModelIntegrator, dt: Optional[float] = None, finalize: bool = True):
    """timestep!(integrator, dt; finalize) (model_integrator.jl:124-131)"""
    dt = integ.timestepper.dt if dt is None else float(dt)
    integ._apply_time_dependent(current_time(integ), dt)
    integ._step(dt, 1, finalize)


def run(integ: ModelIntegrator, steps: Optional[int] = None, period: Optional[float] = None, dt: Optional[float] = None):
    """run!(integrator; steps, period, dt) (model_integrator.jl:72-88); `period` in seconds."""
    dt = integ.timestepper.dt if dt is None else float(dt)
    if steps is None and period is None:
        raise ValueError("either `steps` or `period` must be specified")
    if steps is not None and period is not None:
        raise ValueError("both `steps` and `period` cannot be specified")
    if steps is None:
        steps = int(period // dt)
    if integ._has_time_dependence():
        for n in range(steps):
            integ._apply_time_dependent(current_time(integ), dt)
            integ._step(dt, 1, finalize=(n == steps - 1))
    elif steps > 0:
        integ._step(dt, steps, finalize=True)
    if steps == 0:
        integ.state.compute_auxiliary()  # run! always ends with compute_auxiliary! (model_integrator.jl:85-86)
    return integ


# reference-named process interface on an integrator's state
def compute_auxiliary(state: DeviceState, model=None): state.compute_auxiliary()
def compute_tendencies(state: DeviceState, model=None): state.compute_tendencies()
def closure(state: DeviceState, model=None): state.closure()
def invclosure(state: DeviceState, model=None): state.invclosure()
def update_state(integ: ModelIntegrator, compute_tendencies=True): integ.state.update_state(compute_tendencies)


# ---- restart ("download/upload of all prognostic buffers + clock is sufficient for restart";
# docs/src/running/time_stepping.md:97-139 saves and reloads the state of a simulation) ------------------------------------------
# The restart set: every prognostic variable, the closure variables that belong to them (stored, not recomputed: a restart
# continues bit for bit), the water table, and -- with vegetation -- the net assimilation the stomatal conductance carries from
# the previous evaluation (vegetation_carbon.jl:88-92).  Everything else is recomputed by the next step (auxiliaries) or is the
# caller's (inputs, boundary values, series).
RESTART_FIELDS_SOIL = ("internal_energy", "temperature", "liquid_water_fraction", "saturation_water_ice")
RESTART_FIELDS_RICHARDS = ("pressure_head", "surface_excess_water", "water_table")
RESTART_FIELDS_LAND = ("skin_temperature",)
RESTART_FIELDS_VEGETATION = ("carbon_vegetation", "vegetation_area_fraction", "net_assimilation")
RESTART_FIELDS_CANOPY = ("canopy_water",)


def restart_fields(integ: ModelIntegrator):
    st = integ.state
    mode = getattr(st, "vegetation_mode", "off")
    if mode == "standalone":
        return list(RESTART_FIELDS_VEGETATION)
    names = list(RESTART_FIELDS_SOIL)
    if st.params.flow == _capi.FLOW["richards"]:
        names += RESTART_FIELDS_RICHARDS
    if st.params.seb:
        names += RESTART_FIELDS_LAND
    if mode == "coupled":
        names += RESTART_FIELDS_VEGETATION + RESTART_FIELDS_CANOPY
    return names


def checkpoint(integ: ModelIntegrator) -> dict:
    """The restart set of an integrator as host arrays (trm_download) + the clock + the status word."""
    st = integ.state
    t, it = st.clock()
    return dict(time=t, iteration=it, status=st.status(), dtype=str(np.dtype(st.dtype)), Nh=st.grid.Nh, Nz=st.grid.Nz,
                fields={name: st.get(name) for name in restart_fields(integ)})


def restore(integ: ModelIntegrator, ckpt: dict) -> ModelIntegrator:
    """Continues from `checkpoint(...)` in THIS integrator -- typically a fresh one (a new process, another device): the
    fields are uploaded (trm_upload), the clock is set (trm_set_clock), windowed series are moved forward to the clock.
    Initialise the integrator as for a cold start first (boundary conditions, inputs, series); its initial state is replaced."""
    st = integ.state
    if (ckpt["Nh"], ckpt["Nz"], ckpt["dtype"]) != (st.grid.Nh, st.grid.Nz, str(np.dtype(st.dtype))):
        raise ValueError("the checkpoint belongs to another grid / precision")
    missing = [n for n in restart_fields(integ) if n not in ckpt["fields"]]
    if missing:
        raise ValueError(f"the checkpoint lacks {missing}")
    for name in restart_fields(integ):
        st.set(name, ckpt["fields"][name])
    st.set_clock(ckpt["time"], ckpt["iteration"])
    st.set_status(ckpt.get("status", 0))      # (sticky flags raised before the checkpoint stay raised after the restart)
    integ._rewind_windows(ckpt["time"])
    integ._seek_windows(ckpt["time"])
    st.update_inputs()      # the inputs as update_inputs! leaves them at the restored clock
    return integ


# ---- one host process, several devices ("1 process x 8 HIP devices") ---------------------------------------
class DeviceGroup:
    """The shards of one grid, one `DeviceState` per device, driven from ONE host thread -- the reference's host is one Julia
    process (column_grid.jl:32, model_integrator.jl:72-88).  `step` enqueues the steps on every device without waiting in
    between and waits once (trm_step_all); global diagnostics combine the shards inside the library (trm_reduce_global_all /
    trm_status_global_all: grouped RCCL all-reduces once `comm_init` has run, a host fold otherwise)."""

    def __init__(self, states):
        self.states = list(states)
        self._lib = _capi.lib()
        self._arr = (C.c_void_p * len(self.states))(*[s._ctx for s in self.states])

    def __len__(self):
        return len(self.states)

    def _check(self, rc, what):
        if rc != 0:
            for s in self.states:
                msg = self._lib.trm_last_error(s._ctx)
                if msg:
                    break
            err = _capi.TerrariumHipError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")
            err.code = rc
            raise err

    def comm_init(self):
        """One RCCL communicator per context inside one group (trm_comm_init_all); the contexts must sit on distinct devices."""
        self._check(self._lib.trm_comm_init_all(self._arr, len(self)), "trm_comm_init_all")

    def step(self, dt, nsteps=1, finalize=True, heun=False):
        fn = self._lib.trm_step_heun_all if heun else self._lib.trm_step_all
        self._check(fn(self._arr, len(self), float(dt), int(nsteps), int(finalize)), "trm_step_all")

    def synchronize(self):
        self._check(self._lib.trm_synchronize_all(self._arr, len(self)), "trm_synchronize_all")

    def reduce_global(self, name, op) -> np.ndarray:
        rows = 1 if op == "volume_integral_z" else self.states[0].rows(name)
        out = np.zeros(rows, dtype=np.float64)
        self._check(self._lib.trm_reduce_global_all(self._arr, len(self), _capi.FIELD[name], _capi.REDUCE[op], out.ctypes.data), "trm_reduce_global_all")
        return out

    def status_global(self) -> int:
        f = C.c_uint32()
        self._check(self._lib.trm_status_global_all(self._arr, len(self), C.byref(f)), "trm_status_global_all")
        return int(f.value)

    def gather(self, name) -> np.ndarray:
        """The field on the whole grid: the shards' columns side by side."""
        return np.concatenate([np.atleast_2d(s.get(name)) for s in self.states], axis=-1)


# ---- the small interface functions the reference exports (src/timesteppers/*.jl, model_integrator.jl:39-66, grids) ----------
def default_dt(timestepper) -> float:
    """default_dt(timestepper) (forward_euler.jl:13, heun.jl:16)"""
    return float(timestepper.dt)


def is_adaptive(timestepper) -> bool:
    """is_adaptive(timestepper) (forward_euler.jl:15, heun.jl:18): both explicit steppers use a fixed step."""
    return False


def iteration(integ: ModelIntegrator) -> int:
    """Oceananigans.Solvers.iteration(integrator) (model_integrator.jl:55)"""
    return int(integ.state.clock()[1])


def time_step(integ: ModelIntegrator, dt: Optional[float] = None, **kwargs):
    """Oceananigans.TimeSteppers.time_step!(integrator, dt) = timestep!(integrator, dt) (model_integrator.jl:62-64)"""
    return timestep(integ, dt)


def reset(integ: ModelIntegrator):
    """initialize!(integrator) (model_integrator.jl:96-109): reset!(state) -- every prognostic, auxiliary and tendency field
    to zero (state_variables.jl:102-120) --, clock back to zero, inputs, initializers, process initialisers."""
    return initialize_integrator(integ)


def get_grid(model):
    return model.grid


def znodes(obj, location="center") -> np.ndarray:
    """znodes(field) / znodes(grid, Center() | Face()): the vertical coordinates, bottom cell first (negative downwards)."""
    grid = getattr(obj, "grid", obj)
    grid = getattr(grid, "grid", grid) if not hasattr(grid, "z_centers") else grid
    return grid.z_faces() if str(location).lower().startswith("f") else grid.z_centers()


def zspacings(obj) -> np.ndarray:
    """zspacings(grid, Center()): the layer thicknesses, bottom cell first."""
    grid = getattr(obj, "grid", obj)
    grid = getattr(grid, "grid", grid) if not hasattr(grid, "thickness") else grid
    return np.asarray(grid.thickness, dtype=np.float64)[::-1].copy()
