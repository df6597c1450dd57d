"""`Simulation(integrator; Δt, stop_time)` + `run!(sim)` on the host mirror.

In the reference `ModelIntegrator` implements Oceananigans' `AbstractModel` interface (`time_step!`, `update_state!`,
`iteration`, `time`: src/timesteppers/model_integrator.jl:39-66) so that an Oceananigans `Simulation` can drive it with
callbacks and output writers (examples/simulations/soil_heat_global.jl:117-123, examples/extending/*.jl).  This module is
that driver for the device-resident integrator: schedules (`IterationInterval`, `TimeInterval`), callbacks, and an output
writer that takes snapshots of state variables -- optionally scattered to the full ring grid
(`RingGrids.Field(field, grid)`, src/grids/column_ring_grid.jl:102-115).

The loop hands the library as many steps per call as fit before the next scheduled event (one `trm_step` /
`trm_step_heun` call; with TRM_OPT_STEPS_PER_LAUNCH the columns stay in registers across them); time steps are aligned
to `stop_time` and to `TimeInterval` schedules the way Oceananigans' `aligned_time_step` does (the step that would
overshoot is shortened)."""
import math
from typing import Callable, Dict, Optional, Sequence

import numpy as np

from . import integrator as _integ


class IterationInterval:
    """Actuates every `interval` iterations (Oceananigans.Utils.IterationInterval)."""

    def __init__(self, interval: int, offset: int = 0):
        assert interval >= 1
        self.interval, self.offset = int(interval), int(offset)

    def actuates(self, time, iteration):
        return (iteration - self.offset) % self.interval == 0

    def steps_until_next(self, time, iteration, dt):
        r = (iteration - self.offset) % self.interval
        return self.interval - r

    def next_time(self, time):
        return math.inf


class TimeInterval:
    """Actuates whenever the model time passes a multiple of `interval` seconds (Oceananigans.Utils.TimeInterval)."""

    def __init__(self, interval: float):
        assert interval > 0
        self.interval = float(interval)
        self.first = 0.0
        self.actuations = 0

    def next_time(self, time=None):
        return self.first + (self.actuations + 1) * self.interval

    def actuates(self, time, iteration):
        if time >= self.next_time():
            self.actuations = int(math.floor((time - self.first) / self.interval))
            return True
        return False

    def steps_until_next(self, time, iteration, dt):
        return max(1, int(math.ceil((self.next_time() - time) / dt - 1e-12)))


class Callback:
    """Callback(func, schedule): `func(sim)` whenever the schedule actuates (Oceananigans.Simulations.Callback)."""

    def __init__(self, func: Callable, schedule=None):
        self.func, self.schedule = func, schedule or IterationInterval(1)


class SnapshotWriter:
    """Output writer (stand-in for Oceananigans' JLD2Writer / NetCDFWriter): on every actuation of `schedule` the named
    state variables are downloaded and kept as `[n_snapshots][rows][Nh]` (or scattered to `[rows][nlat][nlon]` when
    `ring_grid` is given); `write()` stores them with the snapshot times in an .npz file."""

    def __init__(self, fields: Sequence[str], schedule, filename: Optional[str] = None, ring_grid=None, fill=np.nan):
        self.fields, self.schedule, self.filename, self.ring_grid, self.fill = list(fields), schedule, filename, ring_grid, fill
        self.times, self.iterations = [], []
        self.data: Dict[str, list] = {f: [] for f in self.fields}

    def __call__(self, sim):
        t, it = sim.integrator.state.clock()
        self.times.append(t)
        self.iterations.append(it)
        st = sim.integrator.state
        for f in self.fields:
            # only the rows the variable consists of cross PCIe (`ground_temperature` = one row of temperature); with a ring
            # grid the scatter runs on the device (trm_download_ring) and the result arrives as [rows][nlat][nlon]
            if self.ring_grid is not None and hasattr(st, "ring_points") and st.ring_points == self.ring_grid.mask.size:
                a = st.get_ring(f, self.fill)
                self.data[f].append(a.reshape(a.shape[:-1] + self.ring_grid.mask.shape))
            else:
                a = st.get(f)
                self.data[f].append(self.ring_grid.scatter(a, self.fill) if self.ring_grid is not None else a)

    def write(self):
        if self.filename:
            np.savez(self.filename, time=np.array(self.times), iteration=np.array(self.iterations),
                     **{f: np.stack(v) for f, v in self.data.items() if v})


class Simulation:
    """Simulation(integrator; Δt, stop_time, stop_iteration).  `callbacks` and `output_writers` are dicts, as in
    Oceananigans; `run(sim)` mirrors `run!(sim)`."""

    def __init__(self, integrator, dt: Optional[float] = None, stop_time: float = math.inf, stop_iteration: float = math.inf):
        self.integrator = integrator
        self.dt = float(integrator.timestepper.dt if dt is None else dt)
        self.stop_time, self.stop_iteration = stop_time, stop_iteration
        self.callbacks: Dict[str, Callback] = {}
        self.output_writers: Dict[str, SnapshotWriter] = {}
        self.running = False
        self.initialized = False

    def add_callback(self, func, schedule=None, name=None):
        self.callbacks[name or f"callback{len(self.callbacks) + 1}"] = Callback(func, schedule)

    # -- Oceananigans model interface of the integrator (model_integrator.jl:39-66) --------------------------------------
    @property
    def time(self):
        return self.integrator.state.clock()[0]

    @property
    def iteration(self):
        return self.integrator.state.clock()[1]

    def _events(self):
        return [(cb.schedule, cb.func) for cb in self.callbacks.values()] + [(w.schedule, w) for w in self.output_writers.values()]

    def _fire(self, initial=False, time_only=False):
        """`time_only`: the clock was moved onto an event time WITHOUT a step (the sliver case of `run`): the iteration has not
        advanced, so schedules that depend on the iteration alone have already fired for it and must not fire twice."""
        t, it = self.integrator.state.clock()
        for schedule, func in self._events():
            if initial:
                if isinstance(schedule, TimeInterval):
                    schedule.first, schedule.actuations = t, 0
                func(self)           # Oceananigans evaluates every callback / writer once at initialisation
            elif time_only and not isinstance(schedule, TimeInterval):
                continue
            elif schedule.actuates(t, it):
                func(self)

    def run(self):
        """run!(sim): initialise, then step until `stop_time` / `stop_iteration` (or `sim.running = False` from a callback)."""
        integ = self.integrator
        if not self.initialized:
            self._fire(initial=True)
            self.initialized = True
        self.running = True
        host_dependent = integ._has_time_dependence()
        stepper = lambda dt, n, finalize: integ._step(dt, n, finalize)      # (feeds the device windows of streamed series)
        while self.running:
            t, it = integ.state.clock()
            if t >= self.stop_time or it >= self.stop_iteration:
                break
            # how many full steps until the next event / stop?
            n = min([s.steps_until_next(t, it, self.dt) for s, _ in self._events()] + [10 ** 9])
            n = min(n, self.stop_iteration - it) if math.isfinite(self.stop_iteration) else n
            t_next = min([s.next_time(t) for s, _ in self._events()] + [self.stop_time])
            whole = int(math.floor((t_next - t) / self.dt + 1e-9)) if math.isfinite(t_next) else n
            n = int(max(0, min(n, whole)))
            if math.isfinite(t_next) and abs(t_next - t) <= 1e-9 * self.dt:
                # The device clock accumulates dt step by step: with a dt that is not exactly representable it can land a few
                # ulp short of the target.  Oceananigans' `minimum_relative_step` treats that as reached: no sliver step.
                integ.state.set_clock(t_next, it)
                self._fire(time_only=True)
                continue
            if n >= 1:
                if host_dependent:      # functions of time are evaluated on the host before every step
                    for k in range(n):
                        integ._apply_time_dependent(integ.state.clock()[0], self.dt)
                        stepper(self.dt, 1, finalize=(k == n - 1))
                else:
                    stepper(self.dt, n, finalize=True)
            else:                       # aligned (shortened) step onto the event time
                dt = t_next - t
                if host_dependent:
                    integ._apply_time_dependent(t, dt)
                stepper(dt, 1, finalize=True)
                integ.state.set_clock(t_next, it + 1)      # land exactly on the event time
            self._fire()
        self.running = False
        for w in self.output_writers.values():
            w.write()
        return self


def run_simulation(sim: Simulation) -> Simulation:
    """`run!(sim)`"""
    return sim.run()
