#!/usr/bin/env python3
"""Benchmark of the fused SoilModel step on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (update_state! + explicit_step! + closure!,
forward_euler.jl:19-31) over every column of the workload: ONE launch of the fused
HIP kernel, reading and writing the whole state through HBM -- no temporal
blocking, state is not kept resident across steps.  The headline workload (N=1)
is BASELINE.json config C3: global N145 ERA5-land mask (56 951 columns) x 32 soil
levels, coupled heat + Richards water transport, fp64.  With --gpus N every rank
holds a full N145-sized shard (weak scaling: columns are independent, there is no
collective on the step path).  Rank 0 prints ONE JSON line.

The CPU oracle is used here ONLY for the `cpu_baseline` leg (a timed, bounded
sample of the same workload on the host cores) -- never for the measured path.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np

STABLE_STEPS = 150   # see main(): longest stretch the explicit Richards scheme is stepped from one state (W <= 100 before it)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling

WORKLOADS = {
    # name: (description, config, hydraulics, mask/columns, Nz, dtype)
    "c2": ("C2: N72 ERA5-land mask (14017 columns) x 30 levels, soil heat conduction only, fp64", "heat", "default", "N72", 30, "f64"),
    "c2n145": ("C2 physics on the N145 mask (56951 columns) x 30 levels: soil heat conduction only, fp64", "heat", "default", "N145", 30, "f64"),
    "c3": ("C3: N145 ERA5-land mask (56951 columns) x 32 levels, coupled heat + Richards (BrooksCorey SWRC, linear K), fp64", "richards", "default", "N145", 32, "f64"),
    "c3vg": ("C3-VG: N145 mask x 32 levels, heat + Richards (VanGenuchten SWRC + Mualem K with ice impedance), fp64", "richards", "vg", "N145", 32, "f64"),
    "c4": ("C4: N145 mask x 32 levels, full bare-ground LandModel (heat + Richards + surface energy balance, PrescribedAtmosphere), default hydraulics (BrooksCorey SWRC, linear K), fp64", "land", "default", "N145", 32, "f64"),
    "c4vg": ("C4-VG: as C4 with the land-model test's hydraulics (VanGenuchten(alpha=2, n=2) SWRC + Mualem K with ice impedance), fp64", "land", "vg", "N145", 32, "f64"),
    "c5": ("C5: synthetic 0.1-degree grid, 812500 columns per GPU x 64 levels, heat + Richards + SEB, default hydraulics, fp32", "land", "default", 812500, 64, "f32"),
    "c5vg": ("C5-VG: as C5 with VanGenuchten SWRC + Mualem K, fp32", "land", "vg", 812500, 64, "f32"),
}


def algorithmic_bytes_per_column_step(config, Nz, wordsize):
    """compulsory reads + reference-visible writes per column per step."""
    if config == "heat":
        return wordsize * (5 * Nz + 1)          # U,sat read; U,T,liq written; T_ub
    b = wordsize * (8 * Nz + 4)                 # U,sat read; U,sat,T,liq,psi,K written; S r+w, water_table, K top face
    if config == "land":
        b += wordsize * 18                      # 7 forcing reads, T_s r+w, 9 flux/diagnostic writes
    return b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--kernel", default="fused", choices=["fused", "unfused"])
    ap.add_argument("--integrator", default="euler", choices=["euler", "heun"], help="ForwardEuler (headline) or Heun (two fused launches per step)")
    ap.add_argument("--series", action="store_true", help="drive the time-dependent boundary value / atmospheric inputs from device-resident time series (forcing feed) instead of constants")
    ap.add_argument("--spinup-ms", type=float, default=200.0, help="untimed device-busy time after the warm-up steps that lets the clocks settle (0: off)")
    ap.add_argument("--skip-kf", action="store_true", help="store hydraulic_conductivity only when finalizing")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target wall time of the CPU baseline sample")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP library has no CPU fallback")
    # Rehearsal knobs (not used by the driver): TRM_BENCH_BACKEND=gloo + TRM_BENCH_SHARE_DEVICE=1 run the N > 1 code
    # path with several ranks on ONE GPU (RCCL refuses two ranks per device).
    backend = os.environ.get("TRM_BENCH_BACKEND", "nccl")
    if os.environ.get("TRM_BENCH_SHARE_DEVICE"):
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    n_gpus = world

    import workloads as W
    import terrarium_jl_amd as trm
    from terrarium_jl_amd import parallel

    desc, config, hydraulics, columns, Nz, dt_name = WORKLOADS[args.workload]
    dtype = np.float64 if dt_name == "f64" else np.float32
    if isinstance(columns, str):
        lat, lon = W.columns_from_mask(columns)
    else:
        lat, lon = W.synthetic_columns(columns)
    if os.environ.get("TRM_BENCH_SHARD_OF"):    # rehearsal: the shard one rank of an N-way strong-scaling run would hold
        lo, hi = parallel.shard_range(lat.size, int(os.environ["TRM_BENCH_SHARD_OF"]), 0)
        lat, lon = lat[lo:hi], lon[lo:hi]
    if args.scaling == "strong" and world > 1:
        lo, hi = parallel.shard_range(lat.size, world, rank)
        lat, lon = lat[lo:hi], lon[lo:hi]
    w = W.make_workload(config, lat, lon, Nz, dtype=dtype, hydraulics=hydraulics)
    Nh = w["Nh"]

    dev = W.setup_device(w, device=local_rank)
    dev.set_option("step_kernel", args.kernel)
    if args.series:
        # the SURVEY 8(d) diurnal cycle sampled every 10 minutes over the run, linear in between
        DAY = 86400.0
        nodes = np.arange(0.0, (args.steps + args.warmup + 2) * w["dt"] + 600.0, 600.0)
        ph = 2 * np.pi * nodes[:, None] / DAY - w["lon"][None, :]
        if config == "land":
            dev.set_forcing_series("air_temperature", nodes, w["T0"][None, :] + 5.0 * np.sin(ph))
            dev.set_forcing_series("surface_shortwave_down", nodes, np.maximum(0.0, 600.0 * np.sin(ph)))
        else:
            dev.set_bc_series("temperature", "top", "value", nodes, w["T0"][None, :] + 10.0 * np.sin(ph))
    if args.skip_kf:
        dev.set_option("write_kf_every_step", 0)
    dt = w["dt"]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # warmup (untimed)
    heun = args.integrator == "heun"
    if args.warmup > 0:
        (dev.step_heun if heun else dev.step)(dt, min(args.warmup, 100), finalize=False)
    # The explicit Richards scheme at dt = 60 s dries the top cells of this synthetic state to sat = 0 (psi = -Inf, then
    # NaN) after ~270 steps -- in the reference as well.  Runs longer than STABLE_STEPS therefore go back to a device-side
    # snapshot of the warmed-up state every STABLE_STEPS steps (a D2D copy inside the timed wall clock, no host data):
    # every step still does its full work on a valid state.
    chunked = config != "heat" and args.steps > STABLE_STEPS
    if chunked or args.spinup_ms > 0:
        dev.save_state()
    # Clock spin-up (untimed, part of the warm-up): an MI355X that comes out of idle needs ~30 ms of sustained load before
    # its clocks settle -- the same 100 steps from the same state take 30.1 us/step right after start-up and 27.4 us/step
    # from the tenth repetition on (profiles/tools/ramp.py).  A land-surface run is hours of sustained stepping, so the
    # timed region should see the settled clocks: the warmed-up state is stepped and restored until the device has
    # been busy for --spinup-ms, then restored once more.  The timed K steps start from exactly the state W steps produced.
    spun = 0.0
    while spun < args.spinup_ms:
        if heun:
            t1 = time.perf_counter()
            dev.step_heun(dt, STABLE_STEPS, finalize=False)
            torch.cuda.synchronize()
            spun += (time.perf_counter() - t1) * 1e3
        else:
            spun += dev.step_timed(dt, STABLE_STEPS, finalize=False)
        dev.restore_state()
    barrier()
    t0 = time.perf_counter()
    ms, done = 0.0, 0
    while done < args.steps:
        n = min(args.steps - done, STABLE_STEPS) if chunked else args.steps
        if chunked and done > 0:
            dev.restore_state()
        if heun:    # no event-timed entry point for Heun: the wall clock of the synchronous call stands in
            t1 = time.perf_counter()
            dev.step_heun(dt, n, finalize=False)
            torch.cuda.synchronize()
            ms += (time.perf_counter() - t1) * 1e3
        else:
            ms += dev.step_timed(dt, n, finalize=False)  # n steps = n launches, HIP events on the library's stream
        done += n
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    status = dev.status()

    # max over ranks, total columns over ranks
    stats = torch.tensor([elapsed, ms, float(Nh)], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    if world > 1:
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed, ms, total_columns = float(mx[0]), float(mx[1]), float(sm[2])
        nan_flag = parallel.global_status(status)
    else:
        total_columns = float(Nh)
        nan_flag = status
    wordsize = 8 if dtype == np.float64 else 4
    bytes_per_colstep = algorithmic_bytes_per_column_step(config, Nz, wordsize)
    kernel_s = ms * 1e-3 / max(args.steps, 1)            # average duration of one step launch on this GPU
    achieved_gbs = bytes_per_colstep * Nh / kernel_s / 1e9
    value = total_columns * args.steps / elapsed

    out = {
        "metric": "column-steps/sec",
        "value": value,
        "unit": "column-steps/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed * 1e3 / max(args.steps, 1),
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": dt_name,
        "data": "synthetic forcing and initial state on the reference's ERA5-land mask columns (SURVEY 8(d)); seeded",
        "config": {"workload": desc, "columns_per_gpu": Nh, "levels": Nz, "dt_s": dt, "kernel": args.kernel, "integrator": args.integrator, "series": bool(args.series), "state_restored_every": STABLE_STEPS if chunked else None,
                   "clock_spinup_ms": args.spinup_ms,
                   "parallelism": f"columns block-sharded over {n_gpus} GPU(s), no data-path collective",
                   "status_flags": int(nan_flag)},
        "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": None,
                     "kernel": "k_step_wave" if args.kernel == "fused" else "unfused sequence",
                     "kernel_ms": kernel_s * 1e3, "algorithmic_bytes_per_column_step": bytes_per_colstep},
    }

    # HBM traffic of the dominant kernel from the PMC counters: collected in separate rocprofv3 --pmc passes of
    # this same command (FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE), summary committed under profiles/
    pmc = os.path.join(ROOT, "profiles", "r01", "pmc_summary_c3_fused.json")
    if args.workload == "c3" and args.kernel == "fused" and os.path.exists(pmc):
        with open(pmc) as f:
            out["roofline"]["traffic"] = json.load(f)["hbm_traffic_bytes_per_launch"]
        out["roofline"]["traffic_source"] = "profiles/r01/pmc_summary_c3_fused.json (bytes per launch)"
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(W, w, args.cpu_seconds)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(W, w, target_seconds):
    """CPU restatement of the reference path (oracle/, OpenMP over columns, reference kernel order) timed on
    this box's host cores on the same workload: whole column set, bounded number of steps.  The thread count is
    auto-tuned over a few candidates (the reference-order passes are memory-bound and stop scaling long before
    all hardware threads are busy); `cores` reports the count actually used for the quoted number."""
    import oracle
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    candidates = sorted({c for c in (8, 16, 32, 64, 128, avail) if c <= avail} | {min(avail, 8)})
    budget = target_seconds
    best = None
    for threads in candidates:
        oracle.set_threads(threads)
        orc = W.setup_oracle(w, omp=True)
        orc.steps(w["dt"], 2)  # touch pages / spin up the thread pool
        t0 = time.perf_counter()
        orc.steps(w["dt"], 10)
        dt = time.perf_counter() - t0
        budget -= dt
        rate = w["Nh"] * 10 / dt
        if best is None or rate > best[1]:
            best = (threads, rate)
        if budget < target_seconds * 0.5:
            break
    threads = best[0]
    oracle.set_threads(threads)
    orc = W.setup_oracle(w, omp=True)
    orc.steps(w["dt"], 2)
    chunk, done, spent = 10, 0, 0.0
    while spent < max(budget, 2.0) and done < 80:
        t0 = time.perf_counter()
        orc.steps(w["dt"], chunk)
        spent += time.perf_counter() - t0
        done += chunk
    return {"value": w["Nh"] * done / spent, "unit": "column-steps/s", "cores": threads, "kind": "port",
            "sample": f"all {w['Nh']} columns x {done} steps of the same workload; CPU restatement of the Terrarium.jl "
                      f"path (not Terrarium.jl itself: Julia is not installed), one pass per reference kernel, OpenMP "
                      f"over columns with {threads} of {avail} hardware threads (best of {candidates}); {spent:.1f} s"}


if __name__ == "__main__":
    main()
