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

Launching N > 1: the driver's form is
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
(one rank per GPU, RANK / LOCAL_RANK / WORLD_SIZE from the environment).  A bare `python bench.py --gpus N` with
no WORLD_SIZE in the environment starts those N ranks itself (fresh child processes, before anything touches the
GPU in the parent) and relays rank 0's line.  --gpus that disagrees with WORLD_SIZE is an error.

The CPU oracle is used here ONLY for the `cpu_baseline` leg (a timed, bounded
sample of the same workload on the host cores) -- never for the measured path.
"""
import argparse
import json
import os
import shutil
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np

# Longest stretch a synthetic state is stepped from one snapshot (see run_timed()).  The explicit Richards scheme at dt = 60 s
# dries top cells out (sat = 0, psi = -Inf, then NaN -- in the reference as well); the first status flag appears within 250
# steps of the warmed-up state for C3, 110 for C4-VG and 80 for C4 / C5 (profiles/tools/first_flag.py, profiles/r03/).
STABLE_STEPS = {"heat": 10 ** 9, "richards": 150, "land": 50, "landveg": 50}
MAX_WARMUP = 40      # warm-up steps executed before the timed stretches (config.warmup_executed reports the count)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling
PROFILE_ROUND = "r05"
# bytes the reference's unfused passes stream per cell and step (count of arrays each kernel reads / writes)
REFERENCE_ORDER_BYTES_PER_CELL_STEP = {"heat": 140, "richards": 300, "land": 300, "landveg": 300}

WORKLOADS = {
    # name: (description, config, hydraulics, mask/columns, Nz, dtype, replicas)
    "c2": ("C2: N72 ERA5-land mask (14017 columns) x 30 levels, soil heat conduction only, fp64", "heat", "default", "N72", 30, "f64", 1),
    "c2n145": ("C2 physics on the N145 mask (56951 columns) x 30 levels: soil heat conduction only, fp64", "heat", "default", "N145", 30, "f64", 1),
    "c3": ("C3: N145 ERA5-land mask (56951 columns) x 32 levels, coupled heat + Richards (BrooksCorey SWRC, linear K), fp64", "richards", "default", "N145", 32, "f64", 1),
    "c3x8": ("C3 physics on 8 copies of the N145 columns (455608 columns x 32 levels, fp64): 0.93 GB of state per step, "
             "3.6x the 256 MiB Infinity Cache -- the same kernel as C3 with every access served by HBM", "richards", "default", "N145", 32, "f64", 8),
    "c3vg": ("C3-VG: N145 mask x 32 levels, heat + Richards (VanGenuchten SWRC + Mualem K with ice impedance), fp64", "richards", "vg", "N145", 32, "f64", 1),
    "c4": ("C4: N145 mask x 32 levels, full bare-ground LandModel (heat + Richards + surface energy balance, PrescribedAtmosphere), default hydraulics (BrooksCorey SWRC, linear K), fp64", "land", "default", "N145", 32, "f64", 1),
    "c4vg": ("C4-VG: as C4 with the land-model test's hydraulics (VanGenuchten(alpha=2, n=2) SWRC + Mualem K with ice impedance), fp64", "land", "vg", "N145", 32, "f64", 1),
    "c4vgveg": ("C4-VG coupled to vegetation: LandModel(vegetation = VegetationCarbon) -- canopy interception, canopy evapotranspiration, plant available water, "
                "photosynthesis / respiration / carbon dynamics -- on the N145 columns x 32 levels, van Genuchten hydraulics, fp64, dt = 0.05 s", "landveg", "vg", "N145", 32, "f64", 1),
    "c5": ("C5: synthetic 0.1-degree grid, 812500 columns per GPU x 64 levels, heat + Richards + SEB, default hydraulics, fp32", "land", "default", 812500, 64, "f32", 1),
    "c5vg": ("C5-VG: as C5 with VanGenuchten SWRC + Mualem K, fp32", "land", "vg", 812500, 64, "f32", 1),
}


def algorithmic_bytes_per_column_step(config, Nz, wordsize):
    """compulsory reads + reference-visible writes per column per step."""
    if config == "heat":
        return wordsize * (5 * Nz + 1)          # U,sat read; U,T,liq written; T_ub
    b = wordsize * (8 * Nz + 4)                 # U,sat read; U,sat,T,liq,psi,K written; S r+w, water_table, K top face
    if config in ("land", "landveg"):
        b += wordsize * 18                      # 7 forcing reads, T_s r+w, 9 flux/diagnostic writes
    if config == "landveg":
        b += wordsize * (2 * Nz + 45)           # sat, liq of the column for the plant available water; ~20 per-column reads, ~25 writes
    return b


def state_bytes_per_step(config, Nz, Nh, wordsize):
    """Bytes of distinct state the fused step touches per launch (what has to fit a cache for it to help)."""
    fields = 3 if config == "heat" else 6       # U, T, liq (+ sat read) / U, sat, T, liq, psi, K
    return (fields + (1 if config == "heat" else 0)) * Nz * Nh * wordsize


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh ranks (nothing has touched the GPU in this process)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # a rank that dies (no such device, ...) must not leave the others waiting in a rendezvous
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        if any(c not in (None, 0) for c in codes):
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.2)
    out = procs[0].stdout.read()
    codes = [p.wait() for p in procs]
    sys.stdout.write(out)
    sys.stdout.flush()
    return max(abs(c) for c in codes)


def build_workload(W, parallel, name, world, rank, scaling):
    desc, config, hydraulics, columns, Nz, dt_name, replicas = WORKLOADS[name]
    dtype = np.float64 if dt_name == "f64" else np.float32
    if isinstance(columns, str):
        lat, lon = W.columns_from_mask(columns)
    else:
        lat, lon = W.synthetic_columns(columns)
    if replicas > 1:
        lat, lon = np.tile(lat, replicas), np.tile(lon, replicas)
    if os.environ.get("TRM_BENCH_SHARD_OF"):    # rehearsal: the shard one rank of an N-way strong-scaling run would hold
        lo, hi = parallel.shard_range(lat.size, int(os.environ["TRM_BENCH_SHARD_OF"]), 0)
        lat, lon = lat[lo:hi], lon[lo:hi]
    if scaling == "strong" and world > 1:
        lo, hi = parallel.shard_range(lat.size, world, rank)
        lat, lon = lat[lo:hi], lon[lo:hi]
    return W.make_workload(config, lat, lon, Nz, dtype=dtype, hydraulics=hydraulics), desc, config, Nz, dt_name


def run_timed(dev, w, config, steps, warmup, spinup_ms, heun, sync, barrier=None, repeats=1):
    """W untimed warm-up steps, an untimed clock spin-up, then `repeats` timed regions of exactly `steps` steps each (one
    launch per step unless the context says otherwise), every region starting from the state the W steps produced (a
    device-side restore between regions, untimed).  Returns (wall seconds per region, device ms of the region's launches
    by HIP events on the library's stream per region, warm-up steps executed, whether a region was chunked)."""
    dt = w["dt"]
    stable = STABLE_STEPS[config]
    warm = min(warmup, MAX_WARMUP, stable // 2)
    step_timed = dev.step_heun_timed if heun else dev.step_timed
    if warm > 0:
        (dev.step_heun if heun else dev.step)(dt, warm, finalize=False)
    dev.save_state()
    # Runs longer than the stable stretch go back to the device-side snapshot of the warmed-up state every `stable` steps (a
    # D2D copy inside the timed wall clock, no host data): every step still does its full work on a valid state.
    chunked = steps > stable
    # Clock spin-up (untimed, part of the warm-up): an MI355X that comes out of idle needs ~30 ms of sustained load before
    # its clocks settle (profiles/tools/ramp.py).  A land-surface run is hours of sustained stepping, so the timed regions
    # should see the settled clocks: the warmed-up state is stepped and restored until the device has been busy for
    # --spinup-ms.
    spun = 0.0
    while spun < spinup_ms:
        spun += step_timed(dt, min(stable, 150), finalize=False)
        dev.restore_state()
    walls, kernels = [], []
    for _ in range(max(1, repeats)):
        dev.restore_state()
        (barrier or sync)()
        t0 = time.perf_counter()
        ms, done = 0.0, 0
        while done < steps:
            n = min(steps - done, stable)
            if done > 0:
                dev.restore_state()
            ms += step_timed(dt, n, finalize=False)  # n steps, HIP events on the library's stream
            done += n
        sync()
        (barrier or sync)()
        walls.append(time.perf_counter() - t0)
        kernels.append(ms)
    return walls, kernels, warm, chunked


def sustained_leg(dev, w, config, seconds, heun, sync):
    """At least `seconds` of back-to-back per-step launches of the measured context (the methodology of the reference's GPU
    benchmark, test/benchmarks/gpu/soil_heat_hydrology_global.jl:58-72: run!(...; period = Hour(1)), i.e. thousands of steps per
    sample): the launches are enqueued asynchronously, the state goes back to the device-side snapshot of the warmed-up state every
    `stable` steps (counted: a D2D copy of every field INSIDE the timed wall clock), the host drains the queue every 16 chunks.
    The short timed regions of the headline are 0.5 ms each -- too short for anything but HIP events to see; this leg is long
    enough for an outside observer (the driver's GPU-busy sampler, a wall clock around the process) to corroborate."""
    dt = w["dt"]
    stable = min(STABLE_STEPS[config], 150)
    step = dev.step_heun if heun else dev.step
    dev.restore_state()
    sync()
    t0 = time.perf_counter()
    for _ in range(20):
        dev.restore_state()
    sync()
    restore_us = (time.perf_counter() - t0) / 20 * 1e6
    dev.set_option("asynchronous", 1)
    steps = restores = 0
    sync()
    t0 = time.perf_counter()
    while True:
        for _ in range(16):
            step(dt, stable, finalize=False)
            dev.restore_state()
            steps += stable
            restores += 1
        sync()
        elapsed = time.perf_counter() - t0
        if elapsed >= seconds:
            break
    dev.set_option("asynchronous", 0)
    return {"seconds": elapsed, "steps": steps, "us_per_step": elapsed / steps * 1e6, "restores": restores, "restore_us": restore_us,
            "us_per_step_without_restores": (elapsed - restores * restore_us * 1e-6) / steps * 1e6, "steps_between_restores": stable,
            "status_flags": dev.status()}


def roofline_object(config, Nz, Nh, wordsize, kernel_s, kernel_name, pmc_name=None, workload=None):
    bytes_per_colstep = algorithmic_bytes_per_column_step(config, Nz, wordsize)
    achieved = bytes_per_colstep * Nh / kernel_s / 1e9
    r = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
         "traffic": None, "kernel": kernel_name, "kernel_ms": kernel_s * 1e3,
         "algorithmic_bytes_per_column_step": bytes_per_colstep, "columns_per_launch": Nh,
         "state_bytes_per_launch": state_bytes_per_step(config, Nz, Nh, wordsize)}
    if workload:
        r["workload"] = workload
    # HBM traffic of the dominant kernel from the PMC counters: collected in separate rocprofv3 --pmc passes of this
    # command by profiles/collect.sh (FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE), summary committed under
    # profiles/<round>/ together with the commit of the kernel it was measured on.  Never measured in this run.
    if pmc_name:
        pmc = os.path.join(ROOT, "profiles", PROFILE_ROUND, pmc_name)
        if os.path.exists(pmc):
            with open(pmc) as f:
                d = json.load(f)
            if d.get("columns") == Nh and d.get("levels") == Nz:
                r["traffic"] = d["hbm_traffic_bytes_per_launch"]
                r["traffic_source"] = (f"from the committed profile profiles/{PROFILE_ROUND}/{pmc_name} (bytes per launch, rocprofv3 --pmc "
                                       f"passes of this command; kernel source at commit {d.get('commit', '?')}), not from this run")
    return r


def _median(xs):
    return float(statistics.median(xs))


def measure(dev, w, config, steps, warmup, spinup_ms, heun, sync, barrier, repeats, reduce_max=None):
    """run_timed + the statistics of the line: per-repeat wall time (max over ranks), median / min / max."""
    walls, kernels, warm, chunked = run_timed(dev, w, config, steps, warmup, spinup_ms, heun, sync, barrier, repeats)
    if reduce_max is not None:
        walls, kernels = reduce_max(walls), reduce_max(kernels)
    K = max(steps, 1)
    return dict(wall_s=_median(walls), kernel_ms=_median(kernels), warm=warm, chunked=chunked, repeats=len(walls),
                ms_per_step=_median(walls) * 1e3 / K, ms_per_step_min=min(walls) * 1e3 / K, ms_per_step_max=max(walls) * 1e3 / K,
                kernel_us_per_step=_median(kernels) * 1e3 / K, kernel_us_per_step_min=min(kernels) * 1e3 / K)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--repeats", type=int, default=10, help="the K-step timed region is repeated this many times (state restored in between, untimed); the line reports the median")
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--kernel", default="fused", choices=["fused", "unfused"])
    ap.add_argument("--integrator", default="euler", choices=["euler", "heun"], help="ForwardEuler (headline) or Heun (one fused launch per step)")
    ap.add_argument("--series", action="store_true", help="drive the time-dependent boundary value / atmospheric inputs from device-resident time series (forcing feed) instead of constants")
    ap.add_argument("--spinup-ms", type=float, default=200.0, help="untimed device-busy time after the warm-up steps that lets the clocks settle (0: off)")
    ap.add_argument("--skip-kf", action="store_true", help="store hydraulic_conductivity only when finalizing")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hbm-resident", action="store_true", help="skip the HBM-resident companion measurements (c3x8, c5)")
    ap.add_argument("--no-single-process", action="store_true", help="skip the leg in which ONE host thread drives one context per device (trm_step_all)")
    ap.add_argument("--no-strong", action="store_true", help="N > 1: skip the strong-scaling companion (BASELINE config 4: N145 sharded over the ranks)")
    ap.add_argument("--multistep", type=int, default=50, help="also time the resident-column multi-step kernel with this many steps per launch (temporal blocking; reported separately)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target wall time of the CPU baseline sample")
    ap.add_argument("--pipeline", type=int, default=None, choices=[0, 1, 2], help="TRM_OPT_PIPELINE_PARTS (default: the library's rule)")
    ap.add_argument("--derive", type=int, default=None, choices=[0, 1, 2, 3], help="TRM_OPT_DERIVE_CLOSURE_FIELDS (default: the library's rule)")
    ap.add_argument("--steps-per-launch", type=int, default=1, help="TRM_OPT_STEPS_PER_LAUNCH of the measured context: 1 (default) = one launch per step, the state streams "
                    "through memory every step -- the per-step HBM roofline; 0 = the library's own choice (what a plain run! gets)")
    ap.add_argument("--sustained-seconds", type=float, default=3.0, help="length of the sustained leg: back-to-back per-step launches of the measured context, reported as "
                    "`sustained` beside the headline (0: off)")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.cpu_baseline_worker:
        return cpu_baseline_worker(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))     # before torch / the library are touched in this process
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (see the module docstring)")

    # The contract is ONE JSON line on stdout.  Libraries write there too (gloo announces its connections on fd 1): from here on
    # whatever anything prints to fd 1 goes to stderr, and the line is written to the descriptor saved here.
    sys.stdout.flush()
    line_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP library has no CPU fallback")
    # Rehearsal knobs (not used by the driver): TRM_BENCH_BACKEND=gloo + TRM_BENCH_SHARE_DEVICE=1 run the N > 1 code
    # path with several ranks on ONE GPU (RCCL refuses two ranks per device).
    backend = os.environ.get("TRM_BENCH_BACKEND", "nccl")
    if os.environ.get("TRM_BENCH_SHARE_DEVICE"):
        local_rank = local_rank % torch.cuda.device_count()
    if local_rank >= torch.cuda.device_count():
        sys.exit(f"bench.py: rank {rank} wants device {local_rank} but this node has {torch.cuda.device_count()} GPU(s)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    n_gpus = world
    stat_device = "cuda" if backend == "nccl" else "cpu"
    # a CPU-side group for waits that must not put a kernel on the waiting ranks' devices (an RCCL barrier spins ON the GPU): the
    # ranks that idle while rank 0 drives every device from one host thread (single_process_leg)
    cpu_group = None
    if world > 1 and backend == "nccl":
        try:
            cpu_group = dist.new_group(backend="gloo")
        except Exception as e:   # noqa: BLE001 -- (no gloo transport on this node: the default group's barrier serves)
            print(f"bench.py: no host-side group ({type(e).__name__}: {e}); idle ranks will wait in an RCCL barrier", file=sys.stderr)

    import workloads as W
    from terrarium_jl_amd import parallel

    def sync():
        torch.cuda.synchronize()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_max(values):        # per-repeat maximum over the ranks
        if world == 1:
            return list(values)
        t = torch.tensor(list(values), dtype=torch.float64, device=stat_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(x) for x in t.cpu()]

    def reduce_sum(x):
        if world == 1:
            return float(x)
        t = torch.tensor([float(x)], dtype=torch.float64, device=stat_device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t[0])

    w, desc, config, Nz, dt_name = build_workload(W, parallel, args.workload, world, rank, args.scaling)
    Nh = w["Nh"]
    wordsize = 8 if dt_name == "f64" else 4

    def configure(dev, steps_per_launch):
        dev.set_option("step_kernel", args.kernel)
        dev.set_option("steps_per_launch", steps_per_launch)
        # The roofline legs measure ONE column kernel per step over all columns of the context (what `roofline` prices and what the
        # rocprof row shows): the interleaved LandModel launches (TRM_OPT_PIPELINE_PARTS: half the columns + the surface processes
        # of the other half per launch; off by default) stay off unless asked for; `land_interleaved` reports them beside the C5 leg.
        dev.set_option("pipeline_parts", args.pipeline if args.pipeline is not None else 0)
        if args.derive is not None:
            dev.set_option("derive_closure_fields", args.derive)
        if args.skip_kf:
            dev.set_option("write_kf_every_step", 0)

    dev = W.setup_device(w, device=local_rank)
    configure(dev, args.steps_per_launch)
    if args.series:
        # the SURVEY 8(d) diurnal cycle sampled every 10 minutes over the run, linear in between
        DAY = 86400.0
        nodes = np.arange(0.0, (args.steps + args.warmup + 2) * w["dt"] + 600.0, 600.0)
        ph = 2 * np.pi * nodes[:, None] / DAY - w["lon"][None, :]
        if config in ("land", "landveg"):
            dev.set_forcing_series("air_temperature", nodes, w["T0"][None, :] + 5.0 * np.sin(ph))
            dev.set_forcing_series("surface_shortwave_down", nodes, np.maximum(0.0, 600.0 * np.sin(ph)))
        else:
            dev.set_bc_series("temperature", "top", "value", nodes, w["T0"][None, :] + 10.0 * np.sin(ph))

    heun = args.integrator == "heun"
    m = measure(dev, w, config, args.steps, args.warmup, args.spinup_ms, heun, sync, barrier, args.repeats, reduce_max)
    status = dev.status()
    total_columns = reduce_sum(Nh)
    nan_flag = parallel.global_status(status) if world > 1 else status
    kernel_s = m["kernel_us_per_step"] * 1e-6            # median duration of one step's launches on the slowest GPU
    value = total_columns * args.steps / m["wall_s"]
    packed = dt_name == "f32" and args.kernel == "fused" and not heun and args.steps_per_launch == 1      # (fp32: two columns per lane)
    program = dev.last_program()      # the instance the library selected for the measured launches (TRM_INFO_LAST_PROGRAM)
    kernel_name = {"column_land": "k_column_land", "packed_f32": "k_step_pk", "packed_land": "k_step_pk_land", "column_euler": "k_column", "column_heun": "k_column", "column_multi": "k_column",
                   "deep": "k_column_deep", "wide": "k_column_wide", "generic_euler": "k_step_wave", "generic_heun": "k_heun_generic"}.get(program["family"], "unfused sequence")
    assert packed == (program["family"] in ("packed_f32", "packed_land"))
    pmc_name = f"pmc_summary_{args.workload}_fused.json" if (args.kernel == "fused" and not heun and not args.series and args.steps_per_launch == 1
                                                            and not os.environ.get("TRM_BENCH_SHARD_OF")) else None      # (the committed counters are of the full size)

    out = {
        "metric": "column-steps/sec",
        "value": value,
        "unit": "column-steps/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": m["ms_per_step"],
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": dt_name,
        "data": "synthetic forcing and initial state on the reference's ERA5-land mask columns (SURVEY 8(d)); seeded",
        "config": {"workload": desc, "columns_per_gpu": Nh, "levels": Nz, "dt_s": w["dt"], "kernel": args.kernel, "integrator": args.integrator, "series": bool(args.series),
                   "steps_per_launch": args.steps_per_launch,
                   "statistic": f"median of {m['repeats']} timed regions of {args.steps} steps each (max over ranks per region); the state is restored between regions, untimed",
                   "repeats": m["repeats"], "ms_per_step_min": m["ms_per_step_min"], "ms_per_step_max": m["ms_per_step_max"],
                   "state_restored_every": STABLE_STEPS[config] if m["chunked"] else None,
                   "warmup_executed": m["warm"], "clock_spinup_ms": args.spinup_ms,
                   "parallelism": f"columns block-sharded over {n_gpus} GPU(s), no data-path collective",
                   "status_flags": int(nan_flag)},
        "roofline": roofline_object(config, Nz, Nh, wordsize, kernel_s, kernel_name, pmc_name),
    }
    out["roofline"]["kernel_ms_min"] = m["kernel_us_per_step_min"] * 1e-3
    out["roofline"]["program"] = program
    hung = False
    if world > 1:
        # global diagnostics through the library's own RCCL path (trm_comm_init / trm_status_global), beside torch's.  It runs
        # in a watchdog thread: should a collective of this path block, the line is still printed (with "timeout") and the
        # process leaves through os._exit with a NON-ZERO code -- a hung collective is a defect, not a pass.
        import threading
        box = {}
        th = threading.Thread(target=lambda: box.setdefault("r", abi_global_status(dev, dist, rank, world, local_rank, backend, status)), daemon=True)
        th.start()
        th.join(60.0)
        hung = th.is_alive()
        res = {"status": "timeout", "rccl_ranks": None} if hung else box.get("r", {"status": "failed", "rccl_ranks": None})
        out["config"]["abi_global_status"] = res["status"]
        out["config"]["rccl_ranks"] = res["rccl_ranks"]
        if hung:
            out["config"]["abi_global_status_note"] = ("trm_comm_init / trm_status_global did not return within 60 s; the ranks leave with exit code 3")
    if world > 1 and not hung and not args.no_strong and args.workload == "c3" and args.scaling == "weak":
        try:
            out["strong"] = strong_leg(W, parallel, args, world, rank, local_rank, configure, sync, barrier, reduce_max, reduce_sum)
        except Exception as e:   # noqa: BLE001 -- a companion leg (the same code on every rank: a failure is a failure on all of them); the headline line is still printed
            out["strong"] = {"error": f"{type(e).__name__}: {e}"}

    if not hung and not args.no_single_process and args.workload == "c3" and args.scaling == "weak" and args.kernel == "fused" and not heun:
        # ONE host thread driving one context per device (the reference's host is one Julia process): rank 0 alone, the other
        # ranks idle at the HOST-side barrier below
        if rank == 0:
            ndev = torch.cuda.device_count()      # (a rehearsal with several ranks on one GPU: the contexts share it)
            try:
                out["single_process"] = single_process_leg(W, parallel, args, [d % ndev for d in range(world)] if world > 1 else [local_rank, local_rank], torch)
            except Exception as e:   # noqa: BLE001 -- a companion leg: its failure is reported in the line, the ranks waiting below are released
                out["single_process"] = {"error": f"{type(e).__name__}: {e}"}
        if world > 1:
            torch.cuda.synchronize()
            dist.barrier(group=cpu_group)       # (gloo: the idle ranks wait on the host, their devices stay free for rank 0's contexts)

    single = rank == 0 and n_gpus == 1
    if single and args.multistep > 1 and args.kernel == "fused" and not heun and not args.series and args.steps_per_launch == 1:
        out["multistep"] = multistep_leg(W, w, desc, config, Nz, Nh, wordsize, args, sync, local_rank)
    if single and not args.no_hbm_resident and args.workload == "c3" and args.kernel == "fused" and not heun:
        out["roofline_hbm_resident"] = hbm_resident_leg(W, parallel, args, sync, local_rank)
    # The sustained leg comes LAST of the device work: three seconds of back-to-back launches leave the device in the state a long run
    # sees, in which an HBM-resident step reads ~8 % slower than after light work (profiles/r05/exp19, exp20) -- the companions above are
    # measured like the headline, from a device that has just run the warm-up and the clock spin-up.
    if not hung:
        if args.sustained_seconds > 0:
            out["sustained"] = sustained_leg(dev, w, config, args.sustained_seconds, heun, sync)
        dev.close()
    if single and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args)
    if rank == 0:
        print(json.dumps(out), file=line_out, flush=True)
    if hung or out.get("single_process", {}).get("global_diagnostics", {}).get("timeout"):
        os._exit(3)          # (a collective that did not return: the line says so, and no thread stuck in it keeps the process alive)
    if world > 1:
        dist.destroy_process_group()


def abi_global_status(dev, dist, rank, world, local_rank, backend, local_status):
    """ORs the status word over the ranks inside the library (RCCL, include/terrarium_hip.h: trm_comm_init) and checks it
    against torch.distributed's answer; reports the world size the library's communicator holds (trm_comm_info).  Failures are
    reported, never fatal: the timed numbers above do not depend on it."""
    try:
        if backend != "nccl" or not hasattr(dev, "comm_init"):
            return {"status": "skipped (needs the nccl backend: one rank per device)", "rccl_ranks": None}
        import torch
        uid = dev.comm_unique_id() if rank == 0 else bytes(128)
        t = torch.frombuffer(bytearray(uid), dtype=torch.uint8).cuda()
        dist.broadcast(t, src=0)
        dev.comm_init(rank, world, bytes(t.cpu().numpy().tobytes()))
        flags = dev.status_global()
        ref = torch.tensor([local_status], dtype=torch.int64, device="cuda")
        dist.all_reduce(ref, op=dist.ReduceOp.BOR)
        ok = int(ref[0]) == flags
        return {"status": "ok" if ok else f"mismatch: library {flags}, torch {int(ref[0])}", "rccl_ranks": dev.comm_world()}
    except Exception as e:   # noqa: BLE001 -- diagnostic leg
        return {"status": f"failed: {type(e).__name__}: {e}", "rccl_ranks": None}


def strong_leg(W, parallel, args, world, rank, device, configure, sync, barrier, reduce_max, reduce_sum):
    """BASELINE config 4 at N > 1: the N145 LandModel columns block-sharded over the ranks (7 119 per GPU at N = 8), total work
    fixed.  Per-step launches and the library's own choice for a plain run! (the resident program), same statistics as the
    headline.  Launch-latency-limited at this size."""
    w, desc, config, Nz, dt_name = build_workload(W, parallel, "c4", world, rank, "strong")
    total = reduce_sum(w["Nh"])
    res = {"workload": desc + f" -- block-sharded over {world} ranks", "columns_this_rank": w["Nh"], "columns_total": int(total), "scaling": "strong"}
    steps = min(args.steps, STABLE_STEPS[config])
    for key, spl in (("per_step", 1), ("library_default", 0)):
        dev = W.setup_device(w, device=device)
        configure(dev, spl)
        m = measure(dev, w, config, steps, min(args.warmup, 10), min(args.spinup_ms, 100.0), False, sync, barrier, args.repeats, reduce_max)
        st = dev.status()
        if os.environ.get("TRM_BENCH_DEBUG"):
            print(f"[rank {rank}] strong {key}: columns {w['Nh']} status {st} kernel us/step {m['kernel_us_per_step']:.2f} wall {m['ms_per_step'] * 1e3:.2f}", file=sys.stderr, flush=True)
        dev.close()
        res[key] = {"steps_per_launch": spl, "steps": steps, "us_per_step": m["ms_per_step"] * 1e3, "us_per_step_min": m["ms_per_step_min"] * 1e3,
                    "kernel_us_per_step": m["kernel_us_per_step"], "column_steps_per_s": total * steps / m["wall_s"], "repeats": m["repeats"],
                    "status_flags": int(parallel.global_status(st))}
    return res


def single_process_leg(W, parallel, args, devices, torch):
    """SURVEY 5 / 8(e) "1 process x 8 HIP devices": ONE host thread, one asynchronous context per entry of `devices` (each a full
    N145 C3 shard: weak scaling), stepped with trm_step_all -- the launches dealt to the devices in turn, one wait at the end --,
    per-step launches and the library's default (the resident program); global diagnostics through trm_status_global_all (grouped
    RCCL all-reduces when every context has a device of its own, a host fold otherwise).  On a one-GPU box the driver's default
    run exercises the same path with two contexts on the one device (they time-slice it: the numbers say nothing about scaling)."""
    import threading
    import terrarium_jl_amd as trm
    w, desc, config, Nz, dt_name = build_workload(W, parallel, "c3", 1, 0, "weak")
    n, distinct = len(devices), len(set(devices)) == len(devices)
    res = {"contexts": n, "devices": len(set(devices)), "columns_per_context": w["Nh"], "host_threads": 1,
           "note": "one host thread, trm_step_all: launches dealt to the contexts in turn, one wait at the end"
                   + ("" if distinct and n > 1 else "; the contexts share ONE device here (time-sliced): a functional run of the path, not a scaling number")}
    steps, reps = min(args.steps, 100), min(args.repeats, 5)

    def sync_all():
        for d in set(devices):
            torch.cuda.synchronize(d)

    for key, spl in (("per_step", 1), ("library_default", 0)):
        states = [W.setup_device(w, device=d) for d in devices]
        for s in states:
            s.set_option("steps_per_launch", spl)
            s.set_option("pipeline_parts", 0)
            s.set_option("asynchronous", 1)
        group = trm.DeviceGroup(states)
        group.step(w["dt"], 10, finalize=False)
        group.synchronize()
        for s in states:
            s.save_state()
        walls = []
        for _ in range(reps):
            for s in states:
                s.restore_state()
            group.synchronize()
            sync_all()
            t0 = time.perf_counter()
            group.step(w["dt"], steps, finalize=False)
            group.synchronize()
            walls.append(time.perf_counter() - t0)
        wall = float(np.median(walls))
        res[key] = {"steps_per_launch": spl, "steps": steps, "repeats": reps, "us_per_step": wall / steps * 1e6, "us_per_step_min": min(walls) / steps * 1e6,
                    "column_steps_per_s": n * w["Nh"] * steps / wall}
        if key == "library_default":
            # global diagnostics from the one thread; the RCCL set-up in a watchdog (a hung collective is reported, not waited for)
            box = {}

            def diagnostics():
                try:
                    if distinct and n > 1:
                        group.comm_init()
                    box["status"] = group.status_global()
                    box["rccl_ranks"] = states[0].comm_world()
                    box["max_T"] = float(np.max(group.reduce_global("temperature", "max")))
                except Exception as e:   # noqa: BLE001 -- diagnostic leg
                    box["error"] = f"{type(e).__name__}: {e}"

            th = threading.Thread(target=diagnostics, daemon=True)
            th.start()
            th.join(90.0)
            res["global_diagnostics"] = {"timeout": True} if th.is_alive() else dict(box, path="grouped RCCL all-reduce" if (distinct and n > 1 and "error" not in box) else "host fold (no communicator: one device)")
            if th.is_alive():
                return res          # (the contexts are left to the process exit)
        for s in states:
            s.close()
    return res


def hbm_resident_leg(W, parallel, args, sync, device):
    """The headline C3 state (6 x 14.6 MB) never leaves the 256 MiB Infinity Cache, so its roofline fraction is an
    algorithmic-bytes fraction of a cache-resident step.  Measured in the same run: the SAME kernel on 8 copies of the
    N145 columns (0.93 GB of state per launch) and the C5 shard (812 500 x 64 fp32, 3.3 GB) -- both far beyond the cache."""
    legs = []
    for name, steps in (("c3x8", 60), ("c5", 30)):
        w, desc, config, Nz, dt_name = build_workload(W, parallel, name, 1, 0, "weak")
        wordsize = 8 if dt_name == "f64" else 4
        dev = W.setup_device(w, device=device)
        dev.set_option("steps_per_launch", 1)
        dev.set_option("pipeline_parts", args.pipeline if args.pipeline is not None else 0)
        m = measure(dev, w, config, steps, 5, min(args.spinup_ms, 100.0), False, sync, None, min(args.repeats, 5))
        status = dev.status()
        interleaved = None
        if config == "land" and args.pipeline is None:     # the interleaved launches at this size, reported beside the roofline leg
            dev.set_option("pipeline_parts", 1)
            dev.restore_state()            # back to the warmed-up state the first measurement started from
            mi = measure(dev, w, config, steps, 0, 0.0, False, sync, None, min(args.repeats, 5))
            interleaved = {"us_per_step": mi["kernel_us_per_step"], "column_steps_per_s": w["Nh"] * steps / mi["wall_s"], "status_flags": int(dev.status()),
                           "note": "TRM_OPT_PIPELINE_PARTS = 1: k_land_pk, half the columns + the other half's surface processes per launch (off by default: no gain measured)"}
        dev.close()
        r = roofline_object(config, Nz, w["Nh"], wordsize, m["kernel_us_per_step"] * 1e-6, "k_step_pk (+ k_surface)" if name == "c5" else "k_column",
                            f"pmc_summary_{name}_fused.json", desc)
        r.update(steps=steps, repeats=m["repeats"], warmup_executed=m["warm"], column_steps_per_s=w["Nh"] * steps / m["wall_s"],
                 kernel_ms_min=m["kernel_us_per_step_min"] * 1e-3, status_flags=int(status))
        if interleaved:
            r["land_interleaved"] = interleaved
        legs.append(r)
        del w, dev
    first = legs[0]
    first["also"] = legs[1:]
    return first


def multistep_leg(W, w, desc, config, Nz, Nh, wordsize, args, sync, device):
    """Temporal blocking ("report it separately and never as the headline fraction"): columns stay in
    registers for `m` steps per launch, fields are written once per launch.  This is what a plain run! gets by default
    (TRM_OPT_STEPS_PER_LAUNCH = 0)."""
    mm = args.multistep
    dev = W.setup_device(w, device=device)
    dev.set_option("steps_per_launch", mm)
    steps = max(mm, (min(args.steps, STABLE_STEPS[config]) // mm) * mm)
    m = measure(dev, w, config, steps, args.warmup, args.spinup_ms, False, sync, None, min(args.repeats, 5))
    status = dev.status()
    dev.close()
    return {"steps_per_launch": mm, "steps": steps, "repeats": m["repeats"], "column_steps_per_s": Nh * steps / m["wall_s"], "us_per_step": m["kernel_us_per_step"],
            "status_flags": int(status),
            "note": "resident-column multi-step kernel: bit-identical to per-step launches; NOT comparable with the per-step HBM roofline"}


def cpu_baseline(args):
    """The CPU leg runs in a child process of its own (started after all GPU work of this process is done): the OpenMP
    runtime reads OMP_PROC_BIND / OMP_PLACES when it is loaded, and torch has long loaded one here."""
    env = dict(os.environ, OMP_PROC_BIND="spread", OMP_PLACES="cores", OMP_WAIT_POLICY="passive")
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "--workload", args.workload, "--cpu-seconds", str(args.cpu_seconds)]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=max(120.0, 8 * args.cpu_seconds))
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not line:
            return {"value": None, "unit": "column-steps/s", "cores": 0, "kind": "port", "sample": f"CPU leg failed (exit {r.returncode}): {r.stderr[-300:]}"}
        return json.loads(line[-1])
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "column-steps/s", "cores": 0, "kind": "port", "sample": "CPU leg timed out"}


def cpu_share():
    """CPUs this process may actually use: the cgroup quota (a GPU box hands a 1-GPU job 16 of its 256 hardware threads' worth of
    CPU time) bounded by the affinity mask.  Threads beyond the quota only fight over it."""
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:              # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = f.read().split()
            quota = None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            quota = q / per if q > 0 else None
        except (OSError, ValueError):
            pass
    return avail, quota


def julia_probe():
    """first choice for the CPU leg is the reference's own `timestep!` if `julia` and a depot with Terrarium.jl
    exist on the box.  Probed, never assumed; nothing can be installed."""
    exe = shutil.which("julia")
    if not exe:
        return {"julia": None, "terrarium_jl": None, "note": "no `julia` on PATH of this box (shutil.which)"}
    try:
        r = subprocess.run([exe, "--startup-file=no", "-e", "using Terrarium; print(pkgversion(Terrarium))"], capture_output=True, text=True, timeout=180)
        ok = r.returncode == 0
        return {"julia": exe, "terrarium_jl": r.stdout.strip() if ok else None,
                "note": "Terrarium.jl importable: the reference leg could be timed with it" if ok else f"`using Terrarium` failed: {r.stderr.strip()[-200:]}"}
    except Exception as e:   # noqa: BLE001
        return {"julia": exe, "terrarium_jl": None, "note": f"probe failed: {e}"}


def cpu_baseline_worker(args):
    """CPU restatement of the reference path (oracle/, reference kernel order) timed on this box's host cores on the same
    workload: whole column set, bounded number of steps.  Headline `value`: one pass per reference kernel with OpenMP over
    columns, thread count auto-tuned over a few candidates, threads bound to cores and spread over the NUMA nodes
    (OMP_PROC_BIND=spread OMP_PLACES=cores, set by the parent), every array first-touched by the thread that owns its columns
    (oracle/terrarium_oracle.hpp: FieldVec / first_touch).  BASELINE.md 4.2's other legs ride along in `legs`: the same driver
    on one thread, and the cache-blocked ("fused") driver -- every pass over one block of columns at a time -- on one thread
    and on the tuned thread count."""
    import oracle
    import workloads as W
    from terrarium_jl_amd import parallel
    w, desc, config, Nz, dt_name = build_workload(W, parallel, args.workload, 1, 0, "weak")
    target_seconds = args.cpu_seconds
    avail, quota = cpu_share()
    limit = min(avail, 128 if quota is None else max(8, int(2 * quota + 0.5)))
    candidates = sorted({c for c in (4, 8, 16, 32, 64, 128) if c <= limit} | {min(limit, 8)})
    dt, Nh = w["dt"], w["Nh"]
    stable = min(STABLE_STEPS[config], 150)

    def timed(orc, fn, steps):
        t0 = time.perf_counter()
        fn(orc, steps)
        return time.perf_counter() - t0

    ref_order = lambda o, n: o.steps(dt, n)
    blocked = lambda o, n: o.steps_blocked(dt, n, 64)
    budget = target_seconds
    scan = {}
    best = None
    for threads in candidates:
        oracle.set_threads(threads)
        orc = W.setup_oracle(w, omp=True)        # (first touch with THIS thread count)
        ref_order(orc, 2)  # spin up the thread pool
        sec = timed(orc, ref_order, 10)
        budget -= sec
        rate = Nh * 10 / sec
        scan[threads] = rate
        if best is None or rate > best[1]:
            best = (threads, rate)
        if budget < target_seconds * 0.5 or rate < 0.8 * best[1]:      # (past the knee: more threads only fight over the cores)
            break
    threads = best[0]
    legs = {}
    # single thread: a few steps are enough (seconds each at N145 size)
    for label, fn, thr, n in (("reference_order_1_thread", ref_order, 1, 3), ("cache_blocked_1_thread", blocked, 1, 3),
                              (f"cache_blocked_{threads}_threads", blocked, threads, 20)):
        oracle.set_threads(thr)
        orc = W.setup_oracle(w, omp=True)
        fn(orc, 1)
        sec = timed(orc, fn, n)
        legs[label] = {"value": Nh * n / sec, "steps": n, "seconds": round(sec, 2)}
        budget -= sec
    oracle.set_threads(threads)
    chunk, done, spent = min(50, stable), 0, 0.0
    while spent < max(budget, 2.0) and done < 5000:
        if done % stable == 0:      # a fresh state every stable stretch (the explicit Richards scheme dries cells out later)
            orc = W.setup_oracle(w, omp=True)
            ref_order(orc, 2)
        spent += timed(orc, ref_order, chunk)
        done += chunk
    value = Nh * done / spent
    legs[f"reference_order_{threads}_threads"] = {"value": value, "steps": done, "seconds": round(spent, 2)}
    bpc = REFERENCE_ORDER_BYTES_PER_CELL_STEP[config]
    probe = julia_probe()
    out = {"value": value, "unit": "column-steps/s", "cores": threads, "kind": "port",
           "sample": f"all {Nh} columns x {done} steps of the same workload; CPU restatement of the Terrarium.jl path (not Terrarium.jl itself: "
                     f"{probe['note']}), one pass per reference kernel, OpenMP over columns with {threads} threads ({avail} hardware threads in the "
                     f"affinity mask, cgroup CPU quota {'none' if quota is None else round(quota, 1)}) "
                     f"(best of a scan, column-steps/s: " + ", ".join(f"{k}: {v:.3g}" for k, v in scan.items()) + f"), threads bound "
                     f"(OMP_PROC_BIND={os.environ.get('OMP_PROC_BIND')}, OMP_PLACES={os.environ.get('OMP_PLACES')}), arrays first-touched by the owning thread; "
                     f"{spent:.1f} s; other legs (column-steps/s): " + ", ".join(f"{k} {v['value']:.3g}" for k, v in legs.items()),
           "streamed_GBps": value * Nz * bpc / 1e9,
           "streamed_bytes_per_cell_step": bpc,
           "thread_scan": {str(k): v for k, v in scan.items()},
           "hardware_threads": avail, "cgroup_cpu_quota": quota,
           "julia_probe": probe,
           "legs": legs}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
