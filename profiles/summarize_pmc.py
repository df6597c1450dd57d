#!/usr/bin/env python3
"""Turns the rocprofv3 --pmc csv output of profiles/collect.sh into the per-launch summary of the dominant step kernel
(profiles/<round>/pmc_summary_<workload>_fused.json):  summarize_pmc.py <dir> <workload> <commit>

HBM traffic per MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts half of the
bytes of a coalesced streaming read, so reads = FETCH_SIZE * 1024 * 2; writes = WRITE_SIZE * 1024.  Infinity-Cache hits
are counted (the counters sit on the L2's fabric side).  SQ_*_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* are in quad-cycles, summed
over all SIMDs."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (workload table and the algorithmic-bytes formula)

SIZES = {"c3": 56951, "c3x8": 455608, "c5": 812500, "c5vg": 812500, "c4": 56951, "c4vg": 56951, "c3vg": 56951, "c4vgveg": 56951, "c2": 14017}


def collect(directory, kernel):
    sums = {}
    names = set()
    rows = []
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            rows += [row for row in csv.DictReader(f) if kernel in row.get("Kernel_Name", "")]
    # one template instance only: the one launched most often (the first launch of a context runs the non-deriving instance)
    freq = {}
    for row in rows:
        freq[row["Kernel_Name"]] = freq.get(row["Kernel_Name"], 0) + 1
    dominant = max(freq, key=freq.get) if freq else None
    for row in rows:
        if row["Kernel_Name"] != dominant:
            continue
        names.add(row["Kernel_Name"].split("(")[0])
        key = (row["Counter_Name"], row["Dispatch_Id"])
        sums[key] = sums.get(key, 0.0) + float(row["Counter_Value"])
    per_counter = {}
    for (name, _), v in sums.items():
        per_counter.setdefault(name, []).append(v)
    return {n: sum(v) / len(v) for n, v in per_counter.items()}, {n: len(v) for n, v in per_counter.items()}, sorted(names)


def main(out, wl, commit):
    desc, config, hydraulics, columns, Nz, dt_name, replicas = bench.WORKLOADS[wl]
    kernel = "k_step_pk" if dt_name == "f32" else "k_column"
    Nh = SIZES[wl]
    word = 8 if dt_name == "f64" else 4
    counters, ndisp, names = {}, {}, []
    for sub in ("fetch", "write", "sq1", "sq2"):
        c, n, nm = collect(os.path.join(out, sub), kernel)
        counters.update(c)
        ndisp.update(n)
        names = nm or names
    cells = Nh * Nz
    read = counters.get("FETCH_SIZE", 0.0) * 1024 * 2
    write = counters.get("WRITE_SIZE", 0.0) * 1024
    waves = counters.get("SQ_WAVES", 0.0) or 1.0
    alg = bench.algorithmic_bytes_per_column_step(config, Nz, word)
    summary = {
        "kernel": names, "workload": desc, "commit": commit, "columns": Nh, "levels": Nz,
        "hbm_read_bytes_per_launch_corrected": read,
        "hbm_write_bytes_per_launch": write,
        "hbm_traffic_bytes_per_launch": read + write,
        "traffic_bytes_per_cell": (read + write) / cells,
        "algorithmic_bytes_per_launch": Nh * alg,
        "algorithmic_bytes_per_cell": alg / Nz,
        "traffic_over_algorithmic": (read + write) / (Nh * alg),
        "per_wave": {k[9:].lower(): counters[k] / waves for k in counters if k.startswith("SQ_INSTS_")},
        "counters": counters,
        "dispatches_averaged": ndisp,
    }
    json.dump(summary, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "unknown")
