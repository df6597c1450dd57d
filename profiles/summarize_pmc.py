#!/usr/bin/env python3
"""Turns the rocprofv3 --pmc csv output of profiles/collect.sh into the per-launch summary of the fused
step kernel (profiles/<round>/pmc_summary_c3_fused.json).

HBM traffic per MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are reported in KiB... on gfx950 FETCH_SIZE counts
half of the bytes actually fetched, so reads = FETCH_SIZE * 1024 * 2; writes = WRITE_SIZE * 1024.
SQ_*_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* are in quad-cycles (x4 for cycles), summed over all SIMDs."""
import csv
import glob
import json
import os
import sys

KERNEL = "k_step_wave"
COLUMNS, LEVELS, WORD = 56951, 32, 8


def collect(directory):
    sums, counts = {}, {}
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if KERNEL not in row.get("Kernel_Name", ""):
                    continue
                name = row["Counter_Name"]
                key = (name, row["Dispatch_Id"])
                sums[key] = sums.get(key, 0.0) + float(row["Counter_Value"])
    per_counter = {}
    for (name, _), v in sums.items():
        per_counter.setdefault(name, []).append(v)
    return {n: sum(v) / len(v) for n, v in per_counter.items()}, {n: len(v) for n, v in per_counter.items()}


def main(out):
    counters, ndisp = {}, {}
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq1", "pmc_sq2"):
        c, n = collect(os.path.join(out, sub))
        counters.update(c)
        ndisp.update(n)
    cells = COLUMNS * LEVELS
    read = counters.get("FETCH_SIZE", 0.0) * 1024 * 2
    write = counters.get("WRITE_SIZE", 0.0) * 1024
    waves = counters.get("SQ_WAVES", 0.0) or 1.0
    summary = {
        "kernel": "trm::k_step_wave<double, true, 0, 32, false>",
        "workload": "C3 N145 x 32, heat + Richards, fp64",
        "columns": COLUMNS, "levels": LEVELS,
        "hbm_read_bytes_per_launch_corrected": read,
        "hbm_write_bytes_per_launch": write,
        "hbm_traffic_bytes_per_launch": read + write,
        "traffic_bytes_per_cell": (read + write) / cells,
        "algorithmic_bytes_per_launch": COLUMNS * WORD * (8 * LEVELS + 4),
        "algorithmic_bytes_per_cell": WORD * (8 * LEVELS + 4) / LEVELS,
        "note": "the kernel reads U, sat, T, liq, psi (5 words) and writes U, sat, T, liq, psi, K (6 words) per cell = 88 B; "
                "the SURVEY's algorithmic figure (65 B) counts only U, sat reads and the 6 writes",
        "per_wave": {k[9:].lower(): counters[k] / waves for k in counters if k.startswith("SQ_INSTS_")},
        "counters": counters,
        "dispatches_averaged": ndisp,
    }
    json.dump(summary, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
