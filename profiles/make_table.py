#!/usr/bin/env python3
"""Markdown table of DESIGN.md section 5 from the committed profile of a round:  python profiles/make_table.py r04
Every number is read from profiles/<round>/: bench_<wl>.json (the bench line of `bench.py --workload <wl> --no-hbm-resident
--multistep 0 --no-cpu-baseline`), kernel_stats_<wl>.csv (rocprofv3 --kernel-trace --stats of the same command: one launch size
per file) and pmc_summary_<wl>_fused.json (rocprofv3 --pmc passes)."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
d = os.path.join(ROOT, rnd)
NAMES = {"c3": "**C3** N145 × 32, heat + Richards, fp64 — headline", "c3x8": "C3 physics on 8 × N145 (HBM-resident, 0.93 GB of state)",
         "c5": "C5 812 500 × 64 LandModel fp32 (packed)", "c4": "C4 N145 LandModel, default hydraulics", "c4vg": "C4 LandModel, van Genuchten",
         "c5vg": "C5 LandModel fp32, van Genuchten", "c2": "C2 N72 × 30 heat-only fp64", "c3vg": "C3 van Genuchten + Mualem K",
         "c4vgveg": "C4-VG coupled to vegetation", "c4_shard8": "C4 as one rank of eight holds it (7 119 columns)"}
print("| workload | steps per region | column-steps/s (wall, median of 10) | µs per step: wall median [min, max] | kernel µs per step (HIP events) | "
      "rocprof CSV: kernel, calls, average µs | `roofline.frac` | PMC traffic ÷ algorithmic | VALU / SALU / SMEM / branches per wave |")
print("|---|---|---|---|---|---|---|---|---|")
for wl, label in NAMES.items():
    if not os.path.exists(os.path.join(d, f"bench_{wl}.json")):
        continue
    b = json.load(open(os.path.join(d, f"bench_{wl}.json")))
    pmc = os.path.join(d, f"pmc_summary_{wl}_fused.json")
    p = json.load(open(pmc)) if os.path.exists(pmc) else None
    rows = list(csv.DictReader(open(os.path.join(d, f"kernel_stats_{wl}.csv"))))
    rows = [r for r in rows if "trm::" in r["Name"]]
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    ks = "; ".join(f"`{r['Name'].split('(')[0].replace('void trm::', '')}` × {r['Calls']}: {float(r['AverageNs']) / 1e3:.2f}" for r in rows[:2])
    c, r, pw = b["config"], b["roofline"], (p["per_wave"] if p else {})
    traffic = f"{p['traffic_over_algorithmic']:.2f}× ({p['hbm_traffic_bytes_per_launch'] / 1e6:.1f} MB)" if p else "—"
    mix = f"{pw.get('valu', 0):.0f} / {pw.get('salu', 0):.0f} / {pw.get('smem', 0):.0f} / {pw.get('branch', 0):.0f}" if p else "—"
    print(f"| {label} | {b['steps']} | {b['value'] / 1e9:.3f} G | {b['ms_per_step'] * 1e3:.2f} [{c['ms_per_step_min'] * 1e3:.2f}, {c['ms_per_step_max'] * 1e3:.2f}] | "
          f"{r['kernel_ms'] * 1e3:.2f} | {ks} | **{r['frac']:.3f}** | {traffic} | {mix} |")
