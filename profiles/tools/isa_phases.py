"""Per-PHASE instruction budget of a step kernel.

The kernel is one inlined function whose phases the optimiser interleaves (it even sinks the whole closure into the store
block), so the budget is taken from a MARKER build: -DTRM_PHASE_MARKERS makes every TRM_PHASE_FENCE(name, live values...) of
trm_column.hpp leave a `; TRM_PHASE name` line in the device assembly and makes the live values opaque at that point, so no
arithmetic crosses the boundary; the instructions between two marker lines are the phase's.  Regions between `rare+ <what>` and
`rare-` markers and loop bodies are the RARE path (lane-serial saturation repair, the status report); the rest is the COMMON
path, which still holds wave-uniform branches a configuration skips (boundary kinds that are not set, the phase-change divide,
finalize) -- the PMC count per wave (profiles/r04/pmc_summary_*.json) is the dynamic total to hold against it.  The marker build's
total differs from the shipped kernel's by what the optimiser shares ACROSS phases; `--shipped other.s` prints the shipped
kernel's totals beside it.
    make -C terrarium.jl_amd/csrc asm F=trm_launch_column_f64_euler_rich EXTRA=-DTRM_PHASE_MARKERS OBJDIR=../../build/markers
    python profiles/tools/isa_phases.py build/markers/trm_launch_column_f64_euler_rich.s _ZN3trm8k_columnIdLb1ELi0ELi32ELi1ELi0ELb0ELb0ELb0ELb1E \\
        [--shipped build/obj/trm_launch_column_f64_euler_rich.s] [--json out.json]"""
import collections
import json
import re
import sys

args = sys.argv[1:]


def opt(name, default=None):
    if name in args:
        i = args.index(name)
        v = args[i + 1]
        del args[i:i + 2]
        return v
    return default


out_json, shipped = opt("--json"), opt("--shipped")
path, pattern = args[0], args[1]


def cls(op):
    if op.startswith("v_cndmask"): return "select"
    if op.startswith("v_cmp"): return "compare"
    if op.startswith("v_mov_b32_dpp") or op.endswith("_dpp"): return "dpp"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): return "lane"
    if re.match(r"v_(pk_)?(mul|add|fma|fmac|max|min|rcp|rsq|sqrt|div_fixup|div_scale|div_fmas|exp|log|ldexp|frexp|trunc|floor|rndne|cvt)_", op): return "fp"
    if op.startswith("v_"): return "mov/int"
    if op.startswith(("s_load", "s_buffer_load")): return "SMEM"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_call")): return "branch"
    if op.startswith(("s_waitcnt", "s_nop", "s_endpgm")): return "wait"
    if op.startswith("s_"): return "SALU"
    if op.startswith(("global_load", "buffer_load", "flat_load")): return "VMEM ld"
    if op.startswith(("global_store", "buffer_store", "flat_store", "global_atomic")): return "VMEM st"
    if op.startswith("ds_"): return "LDS"
    return "other"


VALU = ("fp", "select", "compare", "dpp", "mov/int", "lane")
cols = ("VALU", "fp", "select", "compare", "dpp", "mov/int", "lane", "SALU", "SMEM", "branch", "VMEM ld", "VMEM st", "LDS")


def scan(path):
    name, phase, rare, in_loop = None, "prologue", None, False
    counts = collections.OrderedDict()
    for line in open(path):
        m = re.match(r"^(_Z\S+):", line)
        if m:
            if name:
                break
            if pattern in m.group(1):
                name = m.group(1)
            continue
        if name is None:
            continue
        if ".Lfunc_end" in line:
            break
        s = line.strip()
        m = re.match(r"; TRM_PHASE (.*)", s)
        if m:
            tag = m.group(1).strip()
            if tag.startswith("rare+"): rare = tag[5:].strip() or "rare"
            elif tag.startswith("rare-"): rare = None
            else: phase = tag
            continue
        if re.match(r"^\.LBB\d+_\d+:", line) or re.match(r"^; %bb\.\d+:", line):
            in_loop = "in Loop" in line or "Inner Loop" in line
            continue
        if line.startswith("\t") and not s.startswith((".", ";")):
            key = (phase, "rare" if (rare or in_loop) else "common")
            counts.setdefault(key, collections.Counter())[cls(s.split()[0])] += 1
    return name, counts


name, counts = scan(path)
phases = list(collections.OrderedDict.fromkeys(p for p, _ in counts))
print(name.split("EEvNS")[0][:110])
print(f"  {'phase':24s} {'path':7s}" + "".join(f"{c:>8s}" for c in cols))
table, total = [], {"common": collections.Counter(), "rare": collections.Counter()}
for ph in phases:
    for kind in ("common", "rare"):
        c = counts.get((ph, kind))
        if not c:
            continue
        c["VALU"] = sum(c[k] for k in VALU)
        total[kind].update(c)
        table.append(dict(phase=ph, path=kind, **{k: c[k] for k in cols}))
        print(f"  {ph:24s} {kind:7s}" + "".join(f"{c[k]:8d}" for k in cols))
for kind in ("common", "rare"):
    print(f"  {'TOTAL (marker build)':24s} {kind:7s}" + "".join(f"{total[kind][k]:8d}" for k in cols))
ship = None
if shipped:
    _, sc = scan(shipped)
    ship = collections.Counter()
    for c in sc.values():
        c["VALU"] = sum(c[k] for k in VALU)
        ship.update(c)
    print(f"  {'TOTAL (shipped kernel)':24s} {'all':7s}" + "".join(f"{ship[k]:8d}" for k in cols))
if out_json:
    json.dump(dict(kernel=name, phases=table, total={k: {c: v[c] for c in cols} for k, v in total.items()},
                   shipped_total={c: ship[c] for c in cols} if ship else None), open(out_json, "w"), indent=1)
