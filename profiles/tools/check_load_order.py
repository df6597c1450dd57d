"""Guard for the one property of the step kernels that the compiler's scheduler loses most easily (round 4): every
field / level-record / per-column VECTOR load of a wave's prologue is issued before the wave waits for any of them.  An inline
asm statement, a reused destination register or a branch in the wrong place puts an `s_waitcnt vmcnt` between two loads, and the
later ones start their trip to memory one full latency late -- invisible in the tests, 2 ... 5 % in time.
    for f in trm_launch_column_sig_f64_rich_a trm_launch_column_sig_f64_rich_b trm_launch_column_sig_f64_noflow trm_launch_column_f64_euler_rich trm_launch_packed; do make -C terrarium.jl_amd/csrc asm F=$f; done
    python profiles/tools/check_load_order.py            (exit code 1 if a listed kernel waits between its prologue loads)"""
import re
import sys

CHECK = [("build/obj/trm_launch_column_sig_f64_rich_a.s", "_ZN3trm8k_columnIdLb1ELi0ELi32ELi1ELi0ELb0ELb0ELb0ELb1ELi2E", 3),   # C3: fields only (inputs by s_load)
         ("build/obj/trm_launch_column_sig_f64_rich_a.s", "_ZN3trm8k_columnIdLb1ELi0ELi32ELi1ELi0ELb0ELb0ELb1ELb0ELi2E", 5),   # 8 x N145: fields + 2 per-column
         ("build/obj/trm_launch_column_sig_f64_rich_b.s", "_ZN3trm8k_columnIdLb1ELi0ELi32ELi1ELi0ELb0ELb0ELb1ELb1ELi64E", 3),  # C4
         ("build/obj/trm_launch_column_sig_f64_rich_b.s", "_ZN3trm8k_columnIdLb1ELi1ELi32ELi0ELi0ELb0ELb0ELb0ELb1ELi64E", 5),  # vegetation-coupled: T, liq read
         ("build/obj/trm_launch_column_sig_f64_noflow.s", "_ZN3trm8k_columnIdLb0ELi0ELi32ELi0ELi0ELb0ELb0ELb0ELb1ELi2E", 4),   # C2: heat-only, small grid
         ("build/obj/trm_launch_column_f64_euler_rich.s", "_ZN3trm8k_columnIdLb1ELi1ELi32ELi0ELi0ELb0ELb0ELb0ELb1ELin1E", 5),  # the run-time kinds
         ("build/obj/trm_launch_packed.s", "_ZN3trm9k_step_pkILb1ELi64ELi0ELi2ELi64E", 8)]                                     # C5: 4 fields x 2 columns
bad = 0
for path, kernel, nfield in CHECK:
    try:
        lines = open(path).read().split("\n")
    except OSError:
        print(f"{path}: missing (make asm F=...)"); bad = 1; continue
    start = next(i for i, l in enumerate(lines) if l.startswith(kernel))
    loads, waits = [], []
    for n, l in enumerate(lines[start:start + 400]):
        if re.search(r"\bglobal_load_dword(x2)?\b", l) and "offset" not in l: loads.append(n)      # (the level records are the offset: loads)
        if "s_waitcnt" in l and "vmcnt" in l: waits.append(n)
        if "s_cbranch" in l: break
    first, last = loads[0], loads[nfield - 1]
    between = [w for w in waits if first < w < last]
    print(f"{kernel[:60]:60s} loads at {loads[:nfield]}  first vmcnt wait at {waits[0] if waits else None}  {'WAIT BETWEEN LOADS' if between else 'ok'}")
    bad |= bool(between)
sys.exit(bad)
