#!/usr/bin/env python3
"""Static resource usage (VGPRs, SGPRs, occupancy, scratch, static VALU count) of the step kernels from the device ISA:
    make -C terrarium.jl_amd/csrc asm F=trm_launch_column_f64_euler_rich   (-> build/obj/<F>.s; any trm_launch_*.hip)
    python profiles/tools/kernel_resources.py /tmp/trm.s [filter ...]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
filters = sys.argv[2:] or ["k_column<double, true, 0, 32", "k_step_wave<double, true, 0, 32", "k_step_pk"]
names = re.findall(r"^(_ZN3trm\w+):", txt, re.M)
dem = dict(zip(names, subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")))
for m in re.finditer(r"^(_ZN3trm\w+):.*?; NumVgprs: (\d+).*?; Occupancy: (\d+)", txt, re.S | re.M):
    d = dem.get(m.group(1), m.group(1))
    if not any(f in d for f in filters):
        continue
    body = txt[m.start():m.end()]
    valu = len(re.findall(r"^\s+v_", body, re.M))
    sgpr = re.search(r"; TotalNumSgprs: (\d+)", body).group(1)
    scr = re.search(r"; ScratchSize: (\d+)", body).group(1)
    print(f"{d.split('(')[0][:100]:100s} vgpr {m.group(2):>3s} sgpr {sgpr:>3s} occ {m.group(3)} scratch {scr} static-valu {valu}")
