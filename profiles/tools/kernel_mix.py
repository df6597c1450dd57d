"""Instruction mix of kernels in a device assembly listing:  hipcc ... -S --cuda-device-only -o /tmp/trm.s terrarium_hip.hip
   python profiles/tools/kernel_mix.py /tmp/trm.s k_surface_vegId"""
import collections
import re
import sys

path, pattern = sys.argv[1], sys.argv[2]
name, body = None, collections.defaultdict(list)
for line in open(path):
    m = re.match(r'^(_ZN\S+):', line)
    if m:
        name = m.group(1)
    elif name and line.startswith('\t') and not line.startswith('\t.') and not line.startswith('\t;'):
        body[name].append(line.split()[0])
for k, v in body.items():
    if pattern in k:
        c = collections.Counter(v)
        print(k[:48], len(v), 'readlane', c['v_readlane_b32'], 'writelane', c['v_writelane_b32'], 'scratch', sum(c[x] for x in c if x.startswith('scratch')),
              'div_scale', c['v_div_scale_f64'], 'rcp', c['v_rcp_f64_e32'], 's_load', sum(c[x] for x in c if x.startswith('s_load')))
