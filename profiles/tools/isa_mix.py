"""Opcode-class mix of one kernel in a device assembly listing (static count over the whole kernel body, cold paths included):
   make -C terrarium.jl_amd/csrc asm F=trm_launch_column_f64_euler_rich   (-> build/obj/<F>.s; any trm_launch_*.hip)
   python profiles/tools/isa_mix.py /tmp/trm.s _ZN3trm8k_columnIdLb1ELi0ELi32ELi1ELi0ELb0ELb0E"""
import collections
import re
import sys

path, pattern = sys.argv[1], sys.argv[2]
name, body, meta = None, collections.defaultdict(list), collections.defaultdict(dict)
for line in open(path):
    m = re.match(r'^(_Z\S+):', line)
    if m:
        name = m.group(1)
        continue
    if name is None:
        continue
    m = re.match(r'^\s*; (NumSgprs|NumVgprs|Occupancy|ScratchSize|SGPRBlocks|codeLenInByte)\s*[:=]?\s*(\d+)', line)
    if m:
        meta[name][m.group(1)] = int(m.group(2))
    if line.startswith('\t') and not line.startswith('\t.') and not line.startswith('\t;'):
        body[name].append(line.split()[0])


def cls(op):
    if op.startswith('v_mov_b32_dpp') or op.endswith('_dpp'): return 'VALU dpp'
    if op.startswith('v_cndmask'): return 'VALU select'
    if op.startswith(('v_rcp_f64', 'v_rsq_f64', 'v_sqrt_f64', 'v_div_fixup_f64', 'v_div_scale_f64', 'v_div_fmas_f64', 'v_exp', 'v_log', 'v_rcp_f32', 'v_sqrt_f32')): return 'VALU transcendental / divide'
    if op.startswith('v_cmp'): return 'VALU compare'
    if re.match(r'v_(pk_)?(mul|add|fma|fmac|max|min|ldexp|frexp|trunc|floor|rndne|cvt)_', op) and ('f64' in op or 'f32' in op): return 'VALU fp arithmetic'
    if op.startswith(('v_readlane', 'v_writelane', 'v_readfirstlane')): return 'VALU lane<->scalar'
    if op.startswith('v_'): return 'VALU other (mov / int / addr)'
    if op.startswith('s_load') or op.startswith('s_buffer_load'): return 'SMEM'
    if op.startswith(('s_cbranch', 's_branch', 's_setpc', 's_call')): return 'branch'
    if op.startswith('s_waitcnt'): return 's_waitcnt'
    if op.startswith('s_nop'): return 's_nop'
    if op.startswith('s_'): return 'SALU'
    if op.startswith(('global_load', 'buffer_load', 'flat_load')): return 'VMEM load'
    if op.startswith(('global_store', 'buffer_store', 'flat_store')): return 'VMEM store'
    if op.startswith('global_atomic'): return 'VMEM atomic'
    if op.startswith('ds_'): return 'LDS / bpermute'
    if op.startswith('scratch'): return 'scratch'
    return 'other'


for k, v in body.items():
    if pattern in k:
        c = collections.Counter(cls(op) for op in v)
        print(k.split('EEvNS')[0][:100])
        print('  instructions', len(v), ' '.join(f'{a}={b}' for a, b in sorted(meta[k].items())))
        for a, b in c.most_common():
            print(f'  {a:32s} {b:5d}')
        top = collections.Counter(v).most_common(12)
        print('  top opcodes: ' + ', '.join(f'{a} {b}' for a, b in top))
