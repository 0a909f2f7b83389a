#!/usr/bin/env python3
"""Static instruction mix of one kernel in a hipcc -S listing: tools_isa_mix.py file.s substring"""
import re, sys, collections
lines = open(sys.argv[1]).read().splitlines()
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith('_ZN') and key in l and l.rstrip().split(':')[0][-1] in 'Ey' )
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
ops = collections.Counter()
for l in lines[start + 1:end]:
    l = l.strip()
    m = re.match(r'([a-z_0-9]+)(\s|$)', l)
    if m and not l.startswith('.') and not l.startswith(';') and not l.endswith(':'):
        ops[m.group(1)] += 1
cats = collections.Counter()
for k, v in ops.items():
    if k.startswith(('v_mad_u64', 'v_mul_hi', 'v_mul_lo')): cats['intmul'] += v
    elif k in ('v_exp_f32', 'v_log_f32', 'v_sin_f32', 'v_cos_f32', 'v_sqrt_f32', 'v_rcp_f32', 'v_rsq_f32'): cats['trans'] += v
    elif k.startswith(('ds_bpermute', 'ds_swizzle', 'v_permlane')) or 'dpp' in k: cats['xlane'] += v
    elif k.startswith('ds_'): cats['lds'] += v
    elif k.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): cats['vmem'] += v
    elif k.startswith(('v_readlane', 'v_writelane', 'v_readfirstlane')): cats['lane<->sgpr'] += v
    elif k.startswith('s_'): cats['salu'] += v
    elif k.startswith('v_'): cats['valu'] += v
    else: cats['other'] += v
print(lines[start].split(':')[0], 'instructions:', sum(ops.values()))
print('  ', dict(cats.most_common()))
print('  top:', ops.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 30))
