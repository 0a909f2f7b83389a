#!/bin/bash
# usage: tools_sweep.sh "c2 c3" "0 8 16 32 64"   -> one summary line per (workload, chunks)
for w in $1; do for c in $2; do
  timeout -k 10 120 python bench.py --workload $w --chunks $c --steps 200 --warmup 20 --no-cpu-baseline $3 > gpurun_out/b.json 2>gpurun_out/b.err && python -c "
import json;d=json.load(open('gpurun_out/b.json'));r=d['roofline'];g=d['config']['geometry']
print('$w chunks=%d nq=%d grid=%d'%(g['chunks'],g['nq'],g['grid']),'us/solve=%.1f'%(d['ms_per_step']*1e3),'rollout_us=%.1f combine_us=%.1f'%(r['kernel_ms']*1e3,r['combine_kernel_ms']*1e3),'val=%.3g'%d['value'],'frac=%.3f'%r['frac'])" || tail -1 gpurun_out/b.err; done; done
