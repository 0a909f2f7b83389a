#!/bin/bash
# occupancy experiment: the same C3 work on persistent grids of 1, 2, 3 ... blocks per CU
out=gpurun_out/r2; mkdir -p $out
for ch in 16 32; do for mb in 256 512 768 1024 1536 0; do
  timeout -k 10 120 python bench.py --workload c3 --chunks $ch --max-blocks $mb --no-cpu-baseline --steps 300 --warmup 50 > $out/occ.json 2>$out/occ.err || { tail -3 $out/occ.err; continue; }
  python3 -c "
import json;d=json.loads(open('$out/occ.json').read().strip().splitlines()[-1]);r=d['roofline'];g=d['config']['geometry']
print('c3 chunks=$ch max_blocks=$mb grid=%d: %.1f us/solve  rollout %.1f us'%(g['grid'],d['ms_per_step']*1e3,r['kernel_ms']*1e3))" | tee -a $out/occ.txt
done; done
for mb in 256 512 625 0; do
  timeout -k 10 120 python bench.py --workload c2 --max-blocks $mb --no-cpu-baseline --steps 1000 --warmup 100 > $out/occ.json 2>$out/occ.err || { tail -3 $out/occ.err; continue; }
  python3 -c "
import json;d=json.loads(open('$out/occ.json').read().strip().splitlines()[-1]);r=d['roofline'];g=d['config']['geometry']
print('c2 max_blocks=$mb grid=%d: %.1f us/solve  rollout %.1f us'%(g['grid'],d['ms_per_step']*1e3,r['kernel_ms']*1e3))" | tee -a $out/occ.txt
done
