#!/bin/bash
out=gpurun_out/r2; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "packed or packing" > $out/pytest_packed.log 2>&1; tail -25 $out/pytest_packed.log
for wl in c3 c2; do for pk in -1 0; do
  timeout -k 10 120 python bench.py --workload $wl --packing $pk --no-cpu-baseline --steps 500 --warmup 50 > $out/pk.json 2>$out/pk.err || { tail -3 $out/pk.err; continue; }
  python3 -c "
import json;d=json.loads(open('$out/pk.json').read().strip().splitlines()[-1]);r=d['roofline'];g=d['config']['geometry']
print('$wl packing=$pk', g, ': %.1f us/solve  rollout %.1f us frac %.3f'%(d['ms_per_step']*1e3,r['kernel_ms']*1e3,r['frac']), 'blocking', d['latency']['blocking_get_act_ms'])" | tee -a $out/pk.txt
done; done
