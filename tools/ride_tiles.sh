#!/bin/bash
# usage: tools/ride_tiles.sh <bench args>: deferred mode with riding forced on (100 tiles) / default (2) / off (0)
mkdir -p gpurun_out
for mt in 100 2 0; do
  export MPPI_RIDE_MAX_TILES=$mt
  for r in 1 2; do
  timeout -k 10 120 python bench.py --no-cpu-baseline --pipeline 0 "$@" > gpurun_out/rt_$mt.json 2>gpurun_out/rt_$mt.err || { tail -3 gpurun_out/rt_$mt.err; exit 1; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/rt_$mt.json').read().strip().splitlines()[-1]);r=d['roofline'];print('ride_max_tiles $mt:',round(d['ms_per_step']*1e3,2),'us  rollout',r['kernel_ms'],'combine',r['combine_kernel_ms'])"
  done
done
