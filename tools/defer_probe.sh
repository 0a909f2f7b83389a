#!/bin/bash
mkdir -p gpurun_out
for sp in 1 2 4 8; do  # MPPI_COMBINE_SPLITS probe of the riding combine
  export MPPI_COMBINE_SPLITS=$sp
  for r in 1 2; do
  timeout -k 10 120 python bench.py --no-cpu-baseline --pipeline 0 "$@" > gpurun_out/dp_$sp.json 2>gpurun_out/dp_$sp.err || { tail -3 gpurun_out/dp_$sp.err; exit 1; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/dp_$sp.json').read().strip().splitlines()[-1]);r=d['roofline'];print('splits $sp:',round(d['ms_per_step']*1e3,2),'us  rollout',r['kernel_ms'])"
  done
done
