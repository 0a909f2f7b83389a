#!/bin/bash
# usage: tools/ride_probe.sh <bench args>: deferred mode with the rollout-side split merge on (RS <= 8) / off (0)
mkdir -p gpurun_out
for mm in 8 0; do
  export MPPI_RIDE_MERGE_MAX=$mm
  for r in 1 2; do
  timeout -k 10 120 python bench.py --no-cpu-baseline --pipeline 0 "$@" > gpurun_out/rp_$mm.json 2>gpurun_out/rp_$mm.err || { tail -3 gpurun_out/rp_$mm.err; exit 1; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/rp_$mm.json').read().strip().splitlines()[-1]);r=d['roofline'];print('merge_max $mm:',round(d['ms_per_step']*1e3,2),'us  rollout',r['kernel_ms'])"
  done
done
unset MPPI_RIDE_MERGE_MAX
timeout -k 10 120 python bench.py --no-cpu-baseline --pipeline 2 "$@" > gpurun_out/rp_e.json 2>/dev/null
python3 -c "
import json;d=json.loads(open('gpurun_out/rp_e.json').read().strip().splitlines()[-1]);r=d['roofline'];print('eager:',round(d['ms_per_step']*1e3,2),'us  rollout',r['kernel_ms'],'combine',r['combine_kernel_ms'])"
