#!/bin/bash
# usage: tools/modes.sh <tag> <bench args...>: us/solve in the three enqueue modes + blocking get_act
tag=$1; shift
mkdir -p gpurun_out
for cfg in "0" "1" "0 --blocking" "1 --blocking"; do
  name=$(echo $cfg | tr -d ' -')
  timeout -k 10 120 python bench.py --no-cpu-baseline --pipeline $cfg "$@" > gpurun_out/modes_${tag}_$name.json 2>gpurun_out/modes_${tag}_$name.err || { tail -3 gpurun_out/modes_${tag}_$name.err; exit 1; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/modes_${tag}_$name.json').read().strip().splitlines()[-1]);r=d['roofline'];print('$tag mode $cfg:',round(d['ms_per_step']*1e3,2),'us  rollout',r['kernel_ms'],'combine',r['combine_kernel_ms'],'frac',r['frac'])"
done
