#!/bin/bash
# usage: tools/csweep.sh <workload> "<chunks list>" <bench args...>
wl=$1; cl=$2; shift 2
mkdir -p gpurun_out
for c in $cl; do
  timeout -k 10 120 python bench.py --no-cpu-baseline --workload $wl --chunks $c "$@" > gpurun_out/cs_${wl}_$c.json 2>gpurun_out/cs_${wl}_$c.err || { echo "chunks $c: failed"; tail -2 gpurun_out/cs_${wl}_$c.err; continue; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/cs_${wl}_$c.json').read().strip().splitlines()[-1]);r=d['roofline'];print('$wl chunks $c:',round(d['ms_per_step']*1e3,2),'us  rollout',r['kernel_ms'],'geo',d['config']['geometry'])"
done
