#!/bin/bash
# usage: tools/ab_env.sh <rounds> <ENVVAR> "<values>" <bench args...>   -- alternate one environment
# variable of the engine over its values on one box (same library), bench.py's per-solve time each
rounds=$1; var=$2; vals=$3; shift 3
for r in $(seq 1 $rounds); do for v in $vals; do
  env $var=$v timeout -k 10 200 python bench.py --no-cpu-baseline --no-pmc --no-extra --no-latency "$@" 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);r=d['roofline'];print('$var=$v round $r: %.2f us/solve  kernel %.2f us'%(d['ms_per_step']*1e3,r['kernel_ms']*1e3))"
done; done
