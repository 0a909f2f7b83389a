#!/bin/bash
# usage: tools/abn.sh <rounds> "<variant names (base = product library)>" <bench args...>
rounds=$1; names=$2; shift 2
for r in $(seq 1 $rounds); do for v in $names; do
  if [ $v = base ]; then unset MPPI_GPU_AMD_LIB; else export MPPI_GPU_AMD_LIB=$PWD/mppi_gpu_amd/lib/alt_$v/libmppi_gpu_amd.so; fi
  timeout -k 10 120 python bench.py --no-cpu-baseline --no-pmc --no-extra "$@" 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);r=d['roofline'];l=d.get('latency') or {};print('$v $r: %.2f us/solve  rollout %.2f us  blocking median %.2f us'%(d['ms_per_step']*1e3,r['kernel_ms']*1e3,(l.get('blocking_get_act_ms_quantiles') or {}).get('median',0)*1e3))"
done; done
