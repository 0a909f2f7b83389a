#!/bin/bash
# usage: tools/ab.sh <tag> <rounds> <bench args...>: alternate the product library and the
# variant in mppi_gpu_amd/lib/alt/ (same session, same box), print us/solve and kernel ms of each
tag=$1; rounds=$2; shift 2
mkdir -p gpurun_out
for r in $(seq 1 $rounds); do
  for v in base alt; do
    if [ $v = alt ]; then export MPPI_GPU_AMD_LIB=$PWD/mppi_gpu_amd/lib/alt/libmppi_gpu_amd.so; else unset MPPI_GPU_AMD_LIB; fi
    timeout -k 10 120 python bench.py --no-cpu-baseline "$@" > gpurun_out/ab_${tag}_${v}_$r.json 2>/dev/null || exit 1
    python3 -c "
import json;d=json.loads(open('gpurun_out/ab_${tag}_${v}_$r.json').read().strip().splitlines()[-1]);print('$tag $v $r',round(d['ms_per_step']*1e3,2),'us  rollout',d['roofline']['kernel_ms'],'combine',d['roofline']['combine_kernel_ms'])"
  done
done
