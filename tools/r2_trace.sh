#!/bin/bash
out=gpurun_out/r2; mkdir -p $out
export MPPI_GPU_AMD_LIB=$PWD/mppi_gpu_amd/lib/trace/libmppi_gpu_amd.so
for cfg in "$@"; do echo "-- $cfg"; timeout -k 10 120 python tools/trace_regions.py $cfg 2>&1 | grep -v amdgpu.ids | tee -a $out/trace_regions.txt | tail -14; done
