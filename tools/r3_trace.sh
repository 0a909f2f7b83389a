#!/bin/bash
# region time stamps (trace build) -> gpurun_out/r3/trace_regions.txt
out=gpurun_out/r3; mkdir -p $out
export MPPI_GPU_AMD_LIB=$PWD/mppi_gpu_amd/lib/trace/libmppi_gpu_amd.so
run() { echo "-- TRACE_RIDE=$TRACE_RIDE MPPI_TRACE_TILE=$MPPI_TRACE_TILE $*"; timeout -k 10 120 python tools/trace_regions.py "$@" 2>&1 | grep -v amdgpu.ids; }
{
TRACE_RIDE=1 run 2 10000 200            # C2, row-aligned, riding
TRACE_RIDE=1 run 2 10000 200 0 8        # C2, packed (8 groups per lane), riding
TRACE_RIDE=1 MPPI_TRACE_TILE=0 run 3 100000 200     # C3 riding, first tile
TRACE_RIDE=1 MPPI_TRACE_TILE=5 run 3 100000 200     # C3 riding, sixth tile
} > $out/trace_regions.txt 2>&1
cat $out/trace_regions.txt
