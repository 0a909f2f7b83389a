export MPPI_GPU_AMD_LIB=$PWD/mppi_gpu_amd/lib/trace/libmppi_gpu_amd.so
TRACE_RIDE=1 timeout -k 10 120 python tools/trace_regions.py 2 10000 200 0 8 2>&1 | grep -v amdgpu.ids
TRACE_RIDE=1 MPPI_TRACE_TILE=0 timeout -k 10 120 python tools/trace_regions.py 3 100000 200 2>&1 | grep -v amdgpu.ids | head -30
