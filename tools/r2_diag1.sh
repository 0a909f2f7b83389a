#!/bin/bash
# round-2 diagnostics, one box: issue-rate microbenchmark, host latency probe, baseline benches,
# region stamps.  Everything goes to gpurun_out/r2/.
set -o pipefail
out=gpurun_out/r2; mkdir -p $out
echo "== ubench_issue"; timeout -k 10 200 tools/ubench_issue > $out/ubench_issue.txt 2>&1; tail -30 $out/ubench_issue.txt
echo "== latency probe"; for cfg in "2 10000 200 2000" "3 100000 200 500" "1 100 50 2000"; do timeout -k 10 120 tools/latency_probe $cfg | tee -a $out/latency_probe.txt; done
echo "== bench c2"; timeout -k 10 300 python bench.py > $out/bench_c2.json 2> $out/bench_c2.err || tail -5 $out/bench_c2.err; cat $out/bench_c2.json
echo "== bench c3"; timeout -k 10 300 python bench.py --workload c3 --no-cpu-baseline > $out/bench_c3.json 2> $out/bench_c3.err || tail -5 $out/bench_c3.err; cat $out/bench_c3.json
echo "== trace regions"
export MPPI_GPU_AMD_LIB=$PWD/mppi_gpu_amd/lib/trace/libmppi_gpu_amd.so
for cfg in "3 100000 200 16" "3 100000 200 32" "2 10000 200 16"; do echo "-- $cfg"; timeout -k 10 120 python tools/trace_regions.py $cfg 2>&1 | tee -a $out/trace_regions.txt | tail -14; done
unset MPPI_GPU_AMD_LIB
