#!/bin/bash
# usage: tools_kt.sh <tag> <bench args...>  -> rocprofv3 kernel-trace stats into gpurun_out/prof/<tag>_stats.csv
# KT_STEPS solves (default 4000): --stats averages EVERY dispatch of the process, and the first
# hundred run at rising clocks (tools/kt_gaps.sh: C3 83 us against 65 at steady state); the run must
# be long enough for them not to weigh
tag=$1; shift
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/${tag}_kt -- python3 bench.py --steps ${KT_STEPS:-4000} --warmup 20 --no-cpu-baseline --no-events "$@" > gpurun_out/prof/${tag}_kt.log 2>&1 || { tail -3 gpurun_out/prof/${tag}_kt.log; exit 1; }
cp gpurun_out/prof/${tag}_kt/*/*kernel_stats.csv gpurun_out/prof/${tag}_stats.csv
rm -rf gpurun_out/prof/${tag}_kt     # raw traces: gpurun merges at most 64 MiB back
head -4 gpurun_out/prof/${tag}_stats.csv
grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof/${tag}_kt.log
