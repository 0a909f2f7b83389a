#!/bin/bash
# round-3 profile set, one box: kernel-trace stats, SQ counters, HBM traffic and VALU counters of
# the bench workloads (C2 default = riding launch, C2 eager = plain rollout + combine, C3 default =
# riding packed launch, C3 eager, C4 shard), plus the bench lines themselves.
# Everything lands in gpurun_out/prof/ and gpurun_out/r3/; tools/mkprofiles.py 3 turns it into profiles/.
set -o pipefail
mkdir -p gpurun_out/prof gpurun_out/r3
for cfg in "c2 --workload c2" "c2e --workload c2 --pipeline 1" "c3 --workload c3" "c3e --workload c3 --pipeline 1" "c4 --workload c4"; do
  set -- $cfg; tag=r3_$1
  case $1 in c2|c2e) export KT_STEPS=20000;; *) export KT_STEPS=4000;; esac
  shift
  echo "== $tag kt"; bash tools/kt.sh $tag --no-pmc --no-extra --no-latency "$@" | tail -5
done
for cfg in "c2 --workload c2" "c2e --workload c2 --pipeline 1" "c3 --workload c3" "c3e --workload c3 --pipeline 1"; do
  set -- $cfg; tag=r3_$1; shift
  echo "== $tag traffic"; bash tools/traffic.sh $tag --no-pmc --no-extra --no-latency "$@" | tail -3
  echo "== $tag alu"; bash tools/alu.sh $tag --no-pmc --no-extra --no-latency "$@" | tail -3
  echo "== $tag pmc"; bash tools/pmc.sh $tag --no-pmc --no-extra --no-latency "$@" | tail -3
done
echo "== bench lines"
python bench.py > gpurun_out/r3/bench_c2.json 2> gpurun_out/r3/bench_c2.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3/bench_c2_driverlike.json 2> gpurun_out/r3/bench_c2_driverlike.err
python bench.py --workload c3 > gpurun_out/r3/bench_c3.json 2> gpurun_out/r3/bench_c3.err
python bench.py --workload c4 --no-cpu-baseline > gpurun_out/r3/bench_c4shard.json 2>/dev/null
python bench.py --workload c4full --no-cpu-baseline --steps 300 --warmup 30 > gpurun_out/r3/bench_c4full.json 2>/dev/null
python bench.py --workload c3x2 --no-cpu-baseline --steps 1000 > gpurun_out/r3/bench_c3x2.json 2>/dev/null
{ for w in c2 c3 c4 c3x2 c3x4 c4full; do for m in 1 2 0; do MPPI_STORE_MODE=$m python bench.py --workload $w --steps $( [ $w = c4full ] && echo 200 || echo 1000 ) --warmup 20 --no-cpu-baseline --no-pmc --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(\"$w MPPI_STORE_MODE=$m (1 write-through, 2 non-temporal, 0 the engine's choice by footprint): %.2f us per solve, kernel %.2f us\" % (d[\"ms_per_step\"]*1e3, d[\"roofline\"][\"kernel_ms\"]*1e3))"; done; done; } > gpurun_out/r3/store_mode.txt 2>&1
python bench.py --workload c1 --no-cpu-baseline --no-pmc > gpurun_out/r3/bench_c1.json 2>/dev/null
python bench.py --force-sharded --no-cpu-baseline --no-pmc > gpurun_out/r3/bench_c2_sharded_1rank.json 2>gpurun_out/r3/bench_fs.err
{ for cfg in "2 10000 200 2000" "3 100000 200 500" "1 100 50 2000" "3 3000 50 2000"; do timeout -k 10 120 tools/latency_probe $cfg; done
  # the closed loop with a plant step between two calls, the next solve's noise drawn ahead or not
  for th in 5 20 100; do for pf in 0 1; do for cfg in "2 10000 200 2000" "3 3000 50 2000" "3 100000 200 500"; do
    echo "MPPI_PREFETCH=$pf plant step $th us:"; MPPI_PREFETCH=$pf timeout -k 10 120 tools/latency_probe $cfg $th; done; done; done; } > gpurun_out/r3/latency_probe.txt
# config 5: the closed loop (stand-in plant, blocking get_act + set_x per control step), 1 GPU and --gpus all
g++ -O2 -std=c++17 -I include apps/mppi_closed_loop.cpp -o apps/mppi_closed_loop -L mppi_gpu_amd/lib -lmppi_gpu_amd_sharded -lmppi_gpu_amd -Wl,-rpath,$PWD/mppi_gpu_amd/lib -Wl,-rpath,/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib -pthread 2>/dev/null
{ apps/mppi_closed_loop --dims 3 --samples 100000 --horizon 200 --seconds 2 | grep -E "RESULT|controller";
  apps/mppi_closed_loop --dims 3 --samples 100000 --horizon 200 --seconds 2 --gpus all --transport collective | grep -E "RESULT|controller:";
  apps/mppi_closed_loop --dims 2 --samples 10000 --horizon 200 --seconds 2 | grep -E "RESULT"; } > gpurun_out/r3/closed_loop.txt 2>&1
bash tools/closed_loop_rates.sh > gpurun_out/r3/closed_loop_rates.txt 2>&1
python tools/sweep_cost_error.py > gpurun_out/r3/sweep_cost_error.txt 2>/dev/null
python tools/lambda_speed.py > gpurun_out/r3/lambda_speed.txt 2>/dev/null
{ timeout -k 10 300 python tools/soak_packed.py 5000; timeout -k 10 300 python tools/soak.py 50000 4; } 2>&1 | grep -v amdgpu.ids > gpurun_out/r3/soak.txt
ls gpurun_out/r3 gpurun_out/prof | head -60
