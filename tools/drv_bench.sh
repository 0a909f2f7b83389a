#!/bin/bash
# the driver's own bench invocation (BENCH_rNN.json: --steps 20 --warmup 5), N times on one box
n=${1:-2}
for i in $(seq 1 $n); do
  t0=$(date +%s.%N)
  timeout -k 10 280 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_drv$i.log 2> gpurun_out/bench_drv$i.err || exit 1
  t1=$(date +%s.%N)
  python3 - "$i" "$t0" "$t1" <<'PY'
import json, sys
i, t0, t1 = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
d = json.loads(open("gpurun_out/bench_drv%s.log" % i).read().strip().splitlines()[-1])
r = d["roofline"]
print("run %s: %.1f s  value %.4g  ms/step %.5f  frac %.3f  kernel_ms %.5f  rocprof %s  traffic %s" % (
    i, t1 - t0, d["value"], d["ms_per_step"], r["frac"], r["kernel_ms"], r.get("kernel_ms_rocprof"), r.get("traffic")))
print("   latency", d["latency"]["blocking_get_act_ms"], d["latency"].get("closed_loop_with_plant_step"))
c = d["extra"]["c3"]
print("   c3", c["ms_per_step"], c["roofline"]["frac"], c["roofline"]["solve_frac"])
PY
done
