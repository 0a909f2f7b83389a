#!/bin/bash
# what the box's GPU clock, power and temperature are while the C3 workload runs back to back (boxes
# differ by ~5 % in the VALU-bound launches: this says what they run at)
python bench.py --workload c3 --steps 150000 --warmup 200 --no-cpu-baseline --no-pmc --no-extra --no-latency > gpurun_out/clk_bench.json 2>/dev/null &
bp=$!
while kill -0 $bp 2>/dev/null; do
  /opt/rocm/bin/rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|Power|Sensor junction" | sed 's/^GPU\[0\][ \t]*: //' | tr '\n' ';'
  echo
  sleep 1
done | sort | uniq -c | sort -rn | head -12
python3 -c "
import json;d=json.loads(open('gpurun_out/clk_bench.json').read().strip().splitlines()[-1]);print('c3: %.2f us per solve, kernel %.2f us'%(d['ms_per_step']*1e3,d['roofline']['kernel_ms']*1e3))"
