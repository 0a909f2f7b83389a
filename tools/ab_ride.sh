for i in 1 2; do
for v in 0 1; do
  echo "MPPI_RIDE_LONG=$v"
  MPPI_RIDE_LONG=$v python bench.py --workload c3 --steps 1500 --warmup 200 --no-cpu-baseline --no-pmc --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  ms_per_step',round(d['ms_per_step']*1e3,2),'us kernel',d['roofline']['kernel_ms']*1e3,'combine',d['roofline']['combine_kernel_ms']*1e3, d['config']['launches_in_timed_region'])"
done; done
