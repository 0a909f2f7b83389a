run() { python bench.py "$@" --no-cpu-baseline --no-pmc --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  ms_per_step',round(d['ms_per_step']*1e3,2),'us kernel',round(d['roofline']['kernel_ms']*1e3,2),'combine',round(d['roofline']['combine_kernel_ms']*1e3,2), d['config']['launches_in_timed_region'], 'lat', d.get('latency',{}).get('blocking_get_act_ms'))"; }
for i in 1 2; do
echo "c2 default"; run --workload c2 --steps 3000
echo "c2 packing 8"; run --workload c2 --steps 3000 --packing 8
echo "c2 packing 5"; run --workload c2 --steps 3000 --packing 5
done
echo "c4"; run --workload c4 --steps 1000
echo "c4full"; run --workload c4full --steps 200 --warmup 20
