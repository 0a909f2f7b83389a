run() { python bench.py "$@" --no-cpu-baseline --no-pmc --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  us/solve',round(d['ms_per_step']*1e3,2),'kernel',round(d['roofline']['kernel_ms']*1e3,2), 'packed' if d['config']['geometry']['packed'] else 'row chunks=%d'%d['config']['geometry']['chunks'], 'blocking', d['latency']['blocking_get_act_ms'])"; }
for w in s3 s1 s2; do for i in 1 2; do
echo "$w default"; run --workload $w --steps 3000
echo "$w packed"; run --workload $w --steps 3000 --packing $( [ $w = s2 ] && echo 5 || echo 4 )
done; done
