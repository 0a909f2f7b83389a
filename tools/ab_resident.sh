for w in c4 c3x2 c4full c3; do for r in 0 96 160 208; do
MPPI_STORE_MODE=2 MPPI_NT_RESIDENT_MB=$r python bench.py --workload $w --steps $( [ $w = c4full ] && echo 200 || echo 1000 ) --warmup 20 --no-cpu-baseline --no-pmc --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(\"$w nt, first $r MB write-through: %.2f us per solve\" % (d[\"ms_per_step\"]*1e3))"; done
MPPI_STORE_MODE=1 python bench.py --workload $w --steps $( [ $w = c4full ] && echo 200 || echo 1000 ) --warmup 20 --no-cpu-baseline --no-pmc --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(\"$w all write-through: %.2f us per solve\" % (d[\"ms_per_step\"]*1e3))"; done
