#!/bin/bash
# what rocprofv3 --kernel-trace does to the launch cadence: dispatch durations and the idle gaps
# between consecutive dispatches of one profiled bench run.  usage: tools/kt_gaps.sh <bench args>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof; rm -rf gpurun_out/prof/ktgaps
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof/ktgaps -- python3 bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-events --no-pmc --no-extra --no-latency "$@" > gpurun_out/prof/ktgaps.log 2>&1
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob('gpurun_out/prof/ktgaps/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        if 'mppi' in r['Kernel_Name'] and 'rollout' in r['Kernel_Name']:
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
rows.sort()
n = len(rows)
dur = [(e - s) * 1e-3 for s, e in rows]
gap = [(rows[i + 1][0] - rows[i][1]) * 1e-3 for i in range(n - 1)]
import statistics as st
for name, sl in (('first 100', slice(0, 100)), ('last 200', slice(n - 200, n))):
    print(name, 'dispatches: duration median %.2f us mean %.2f, gap to the next median %.2f us mean %.2f' % (
        st.median(dur[sl]), st.mean(dur[sl]), st.median(gap[sl.start:sl.stop - 1 if sl.stop else None]), st.mean(gap[sl.start:sl.stop - 1 if sl.stop else None])))
PY
grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof/ktgaps.log
mkdir -p gpurun_out/prof; rm -rf gpurun_out/prof/ktgaps
