#!/bin/bash
# usage: tools/alu.sh <tag> <bench args...>
# VALU side of the roofline: instruction counts by class and VALU-busy time of every mppi kernel
# (two SQ counter passes and one GRBM pass, --kernel-trace only), summary -> gpurun_out/prof/<tag>_alu.json:
# per kernel the mean per-dispatch counter totals and the mean dispatch duration of the same runs.
tag=$1; shift
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_CVT" \
           "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/prof/${tag}_alu$i -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-events "$@" > gpurun_out/prof/${tag}_alu$i.log 2>&1 || { tail -3 gpurun_out/prof/${tag}_alu$i.log; exit 1; }
done
python3 - "$tag" "$@" <<'PY'
import csv, collections, glob, json, sys
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(f'gpurun_out/prof/{tag}_alu*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob(f'gpurun_out/prof/{tag}_alu*/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        dur[r['Kernel_Name']].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3)
out = {}
for k, d in agg.items():
    if 'mppi' not in k:
        continue
    ent = {c: sum(v) / len(v) for c, v in sorted(d.items())}
    if dur.get(k):
        ent['dispatch_us_mean_in_these_passes'] = sum(dur[k]) / len(dur[k])
        ent['dispatches'] = len(dur[k])
    out[k] = ent
json.dump({"args": sys.argv[2:], "kernels": out}, open(f'gpurun_out/prof/{tag}_alu.json', 'w'), indent=1)
for k, e in out.items():
    print(k[:70], {c: round(v, 1) for c, v in e.items()})
PY
rm -rf gpurun_out/prof/${tag}_alu1 gpurun_out/prof/${tag}_alu2 gpurun_out/prof/${tag}_alu3
