#!/bin/bash
# usage: tools_pmc.sh <tag> "<bench args>"  -- two SQ counter passes, summary to gpurun_out/prof/<tag>_pmc.txt
tag=$1; shift
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/prof/${tag}_p$i -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-events "$@" > gpurun_out/prof/${tag}_p$i.log 2>&1 || { tail -3 gpurun_out/prof/${tag}_p$i.log; exit 1; }
done
python3 - "$tag" <<'PY'
import csv,collections,glob,sys
tag=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f'gpurun_out/prof/{tag}_p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
with open(f'gpurun_out/prof/{tag}_pmc.txt','w') as out:
    for k,d in agg.items():
        if 'mppi' not in k: continue
        w=sum(d['SQ_WAVES'])/len(d['SQ_WAVES']) if 'SQ_WAVES' in d else 1
        line=k[:60]+' waves=%d'%w+' | per-wave: '+' '.join('%s=%.0f'%(c.replace('SQ_',''),sum(v)/len(v)/w) for c,v in sorted(d.items()) if c!='SQ_WAVES')
        print(line); out.write(line+'\n')
PY
rm -rf gpurun_out/prof/${tag}_p1 gpurun_out/prof/${tag}_p2
