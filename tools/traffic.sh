#!/bin/bash
# usage: tools/traffic.sh <tag> <bench args...>
# HBM traffic of the rollout kernel from the TCC counters, one counter per pass
# (guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
# reports half the bytes of a wide coalesced read -> doubled here; WRITE_SIZE is exact for
# 16-B-per-lane streaming stores).  Summary -> gpurun_out/prof/<tag>_traffic.json
tag=$1; shift
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/prof/${tag}_$ctr -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-events "$@" > gpurun_out/prof/${tag}_$ctr.log 2>&1 || { tail -3 gpurun_out/prof/${tag}_$ctr.log; exit 1; }
done
python3 - "$tag" "$@" <<'PY'
import csv,glob,sys,json,collections
tag=sys.argv[1]
out={}
for ctr in ("FETCH_SIZE","WRITE_SIZE"):
    agg=collections.defaultdict(list)
    for f in glob.glob(f'gpurun_out/prof/{tag}_{ctr}/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name']==ctr: agg[r['Kernel_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        if 'mppi' in k: out.setdefault(k,{})[ctr+"_KiB_mean"]=sum(v)/len(v)
res={}
for k,d in out.items():
    f=d.get("FETCH_SIZE_KiB_mean",0)*1024; w=d.get("WRITE_SIZE_KiB_mean",0)*1024
    res[k]={"fetch_bytes_raw":f,"fetch_bytes_corrected_x2":2*f,"write_bytes":w,"hbm_bytes":2*f+w}
json.dump({"args":sys.argv[2:],"kernels":res},open(f'gpurun_out/prof/{tag}_traffic.json','w'),indent=1)
print(json.dumps(res,indent=1))
PY
rm -rf gpurun_out/prof/${tag}_FETCH_SIZE gpurun_out/prof/${tag}_WRITE_SIZE
