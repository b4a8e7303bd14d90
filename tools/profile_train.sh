#!/bin/bash
# Run ON THE GPU BOX via gpurun: rocprofv3 kernel statistics of the training step (tools/train_step.py).
#   gpurun --timeout 600 -- 'bash tools/profile_train.sh <tag>'     ->  gpurun_out/<tag>_train_kernel_stats.csv
TAG=${1:-r}; O=$PWD/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
D=$O/prof_train_$TAG; rm -rf $D
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o tr -- python3 tools/train_step.py --steps 60 --warmup 15 > $O/${TAG}_train_prof.log 2>&1 || { tail -5 $O/${TAG}_train_prof.log; exit 1; }
F=$(find $D -name "*kernel_stats.csv" | head -1)
cp "$F" $O/${TAG}_train_kernel_stats.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/${TAG}_train_kernel_stats.csv")))
steps=75
for r in rows[:14]:
    n=r['Name'].replace('void ','').replace('stgcn::','').replace('(anonymous namespace)::','').split('(')[0]
    print(f"{n[:44]:44s} x{int(r['Calls'])/steps:4.1f} {float(r['TotalDurationNs'])/steps/1e3:8.1f} us/step")
print("total", round(sum(float(r['TotalDurationNs']) for r in rows)/steps/1e3,1))
PY
