#!/bin/bash
# Run ON THE GPU BOX via gpurun: per-kernel times of the training step under the shipped library and a variant
#   gpurun -- 'bash tools/gpu_ab_kernels.sh <variant>'
R=$PWD; V=$1
cd /tmp && export TMPDIR=/tmp
for w in A B; do
  if [ $w = B ]; then export STGCN_LIB=$R/st-gcn-altformer_amd/stgcn_amd/libstgcn_hip_$V.so; else unset STGCN_LIB; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab_$w -o run -- python3 $R/tools/train_step.py --steps 30 --warmup 5 > $R/gpurun_out/ab_$w.log 2>&1 || { tail -3 $R/gpurun_out/ab_$w.log; exit 1; }
done
python3 - <<PY
import csv
def load(w):
    d={}
    for r in csv.DictReader(open("$R/gpurun_out/ab_%s/run_kernel_stats.csv"%w)):
        n=r["Name"].replace("stgcn::(anonymous namespace)::","").replace("void ","").split("(")[0]
        d[n]=(int(r["Calls"]),float(r["AverageNs"])/1e3)
    return d
a,b=load("A"),load("B")
for k in sorted(set(a)|set(b), key=lambda k:-(a.get(k,(0,0))[1]*a.get(k,(0,0))[0])):
    ca,ta=a.get(k,(0,0)); cb,tb=b.get(k,(0,0))
    if max(ta,tb)>8: print(f"{k[:60]:60s} A {ca:3d} x {ta:8.1f} us   B {cb:3d} x {tb:8.1f} us")
PY
