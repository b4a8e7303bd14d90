#!/bin/bash
# Run ON THE GPU BOX via gpurun: prices the phases of agcn_bwd_gather_kernel with the diagnostic library's run-time switches
# (STGCN_ABLATE bits: 1 loads off, 2 h FMAs off, 4 Gram MFMAs off, 8 h exchange off, 16 feature computation off) — kernel times
# from rocprofv3 --stats of the training step (results are wrong with a switch on; only the gather kernel's time is read).
R=$PWD
export STGCN_LIB=$R/st-gcn-altformer_amd/stgcn_amd/libstgcn_hip_abl.so
cd /tmp && export TMPDIR=/tmp
for m in 0 2 8 10 4 16 30 31; do
  STGCN_ABLATE=$m timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/price_$m -o run -- python3 $R/tools/train_step.py --steps 10 --warmup 3 > $R/gpurun_out/price_$m.log 2>&1 || { tail -3 $R/gpurun_out/price_$m.log; exit 1; }
  python3 - <<PY
import csv
for r in csv.DictReader(open("$R/gpurun_out/price_$m/run_kernel_stats.csv")):
    if "agcn_bwd_gather" in r["Name"]: print("ablate $m: gather %.1f us" % (float(r["AverageNs"])/1e3))
PY
done
