#!/bin/bash
# Run ON THE GPU BOX via gpurun: rocprofv3 kernel statistics of tools/capture_generic_unit.py (a TCN_GCN_unit-shaped training step).
O=$PWD/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
D=$O/prof_generic; rm -rf $D
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o tr -- python3 tools/capture_generic_unit.py > $O/generic_prof.log 2>&1 || { tail -5 $O/generic_prof.log; exit 1; }
F=$(find $D -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:22]:
    n=r['Name'].replace('void ','').replace('stgcn::','').replace('(anonymous namespace)::','').split('(')[0]
    print(f"{n[:60]:60s} calls {int(r['Calls']):5d}  {float(r['TotalDurationNs'])/1e6:8.2f} ms  {100*float(r['TotalDurationNs'])/tot:5.1f} %  avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
