#!/bin/bash
# Run ON THE GPU BOX via gpurun: SQ counters of the weight-gradient kernel (two passes, SQ counters only).
#   gpurun --timeout 600 -- 'bash tools/pmc_wgrad.sh'
export TMPDIR=/tmp
O=$PWD/gpurun_out; mkdir -p $O; rm -rf /tmp/pw1 /tmp/pw2
timeout -k 10 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --output-format csv -d /tmp/pw1 -o p -- python3 tools/time_tcn_bwd.py --val 0 > $O/pw1.log 2>&1 || { tail -3 $O/pw1.log; exit 1; }
timeout -k 10 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM \
  --output-format csv -d /tmp/pw2 -o p -- python3 tools/time_tcn_bwd.py --val 0 > $O/pw2.log 2>&1 || { tail -3 $O/pw2.log; exit 1; }
python3 - $(find /tmp/pw1 /tmp/pw2 -name "*counter_collection.csv") <<'PY'
import csv, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "wgrad" in k or "bn_relu_bwd" in k:
            acc[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} {sum(v)/len(v):16.0f}  ({len(v)})")
PY
