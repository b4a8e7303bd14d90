#!/bin/bash
# Run ON THE GPU BOX via gpurun: HBM-side read bytes per kernel of the training step (one counter per pass: FETCH_SIZE).
#   gpurun --timeout 600 -- 'bash tools/pmc_fetch_train.sh'
export TMPDIR=/tmp
O=$PWD/gpurun_out; mkdir -p $O; rm -rf /tmp/pm
timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pm -o p -- python3 tools/train_step.py --steps 4 --warmup 2 > $O/pm_fetch.log 2>&1 || { tail -3 $O/pm_fetch.log; exit 1; }
F=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
python3 - "$F" <<'PY'
import csv, collections, sys
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == "FETCH_SIZE":
        acc[r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("stgcn::", "").split("(")[0][:44]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]) / len(kv[1])):
    mb = sum(v) / len(v) * 1024 * 2 / 1e6            # KiB -> bytes, gfx950 x2 correction (MI355X_MICROARCH.md)
    if mb > 20:
        print(f"{k:46s} {mb:9.0f} MB read per launch ({len(v)} launches)")
PY
