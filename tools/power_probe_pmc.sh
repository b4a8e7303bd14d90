#!/bin/bash
# Run ON THE GPU BOX via gpurun: the power probe (same instruction stream on random / zero operands) twice — once bare
# (HIP-event kernel times) and once under rocprofv3 PMC (clock = GRBM_GUI_ACTIVE / 8 / duration, MFMA-busy) — condensed
# into gpurun_out/<tag>_power_probe.txt (copy it to profiles/).   gpurun -- 'bash tools/power_probe_pmc.sh r03'
TAG=${1:-r03}; O=$PWD/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 300 python3 tools/power_probe.py > $O/${TAG}_power_bare.log 2>&1 || { tail -5 $O/${TAG}_power_bare.log; exit 1; }
D=$O/prof_power_$TAG; rm -rf $D
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $D -- python3 tools/power_probe.py > $O/${TAG}_power_pmc.log 2>&1 || { tail -5 $O/${TAG}_power_pmc.log; exit 1; }
F=$(find $D -name "*counter_collection.csv" | head -1)
python3 - "$F" $O/${TAG}_power_bare.log > $O/${TAG}_power_probe.txt <<'PY'
import csv, sys, collections
rows = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    if "stem_bf16_v6_kernel" in r["Kernel_Name"]:
        d = rows[int(r["Dispatch_Id"])]
        d[r["Counter_Name"]] = float(r["Counter_Value"])
        d["dur"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
ids = sorted(rows)
bare = [l.strip() for l in open(sys.argv[2]) if "fused kernel" in l]
print("tools/power_probe.py: the fused stem kernel (KF6, 256 clips, T=180, V=22, bf16x3), same instruction stream, operands swapped")
print("bare run (HIP events, 200 launches each):")
for l in bare:
    print("   ", l)
n = len(ids) // 5                     # five configurations of 220 launches each, in the order printed above
names = [l.split(":")[0].strip() for l in bare] or ["cfg%d" % i for i in range(5)]
print("under rocprofv3 --pmc (last 100 launches of each configuration): kernel us, clock GHz, MFMA-busy fraction")
for c in range(5):
    seg = ids[c * n:(c + 1) * n][-100:]
    dur = sum(rows[i]["dur"] for i in seg) / len(seg)
    cyc = sum(rows[i]["GRBM_GUI_ACTIVE"] for i in seg) / len(seg) / 8.0
    busy = sum(rows[i]["SQ_VALU_MFMA_BUSY_CYCLES"] for i in seg) / len(seg)
    print(f"    {names[c % len(names)]:34s}: {dur / 1e3:7.1f} us   {cyc / dur:5.3f} GHz   busy {busy / (1024.0 * cyc):5.3f}")
PY
cat $O/${TAG}_power_probe.txt
