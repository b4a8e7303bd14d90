#!/bin/bash
# Run ON THE GPU BOX via gpurun: counters of the generic attention kernel alone at two shapes (each counter group in its own pass)
R=$PWD; export TMPDIR=/tmp; cd /tmp
for shp in "64 64 64 90" "256 256 64 23"; do
  set -- $shp; tag=k1g_$1_$4
  for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
             "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
             "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM"; do
    g=$(echo $grp | cut -c1-14 | tr ' ' '_')
    timeout -k 10 120 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/$tag/$g -o run -- python3 $R/tools/time_generic_attention.py --cin $1 --cout $2 --clips $3 --frames $4 > $R/gpurun_out/$tag.$g.log 2>&1 || tail -3 $R/gpurun_out/$tag.$g.log
  done
done
python3 - <<PY
import csv, glob, collections
for tag in ("k1g_64_90","k1g_256_23"):
    agg=collections.defaultdict(list)
    for f in glob.glob("$R/gpurun_out/%s/*/run_counter_collection.csv"%tag):
        for r in csv.DictReader(open(f)):
            if "attention_generic_mfma" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(tag, {k: round(sum(v)/len(v)) for k,v in sorted(agg.items())})
PY
