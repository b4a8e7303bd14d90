#!/bin/bash
# Run ON THE GPU BOX via gpurun: A/B of two builds of the library on ONE box, interleaved processes (steady-state block of bench.py).
#   gpurun -- 'bash tools/ab_libs.sh prod2 [bench args]'      (B = st-gcn-altformer_amd/stgcn_amd/libstgcn_hip_<variant>.so)
V=$1; shift
O=gpurun_out; mkdir -p $O
B=$PWD/st-gcn-altformer_amd/stgcn_amd/libstgcn_hip_$V.so
STGCN_LIB=$B timeout -k 10 300 python -m pytest tests -m gpu -q -x -k "stem_vs_golden or ragged or persistent or layout_fusion" > $O/ab_${V}_pytest.log 2>&1; tail -2 $O/ab_${V}_pytest.log
for r in 1 2 3; do
  for which in A B; do
    if [ $which = B ]; then export STGCN_LIB=$B; else unset STGCN_LIB; fi
    timeout -k 10 200 python bench.py --steps 20 --warmup 5 --steady-steps 400 --alt-steps 0 --train-steps 0 --other-steps 0 --no-cpu-baseline "$@" > $O/ab_${V}_$which$r.log 2>&1 || { tail -5 $O/ab_${V}_$which$r.log; exit 1; }
    python - <<PY
import json
d=json.loads([l for l in open("$O/ab_${V}_$which$r.log") if l.startswith("{")][-1])
print("$which$r", "driver-form", d["value"], "kernel_ms", d["roofline"]["kernel_ms"], "| steady", d["steady_state"]["value"], "kernel_ms", d["steady_state"]["kernel_ms"])
PY
  done
done
