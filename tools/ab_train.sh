#!/bin/bash
# Run ON THE GPU BOX via gpurun: same-box A/B of the training step (tools/train_step.py) between the shipped library (A) and
# st-gcn-altformer_amd/stgcn_amd/libstgcn_hip_<variant>.so (B), interleaved processes.   gpurun -- 'bash tools/ab_train.sh <variant>'
B=$PWD/st-gcn-altformer_amd/stgcn_amd/libstgcn_hip_$1.so
for r in 1 2 3; do for w in A B; do if [ $w = B ]; then export STGCN_LIB=$B; else unset STGCN_LIB; fi
 echo -n "$w$r train ms/step "; timeout -k 10 120 python tools/train_step.py --steps 80 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
done; done
