#!/bin/bash
# round 3, final binary: every tracked profile re-collected (bench default, configs[3], configs[2], training step, generic unit
# steps), then the default bench line and the whole GPU suite
set -o pipefail
bash tools/collect_profiles.sh r03 || exit 1
bash tools/collect_profiles.sh r03_cfg3 --frames 200 --graph LMDHG || exit 1
bash tools/collect_profiles.sh r03_cfg2 --clips-per-gpu 512 --frames 500 || exit 1
bash tools/collect_counters.sh r03_train tools/train_step.py --steps 20 --warmup 5 || exit 1
bash tools/collect_counters.sh r03_generic tools/generic_unit_step.py --steps 10 || exit 1
timeout -k 10 300 python bench.py > gpurun_out/r03_final_bench.json 2> gpurun_out/r03_final_bench.err || { tail -5 gpurun_out/r03_final_bench.err; exit 1; }
tail -c 600 gpurun_out/r03_final_bench.json
