#!/bin/bash
# round 3: counters for what was prose: training step, configs[2], the generic TCN_GCN_unit step, power probe
set -o pipefail
bash tools/collect_counters.sh r03_train tools/train_step.py --steps 20 --warmup 5 || exit 1
bash tools/collect_profiles.sh r03_cfg2 --clips-per-gpu 512 --frames 500 || exit 1
bash tools/collect_counters.sh r03_generic tools/generic_unit_step.py --steps 10 || exit 1
bash tools/power_probe_pmc.sh r03 || exit 1
