#!/bin/bash
# round 3: wide-frame KF6 (V = 46) — targeted tests first, then the configs[3] bench line
O=gpurun_out; mkdir -p $O; TAG=${1:-r03b}
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "lmdhg or 46 or ragged or full_size or layout or 256_channels or stem_vs" > $O/${TAG}_pytest_wide.log 2>&1
rc=$?; tail -8 $O/${TAG}_pytest_wide.log
if [ $rc -ne 0 ]; then grep -n "Error\|error\|assert" $O/${TAG}_pytest_wide.log | head -30; exit $rc; fi
timeout -k 10 300 python bench.py --frames 200 --graph LMDHG --steps 20 --warmup 5 --alt-steps 0 --train-steps 0 --no-cpu-baseline > $O/${TAG}_bench_cfg3.log 2>&1 || { tail -20 $O/${TAG}_bench_cfg3.log; exit 1; }
tail -1 $O/${TAG}_bench_cfg3.log | cut -c1-1800
