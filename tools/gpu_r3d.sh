#!/bin/bash
# round 3: KF7 (fp16 + scaled-e4m3 residuals) — its tests, then A/B bench lines bf16x3 vs f16mx in separate processes
O=gpurun_out; mkdir -p $O; TAG=${1:-r03d}
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "f16mx or wide_frame" > $O/${TAG}_pytest.log 2>&1
rc=$?; tail -12 $O/${TAG}_pytest.log
if [ $rc -ne 0 ]; then grep -n "Error\|error\|assert" $O/${TAG}_pytest.log | head -30; exit $rc; fi
for m in f16mx bf16x3 f16mx; do
  timeout -k 10 300 python bench.py --math $m --steps 20 --warmup 5 --alt-steps 0 --train-steps 0 --other-steps 0 --no-cpu-baseline > $O/${TAG}_bench_$m.log 2>&1 || { tail -20 $O/${TAG}_bench_$m.log; exit 1; }
  python - <<PY
import json
d=json.loads([l for l in open("$O/${TAG}_bench_$m.log") if l.startswith("{")][-1])
print("$m", d["value"], "kernel", d["roofline"]["kernel"], d["roofline"]["kernel_ms"], "steady", d["steady_state"]["value"], d["steady_state"]["kernel_ms"], d["steady_state"]["roofline_frac"])
PY
done
