#!/bin/bash
# Run ON THE GPU BOX via gpurun: the GPU suite (stops at the first failure), then the K1 timer and a short default bench.
#   gpurun --timeout 900 -- 'bash tools/gpu_quick.sh <tag> [pytest -k expr]'
TAG=${1:-q}; KEXPR=${2:-}
O=gpurun_out; mkdir -p $O
if [ -n "$KEXPR" ]; then
  timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "$KEXPR" > $O/${TAG}_pytest.log 2>&1
else
  timeout -k 10 800 python -m pytest tests -m gpu -q -x > $O/${TAG}_pytest.log 2>&1
fi
rc=$?; tail -4 $O/${TAG}_pytest.log
if [ $rc -ne 0 ]; then grep -n "Error\|error\|assert" $O/${TAG}_pytest.log | head -20; exit $rc; fi
timeout -k 10 200 python tools/time_k1.py > $O/${TAG}_k1.log 2>&1 || { tail -5 $O/${TAG}_k1.log; exit 1; }
cat $O/${TAG}_k1.log
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --alt-steps 0 --no-cpu-baseline > $O/${TAG}_bench.log 2>&1 || { tail -5 $O/${TAG}_bench.log; exit 1; }
python - <<PY
import json
d=json.loads([l for l in open("$O/${TAG}_bench.log") if l.startswith("{")][-1])
print("bench:", d["value"], "ms/step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"], "steady", d.get("steady_state"))
PY
