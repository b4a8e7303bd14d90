#!/bin/bash
# Run ON THE GPU BOX via gpurun: GPU test suite, then the bench lines of the round (default, strong-scaling single rank,
# 2-rank gloo rehearsal of the strong mode).  A step that TIMES OUT ends the script (no further GPU step after a hang);
# an ordinary test failure does not stop the benches.
#   gpurun --timeout 1200 -- 'bash tools/gpu_round.sh <tag>'
TAG=${1:-r02}
O=gpurun_out
mkdir -p $O
step() {   # step <seconds> <log> <cmd...>
    local secs=$1 log=$2; shift 2
    timeout -k 10 "$secs" "$@" > "$log" 2>&1
    local rc=$?
    echo "[$(date +%H:%M:%S)] rc=$rc : $* -> $log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT - stopping"; exit 1; fi
    return $rc
}
step 900 $O/${TAG}_pytest.log python -m pytest tests -m gpu -q -x --durations=15
tail -5 $O/${TAG}_pytest.log
step 300 $O/${TAG}_bench_default.log python bench.py --steps 20 --warmup 5 && tail -1 $O/${TAG}_bench_default.log | cut -c1-1500
step 300 $O/${TAG}_bench_strong1.log python bench.py --global-clips 8192 --steps 5 --warmup 2 --steady-steps 20 --alt-steps 0 --no-cpu-baseline && tail -1 $O/${TAG}_bench_strong1.log | cut -c1-600
STGCN_DIST_BACKEND=gloo step 300 $O/${TAG}_bench_strong2_gloo.log python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --global-clips 8192 --steps 5 --warmup 2 --steady-steps 20 --alt-steps 0 && tail -1 $O/${TAG}_bench_strong2_gloo.log | cut -c1-600
step 300 $O/${TAG}_bench_nofuse.log python bench.py --no-fuse --steps 50 --warmup 10 --steady-steps 0 --alt-steps 0 --no-cpu-baseline && tail -1 $O/${TAG}_bench_nofuse.log | cut -c1-400
exit 0
