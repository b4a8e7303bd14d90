#!/bin/bash
# round 3, first GPU call: suite + default bench + the 2-rank gloo rehearsal of the new N>1 default (strong, configs[4])
O=gpurun_out; mkdir -p $O; TAG=r03a
timeout -k 10 900 python -m pytest tests -m gpu -q -x --durations=10 > $O/${TAG}_pytest.log 2>&1
rc=$?; tail -6 $O/${TAG}_pytest.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT; exit 1; fi
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_default.log 2>&1 || { tail -20 $O/${TAG}_bench_default.log; exit 1; }
tail -1 $O/${TAG}_bench_default.log | cut -c1-2500
STGCN_DIST_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --steady-steps 20 --alt-steps 0 > $O/${TAG}_bench_n2_gloo.log 2>&1 || { tail -20 $O/${TAG}_bench_n2_gloo.log; exit 1; }
tail -1 $O/${TAG}_bench_n2_gloo.log | cut -c1-1500
STGCN_DIST_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --layout ntvc --weak --steps 5 --warmup 2 --steady-steps 0 --alt-steps 0 > $O/${TAG}_bench_n2_ntvc.log 2>&1 || { tail -20 $O/${TAG}_bench_n2_ntvc.log; exit 1; }
tail -1 $O/${TAG}_bench_n2_ntvc.log | cut -c1-800
