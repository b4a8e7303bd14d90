#!/bin/bash
# full GPU suite, then the generic unit steps, the stem training step and a trace of the generic steps
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r3f_pytest.log 2>&1
rc=$?; tail -4 gpurun_out/r3f_pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python tools/generic_unit_step.py --steps 30 | tee gpurun_out/r3f_generic.json || exit 1
timeout -k 10 200 python tools/train_step.py --steps 30 --warmup 5 | tee gpurun_out/r3f_train.json || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r3f -o run -- python3 $GRAFT_REPO_ROOT/tools/generic_unit_step.py --steps 10 > $GRAFT_REPO_ROOT/gpurun_out/r3f_trace.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/r3f_trace.log
