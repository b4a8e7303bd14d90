#!/bin/bash
# generic chain after the subset batching: parity, then timing + per-dispatch trace
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "generic or tcn_gcn_unit or eval_mode_backward_unit_agcn or patch_embedding or moment_form or more_clips" > gpurun_out/r3e_pytest.log 2>&1
rc=$?; tail -5 gpurun_out/r3e_pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python tools/generic_unit_step.py --steps 30 | tee gpurun_out/r3e_generic.json || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r3e -o run -- python3 $GRAFT_REPO_ROOT/tools/generic_unit_step.py --steps 10 > $GRAFT_REPO_ROOT/gpurun_out/r3e_trace.log 2>&1
tail -2 $GRAFT_REPO_ROOT/gpurun_out/r3e_trace.log
