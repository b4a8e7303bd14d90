#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: rocprofv3 kernel-trace stats and PMC passes of the
# default bench command; raw CSVs go to gpurun_out/prof_<tag>/, summaries are made by summarize_profiles.py.
#   gpurun -- 'bash tools/collect_profiles.sh r02 [extra bench args]'
# (rocprofv3 gets the program itself after `--`: python3 bench.py …, never a wrapper; counters in their own passes)
set -o pipefail
TAG=${1:-r02}; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out/prof_$TAG
mkdir -p "$R"
B="python3 bench.py --steps 10 --warmup 3 --no-extras $*"
# the trace pass runs longer: the --stats average includes the warm-up launches (cold clocks), which 13 launches do not dilute
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/trace" -- python3 bench.py --steps 100 --warmup 10 --no-extras $* > "$R/trace.log" 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$R/pmc_fetch" -- $B > "$R/pmc_fetch.log" 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$R/pmc_write" -- $B > "$R/pmc_write.log" 2>&1 || exit 1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --output-format csv -d "$R/pmc_sq1" -- $B > "$R/pmc_sq1.log" 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS \
  --output-format csv -d "$R/pmc_sq2" -- $B > "$R/pmc_sq2.log" 2>&1 || exit 1
tail -1 "$R/trace.log" | cut -c1-300
echo "profiles collected in $R"
