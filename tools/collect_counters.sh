#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: rocprofv3 kernel-trace statistics + PMC passes (each counter group in
# its own run, never combined with a trace) of ANY program of this repo; raw CSVs -> gpurun_out/prof_<tag>/, condensed by
# tools/summarize_profiles.py <tag> into profiles/<tag>_{kernel_stats.csv,counters.json}.
#   gpurun -- 'bash tools/collect_counters.sh r03_train tools/train_step.py --steps 20 --warmup 5'
# (rocprofv3 gets the program itself after `--`: python3 <script> ..., never a wrapper)
set -o pipefail
TAG=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out/prof_$TAG
rm -rf "$R"; mkdir -p "$R"
echo "python3 $*" > "$R/program.txt"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/trace" -- python3 "$@" > "$R/trace.log" 2>&1 || { tail -5 "$R/trace.log"; exit 1; }
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$R/pmc_fetch" -- python3 "$@" > "$R/pmc_fetch.log" 2>&1 || { tail -5 "$R/pmc_fetch.log"; exit 1; }
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$R/pmc_write" -- python3 "$@" > "$R/pmc_write.log" 2>&1 || { tail -5 "$R/pmc_write.log"; exit 1; }
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --output-format csv -d "$R/pmc_sq1" -- python3 "$@" > "$R/pmc_sq1.log" 2>&1 || { tail -5 "$R/pmc_sq1.log"; exit 1; }
tail -2 "$R/trace.log" | cut -c1-300
echo "counters collected in $R"
