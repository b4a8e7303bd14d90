#!/bin/bash
# Run ON THE GPU BOX via gpurun, in two calls (each well under the 1200 s limit):
#   gpurun --timeout 1200 -- 'bash tools/final_round.sh <tag> tests'     GPU suite + the bench lines of the round
#   gpurun --timeout 1200 -- 'bash tools/final_round.sh <tag> profiles'  rocprofv3 stats + PMC passes (fused, two-stage, f32, training)
# then, in the container:  for t in <tag> <tag>_twostage <tag>_f32; do python tools/summarize_profiles.py $t; done
TAG=${1:-r02}; WHAT=${2:-tests}
O=gpurun_out; mkdir -p $O
if [ "$WHAT" = tests ]; then
    bash tools/gpu_round.sh $TAG || exit 1
    timeout -k 10 120 python tools/train_step.py --steps 100 > $O/${TAG}_train_step.log 2>&1; tail -1 $O/${TAG}_train_step.log | cut -c1-300
else
    bash tools/collect_profiles.sh $TAG || { echo "collect $TAG failed"; exit 1; }
    bash tools/collect_profiles.sh ${TAG}_twostage --no-fuse || { echo "collect twostage failed"; exit 1; }
    bash tools/collect_profiles.sh ${TAG}_f32 --math f32 || { echo "collect f32 failed"; exit 1; }
    bash tools/profile_train.sh $TAG || exit 1
fi
