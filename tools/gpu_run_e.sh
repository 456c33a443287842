#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02e
mkdir -p $O
cd $R
SSF_PROFILE_PER_PASS=1 timeout 300 python bench.py --extras 0 --cpu-frames 0 --steps 96 --profile-frames 32 > $O/per_pass.json 2> $O/per_pass.err
SSF_PROFILE_PER_PASS=1 timeout 300 python bench.py --extras 0 --cpu-frames 0 --steps 96 --profile-frames 32 --pipeline-depth 0 --extract-batch 1 > $O/per_pass_latency.json 2>> $O/per_pass.err
echo done
