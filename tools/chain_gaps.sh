#!/bin/bash
# rocprofv3 kernel trace of the bench in pipelined and in latency mode -> per-link gaps of the track chain (tools/chain_gaps.py)
#   gpurun -- 'bash tools/chain_gaps.sh <outdir>'
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:?outdir}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in pipelined latency; do
  extra=""; [ $mode = latency ] && extra="--pipeline-depth 0 --extract-batch 1"
  timeout 600 rocprofv3 --kernel-trace -d $O/t_$mode -o trace -- python $R/bench.py --cpu-frames 0 --profile-frames 0 --extras 0 --steps 240 --warmup 8 $extra > $O/log_$mode.txt 2>&1
  DB=$(find $O/t_$mode -name "*.db" | head -1)
  python $R/tools/chain_gaps.py $DB > $O/chain_gaps_$mode.txt 2>&1
done
find $O -name "*.db" -delete
echo done
