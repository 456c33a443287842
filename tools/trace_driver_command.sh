#!/bin/bash
# rocprofv3 --kernel-trace --stats of the DRIVER'S exact command (20 timed frames after 5 warm-up frames; without the extras / CPU legs,
# which run after the timed region anyway):   gpurun -- 'bash tools/trace_driver_command.sh <outdir>'
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:?outdir}; mkdir -p $O
NOTE="bench.py --gpus 1 --steps 20 --warmup 5 --extras 0 --cpu-frames 0 (the command the driver runs, without the extras / CPU legs)"
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --extras 0 --cpu-frames 0 > $O/trace.log 2>&1 )
cd $R
DB=$(find $O/trace -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB $O/rocprof_summary.txt "$NOTE" $O/rocprof.json 307200 | head -14
[ -n "$DB" ] && python tools/rocprof_dist.py $DB > $O/rocprof_distribution.txt 2>&1
find $O -name "*.db" -delete
