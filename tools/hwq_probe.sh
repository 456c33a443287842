#!/bin/bash
# Is the "fifth queue" of DESIGN.md 4.2 the part's limit or the runtime's default of four hardware queues per process (GPU_MAX_HW_QUEUES)?
# The metric's workload at pipeline depth 2 and 3 with the default and with eight queues, alternated.  gpurun -- 'bash tools/hwq_probe.sh <outdir>'
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:?outdir}; mkdir -p $O; cd $R
for r in 1 2; do for q in default 8; do for d in 2 3; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  timeout 300 python bench.py --pipeline-depth $d --extras 0 --cpu-frames 0 --profile-frames 0 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('queues $q depth $d', round(d['value'],1))" >> $O/summary.txt
done; done; done
cat $O/summary.txt
