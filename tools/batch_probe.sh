#!/bin/bash
# BASELINE config 5 and the metric's workload with the pre-filter at 8 / 12 / 16 frames per extract launch (variant b16: -DSSF_MAX_BATCH=16)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:?outdir}; mkdir -p $O; cd $R
export SSF_PRODUCT_VARIANT=b16
for r in 1 2; do for b in 8 12 16; do
  timeout 400 python bench.py --config 5 --extract-batch $b --extras 0 --cpu-frames 0 --profile-frames 0 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config5 batch $b', round(d['value'],1))" >> $O/summary.txt
done; done
cat $O/summary.txt
