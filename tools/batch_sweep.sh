#!/bin/bash
# The metric's workload at 5..12 frames per extract launch: steady state (1200 frames), the driver's 20-frame form, and the live
# launch time of the dominant kernel (a launch's workgroups come in "rounds" of the part's resident slots: 315 tiles per frame).
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:?outdir}; mkdir -p $O; cd $R
for r in 1 2; do for b in ${BATCHES:-8 5 6 7 9 10 11 12}; do
  timeout 300 python bench.py --extract-batch $b --extras 0 --cpu-frames 0 2>/dev/null | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('steady batch $b run $r', round(d['value'],1), 'pass_us', round(r['avg_launch_us'],2), 'frac', round(r['frac'],4), 'seq_ms', d.get('sequential_ms_per_frame'))" >> $O/summary.txt
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --extract-batch $b --extras 0 --cpu-frames 0 2>/dev/null | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('s20    batch $b run $r', round(d['value'],1), 'pass_us', round(r['avg_launch_us'],2), 'frac', round(r['frac'],4))" >> $O/summary.txt
done; done
cat $O/summary.txt
