#!/bin/bash
# round 6: the driver's 20-frame form at 12 frames per extract launch under different leading batches, against 8 frames per launch
#   gpurun -- 'bash tools/ramp_probe_r06.sh <outdir>'
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:?outdir}; mkdir -p $O; cd $R
export SSF_PRODUCT_VARIANT=lab      # (the ramp switch lives in the lab build of the sources)
run() { local b=$1 ramp=$2 steps=$3; if [ "$ramp" = default ]; then unset SSF_SEQ_RAMP; else export SSF_SEQ_RAMP=$ramp; fi
  python bench.py --gpus 1 --steps $steps --warmup 5 --extract-batch $b --extras 0 --cpu-frames 0 2>/dev/null | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d.get('pipeline_fill') or {}
print('batch $b ramp $ramp steps $steps: %.0f frames/s  frac %.3f  frame0 done %.0f us  batches %s' % (d['value'], d['roofline']['frac'], (f.get('frame_done_us') or [0])[0], [(b['frames']) for b in f.get('extract_batches_launched',[])]))" >> $O/summary.txt; }
for r in 1 2 3; do
  run 8 default 20; run 12 default 20; run 12 3,5 20; run 12 4,6 20; run 12 3,6 20; run 12 2,6 20; run 12 3,5,8 20
done
run 8 default 1200; run 12 default 1200; run 12 3,5 1200; run 12 3,5,8 1200
cat $O/summary.txt
