#!/bin/bash
# The driver's 20-frame form under different leading extract batches (SSF_SEQ_RAMP): bash tools/ramp_probe.sh "2,4" "1,2,4" ...
export SSF_PRODUCT_VARIANT=lab      # (the switch lives in the lab build of the sources)
for ramp in "$@"; do
  for r in 1 2 3; do
    if [ "$ramp" = default ]; then unset SSF_SEQ_RAMP; else export SSF_SEQ_RAMP=$ramp; fi
    python bench.py --gpus 1 --steps 20 --warmup 5 --extras 0 --cpu-frames 0 2>/dev/null | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d.get('pipeline_fill') or {}
print('ramp $ramp run $r: %.0f frames/s  frame0 done %.0f us  region %.0f us  batches %s' % (d['value'], (f.get('frame_done_us') or [0])[0], f.get('region_us',0), [(b['frames']) for b in f.get('extract_batches_launched',[])]))"
  done
done
