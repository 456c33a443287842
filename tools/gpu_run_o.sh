#!/bin/bash
# leading-batch sizes of a sequence (SSF_SEQ_RAMP) against the driver's 20-frame run and the 1200-frame run
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r02o}
mkdir -p $O
cd $R
for rep in 1 2 3; do
  for ramp in default 2 1 4 2,4 2,8 3 2,2; do
    if [ "$ramp" = default ]; then unset SSF_SEQ_RAMP; else export SSF_SEQ_RAMP=$ramp; fi
    timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --extras 0 --cpu-frames 0 > $O/s20_${ramp}_$rep.json 2>> $O/err.log
  done
done
for ramp in default 2 1; do
  if [ "$ramp" = default ]; then unset SSF_SEQ_RAMP; else export SSF_SEQ_RAMP=$ramp; fi
  timeout 300 python bench.py --extras 0 --cpu-frames 0 > $O/s1200_${ramp}.json 2>> $O/err.log
done
for f in $O/*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(d["value"],1))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
