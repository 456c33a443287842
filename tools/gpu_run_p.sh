#!/bin/bash
# depth pre-filter kernel variants inside the frame: BASELINE config 5 (pre-filter on), same box
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r02t}
mkdir -p $O
cd $R
for rep in 1 2 3; do
  SSF_BILATERAL_GENERIC=1 timeout 300 python bench.py --config 5 --extras 0 --cpu-frames 0 > $O/c5_generic_$rep.json 2>> $O/err.log
  SSF_BIL_WAVES=2 timeout 300 python bench.py --config 5 --extras 0 --cpu-frames 0 > $O/c5_w2_$rep.json 2>> $O/err.log
  SSF_BIL_WAVES=3 timeout 300 python bench.py --config 5 --extras 0 --cpu-frames 0 > $O/c5_w3_$rep.json 2>> $O/err.log
  SSF_BIL_WAVES=4 timeout 300 python bench.py --config 5 --extras 0 --cpu-frames 0 > $O/c5_w4_$rep.json 2>> $O/err.log
done
for f in $O/*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(d["value"],1), d["stage_ms"])
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
