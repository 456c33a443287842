#!/bin/bash
# generic same-box A/B of one environment switch: tools/gpu_run_q.sh <out> <VAR> <value A> <value B>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r02u}
VAR=$2; A=$3; B=$4
mkdir -p $O
cd $R
for rep in 1 2 3; do
  for v in $A $B; do
    env $VAR=$v timeout 300 python bench.py --extras 0 --cpu-frames 0 > $O/s1200_${v}_$rep.json 2>> $O/err.log
    env $VAR=$v timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --extras 0 --cpu-frames 0 > $O/s20_${v}_$rep.json 2>> $O/err.log
    env $VAR=$v timeout 300 python bench.py --extras 0 --cpu-frames 0 --pipeline-depth 0 --extract-batch 1 --steps 200 > $O/lat_${v}_$rep.json 2>> $O/err.log
  done
done
for f in $O/*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(d["value"],1), {k: round(v,4) for k,v in d["stage_ms"].items()})
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
