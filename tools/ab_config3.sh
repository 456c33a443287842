#!/bin/bash
# same-box A/B of BASELINE config 3: the product against the lab build's tile-sorted rows, alternated
#   gpurun -- 'bash tools/ab_config3.sh <outdir> [rounds]'
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:?outdir}; mkdir -p $O; cd $R
N=${2:-2}
run() { # tag, env...
  local tag=$1; shift
  env "$@" timeout 400 python bench.py --config 3 --extras 0 --cpu-frames 0 --profile-frames 0 2> $O/$tag.err | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', round(d['value'],1))" >> $O/summary.txt
}
for r in $(seq 1 $N); do
  run product_$r
  run sorted_$r SSF_PRODUCT_VARIANT=lab SSF_BIN_MIN_ROWS=300000
  run sorted_nowaiter_$r SSF_PRODUCT_VARIANT=lab SSF_BIN_MIN_ROWS=300000 SSF_NO_MATCH_IN_WAITER=1
  run lab_unsorted_$r SSF_PRODUCT_VARIANT=lab
done
cat $O/summary.txt
