#!/bin/bash
# same-box comparison of the product against SEVERAL variant builds, alternated: the metric's workload and BASELINE config 3
#   gpurun -- 'bash tools/ab_variants.sh <outdir> <rounds> <tag> [<tag> ...]'
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:?outdir}; mkdir -p $O; cd $R
N=${2:?rounds}; shift 2
run() { local tag=$1; shift; timeout 400 python bench.py "$@" --extras 0 --cpu-frames 0 --profile-frames 0 2> $O/$tag.err | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', round(d['value'],1))" >> $O/summary.txt; }
for r in $(seq 1 $N); do
  unset SSF_PRODUCT_VARIANT; run product_c2_$r; run product_c3_$r --config 3
  for V in "$@"; do export SSF_PRODUCT_VARIANT=$V; run ${V}_c2_$r; run ${V}_c3_$r --config 3; done
done
sort $O/summary.txt
