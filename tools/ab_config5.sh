#!/bin/bash
# same-box A/B of BASELINE config 5 (pre-filter in the frame) and of the pre-filter kernel's own time: product against variant builds
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:?outdir}; mkdir -p $O; cd $R; shift
for r in 1 2; do for v in product "$@"; do
  if [ $v = product ]; then unset SSF_PRODUCT_VARIANT; else export SSF_PRODUCT_VARIANT=$v; fi
  timeout 400 python bench.py --config 5 --extras 0 --cpu-frames 0 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); pk=d['per_kernel']; print('$v', round(d['value'],1), 'bilateral', round(pk.get('bilateral_prefilter',{}).get('avg_us',0),1), 'us per launch')" >> $O/summary.txt
done; done
cat $O/summary.txt
