#!/bin/bash
# BASELINE config 3 (and the default workload as a control): the product against variant builds, alternated, per-kernel brackets beside the rate
#   gpurun -- 'bash tools/ab_config3_variants.sh <outdir> <rounds> <variant> ...'
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:?outdir}; mkdir -p $O; cd $R
N=${2:?rounds}; shift 2
run() { local tag=$1 cfg=$2; timeout 400 python bench.py --config $cfg --extras 0 --cpu-frames 0 2> $O/${tag}_c$cfg.err | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['per_kernel']
print('$tag config $cfg', round(d['value'],1), 'frames/s  frame_frac', round(d['frame_roofline']['frac'],3), ' '.join('%s %.1f' % (n, k[n]['avg_us']) for n in ('icp_accumulate','match','bin_rows','update_insert','reorder_move_icp','reorder_move') if n in k))" >> $O/summary.txt; }
for r in $(seq 1 $N); do for V in product "$@"; do
  unset SSF_PRODUCT_VARIANT; [ $V != product ] && export SSF_PRODUCT_VARIANT=$V
  run ${V}_$r 3
done; done
for V in product "$@"; do unset SSF_PRODUCT_VARIANT; [ $V != product ] && export SSF_PRODUCT_VARIANT=$V; run ${V}_ctl 2; done
cat $O/summary.txt
