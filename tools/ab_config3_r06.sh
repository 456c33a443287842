#!/bin/bash
# round 6: BASELINE config 3, the product against the lab build's tile-sorted copy of the visible rows (lab/tile_bins.inc) at the
# current sources, alternated, with the per-kernel hipEvent brackets beside the frame rate
#   gpurun -- 'bash tools/ab_config3_r06.sh <outdir> [rounds]'
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:?outdir}; mkdir -p $O; cd $R
N=${2:-2}
run() { # tag, env...
  local tag=$1; shift
  env "$@" timeout 400 python bench.py --config 3 --extras 0 --cpu-frames 0 2> $O/$tag.err | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['per_kernel']
print('$tag', round(d['value'],1), 'frames/s  frame_frac', round(d['frame_roofline']['frac'],3), ' '.join('%s %.1f' % (n, k[n]['avg_us']) for n in ('icp_accumulate','match','bin_rows','update_insert','reorder_move_icp','reorder_move') if n in k))" >> $O/summary.txt
}
for r in $(seq 1 $N); do
  run product_$r
  run lab_sorted_$r SSF_PRODUCT_VARIANT=lab SSF_BIN_MIN_ROWS=300000
  run lab_unsorted_$r SSF_PRODUCT_VARIANT=lab
done
cat $O/summary.txt
