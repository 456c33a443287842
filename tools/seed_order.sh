#!/bin/bash
# BASELINE config 3 with the seeded rows in the order a map built by the pipeline has (--seed-order image) beside the BASELINE workload
# (random order of the seeding), alternated on one box.   gpurun -- 'bash tools/seed_order.sh <outdir>'
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:?outdir}; mkdir -p $O; cd $R
for r in 1 2; do for o in seeded image; do
  timeout 400 python bench.py --config 3 --seed-order $o --extras 0 --cpu-frames 0 > $O/c3_${o}_$r.json 2> $O/c3_${o}_$r.err
  tail -n 1 $O/c3_${o}_$r.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); pk=d['per_kernel']; print('$o', round(d['value'],1), {k: round(pk[k]['avg_us'],1) for k in ('icp_accumulate','match','update_insert','reorder_move_icp') if k in pk})" >> $O/summary.txt
done; done
cat $O/summary.txt
