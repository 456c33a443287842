#!/bin/bash
# round 6: the relabelling pass with the replay ahead of the decisions (64 registers) and direct-to-LDS tile loads, against the product
# (the arms it compares -- -DSSF_PASS_LDSDMA / -DSSF_PASS_REPLAY_FIRST -- exist in the tree at the commit named in profiles/pass_ldsdma_r06.txt; they were removed afterwards)
#   gpurun -- 'bash tools/ab_pass_r06.sh <outdir> <rounds> <variant> [<variant> ...]'
R=${GRAFT_REPO_ROOT:-$(pwd)}; OD=${1:?outdir}; O=$R/gpurun_out/$OD; mkdir -p $O; cd $R
N=${2:?rounds}; shift 2
[ -x tools/probe/bin/glds_align ] && tools/probe/bin/glds_align > $O/glds_align.txt 2>&1
for V in "$@"; do
  SSF_PRODUCT_VARIANT=$V timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "relabelling_pass or sequence_bit_exact or odd_image or segmentation_parameter or energy_parameter or edge_cases or hostile or pipelined_equals" > $O/parity_$V.log 2>&1
  echo "$V parity rc=$? $(tail -n 1 $O/parity_$V.log)" >> $O/summary.txt
done
bash tools/kernel_ab.sh $OD $N update_pass_rgbd,update_pass_rgb,render_moments product "$@" > /dev/null 2>&1
for r in $(seq 1 $N); do for V in product "$@"; do
  unset SSF_PRODUCT_VARIANT; [ $V != product ] && export SSF_PRODUCT_VARIANT=$V
  timeout 300 python bench.py --extras 0 --cpu-frames 0 --steps 20 --warmup 5 2>/dev/null | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$V driver-form run $r', round(d['value'],1), 'frac', d['roofline']['frac'], 'steady', d['config'].get('steady_state_frames_per_sec'))" >> $O/driver_form.txt
done; done
cat $O/summary.txt $O/kernel_ab.txt $O/driver_form.txt
