#!/bin/bash
# Is the relabelling pass bound by SCALAR or by VECTOR issue?  Lab builds with N extra dependent scalar (s_mul_i32) / vector (v_mul_f32)
# instructions per wave (tools/build_variant.sh sjunk150 -DSSF_EXPERIMENTS -DSSF_PASS_SJUNK=150, ...): per-launch time of the two passes.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:?outdir}; mkdir -p $O; cd $R; shift
for r in 1 2; do for V in "$@"; do
  SSF_PRODUCT_VARIANT=$V timeout 300 python bench.py --extras 0 --cpu-frames 0 --steps 240 2>/dev/null | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['per_kernel']
print('$V run $r', round(d['value'],1), 'rgbd_us', round(k['update_pass_rgbd']['avg_us'],2), 'rgb_us', round(k['update_pass_rgb']['avg_us'],2))" >> $O/junk.txt
done; done
cat $O/junk.txt
