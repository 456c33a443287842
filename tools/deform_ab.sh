#!/bin/bash
# gpurun -- 'bash tools/deform_ab.sh <outdir> <variant> ...'
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:?outdir}; mkdir -p $O; cd $R; shift
for r in 1 2; do for V in product "$@"; do
  unset SSF_PRODUCT_VARIANT; [ $V != product ] && export SSF_PRODUCT_VARIANT=$V
  timeout 300 python tools/deform_probe.py 1000000 2>/dev/null | tail -n 1 >> $O/deform.txt
done; done
for V in product "$@"; do
  unset SSF_PRODUCT_VARIANT; [ $V != product ] && export SSF_PRODUCT_VARIANT=$V
  timeout 300 python tools/deform_probe.py 1000000 coherent 2>/dev/null | tail -n 1 >> $O/deform.txt
done
cat $O/deform.txt
