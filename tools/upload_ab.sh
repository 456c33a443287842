#!/bin/bash
# same-box comparison of upload-worker counts (variants built with tools/build_variant.sh upN -DSSF_UPLOAD_THREADS=N): host frames
# (pageable), with and without the pre-filter, 1200 frames.   gpurun -- 'bash tools/upload_ab.sh <outdir> <tag> [<tag> ...]'
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:?outdir}; mkdir -p $O; shift; cd $R
for round in 1 2; do
  for V in product "$@"; do
    if [ $V = product ]; then unset SSF_PRODUCT_VARIANT; else export SSF_PRODUCT_VARIANT=$V; fi
    for pf in 0 1; do
      PROBE_KINDS=pageable PROBE_PREFILTER=$pf python tools/host_buffer_probe.py 2>/dev/null | grep frames | sed "s/^/$V /" >> $O/summary.txt
    done
  done
done
nproc >> $O/summary.txt
cat $O/summary.txt
