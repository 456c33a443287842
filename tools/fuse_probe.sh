#!/bin/bash
# tools/fuse_probe.py on the lab build and on ablation variants of the out-of-view arm (tools/build_variant.sh <tag> -DSSF_EXPERIMENTS ...):
#   gpurun -- 'bash tools/fuse_probe.sh <outdir> <tag> [<tag> ...]'
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:?outdir}; mkdir -p $O; cd $R; shift
unset SSF_PRODUCT_VARIANT
( echo "== lab"; timeout 300 python tools/fuse_probe.py 2>&1 | grep -v amdgpu.ids ) >> $O/fuse_probe.txt
for V in "$@"; do ( echo "== $V"; SSF_PRODUCT_VARIANT=$V timeout 300 python tools/fuse_probe.py 2>&1 | grep -v amdgpu.ids ) >> $O/fuse_probe.txt; done
cat $O/fuse_probe.txt
