#!/bin/bash
# host frames of a sequence with the runtime's SDMA copies against blit-kernel copies (HSA_ENABLE_SDMA=0), alternated on one box
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:?outdir}; mkdir -p $O; cd $R
for r in 1 2 3; do for sd in 1 0; do for pf in 0 1; do
  HSA_ENABLE_SDMA=$sd PROBE_KINDS=pageable PROBE_PREFILTER=$pf python tools/host_buffer_probe.py 2>/dev/null | grep frames | sed "s/^/sdma=$sd /" >> $O/summary.txt
done; done; done
cat $O/summary.txt
