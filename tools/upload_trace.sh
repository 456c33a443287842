#!/bin/bash
# rocprofv3 kernel + memory-copy trace of a sequence of HOST frames (tools/host_buffer_probe.py, pageable): when do the copies run,
# how long do they take, what waits for them.   gpurun -- 'bash tools/upload_trace.sh <outdir>'
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:?outdir}; mkdir -p $O
cd $R; PROBE_KINDS=device,pageable python tools/host_buffer_probe.py > $O/untraced.txt 2>&1
cd /tmp && export TMPDIR=/tmp
PROBE_NF=300 PROBE_KINDS=pageable timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d $O/t -o trace -- python $R/tools/host_buffer_probe.py > $O/traced.txt 2>&1
DB=$(find $O/t -name "*.db" | head -1)
python $R/tools/upload_trace.py $DB > $O/upload_trace.txt 2>&1
find $O -name "*.db" -size +30M -delete
cat $O/untraced.txt $O/traced.txt | grep -v amdgpu; head -60 $O/upload_trace.txt
