#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02i
mkdir -p $O
cd $R
timeout 200 python tools/repro_config2.py 10 > $O/repro_default.txt 2>&1; echo "rc=$?" >> $O/repro_default.txt
SSF_UPLOAD_PAGEABLE=1 timeout 200 python tools/repro_config2.py 10 > $O/repro_pageable.txt 2>&1; echo "rc=$?" >> $O/repro_pageable.txt
SSF_ICP_AHEAD=0 timeout 200 python tools/repro_config2.py 10 > $O/repro_noahead.txt 2>&1; echo "rc=$?" >> $O/repro_noahead.txt
timeout 200 python tools/startup_probe.py > $O/startup_probe.txt 2>&1
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
echo done
