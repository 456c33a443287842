#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02j
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 200 python tools/startup_probe.py > $O/startup_probe_urgent.txt 2>&1
SSF_URGENT_FIRST=0 timeout 200 python tools/startup_probe.py > $O/startup_probe_flat.txt 2>&1
for rep in 1 2 3; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --extras 0 --cpu-frames 0 > $O/new_s20_$rep.json 2>> $O/new.err
  SSF_URGENT_FIRST=0 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --extras 0 --cpu-frames 0 > $O/flat_s20_$rep.json 2>> $O/new.err
done
for rep in 1 2; do
  timeout 300 python bench.py --extras 0 --cpu-frames 0 > $O/new_1200_$rep.json 2>> $O/new.err
  SSF_URGENT_FIRST=0 timeout 300 python bench.py --extras 0 --cpu-frames 0 > $O/flat_1200_$rep.json 2>> $O/new.err
done
echo done
