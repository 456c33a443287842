#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r02d}
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 300 python tools/pass_probe.py > $O/pass_probe.txt 2>&1
for rep in 1 2; do
  ( cd tools/ab/r01 && timeout 300 python bench.py --cpu-frames 0 ) > $O/r01_1200_$rep.json 2>> $O/r01.err
  timeout 300 python bench.py --extras 0 --cpu-frames 0 > $O/new_1200_$rep.json 2>> $O/new.err
  ( cd tools/ab/r01 && timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 ) > $O/r01_s20_$rep.json 2>> $O/r01.err
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --extras 0 --cpu-frames 0 > $O/new_s20_$rep.json 2>> $O/new.err
done
timeout 300 python bench.py --extras 0 --cpu-frames 0 --pipeline-depth 0 --extract-batch 1 --steps 200 > $O/new_latency.json 2>> $O/new.err
timeout 300 python bench.py --config 3 --extras 0 --cpu-frames 0 > $O/new_config3.json 2>> $O/new.err
echo done
