#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02f
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
for rep in 1 2 3; do
  ( cd tools/ab/r01 && timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 ) > $O/r01_s20_$rep.json 2>> $O/r01.err
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --extras 0 --cpu-frames 0 > $O/new_s20_$rep.json 2>> $O/new.err
  SSF_SEQ_RAMP=0 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --extras 0 --cpu-frames 0 > $O/new_s20_oldramp_$rep.json 2>> $O/new.err
done
for rep in 1 2; do
  ( cd tools/ab/r01 && timeout 300 python bench.py --cpu-frames 0 ) > $O/r01_1200_$rep.json 2>> $O/r01.err
  timeout 300 python bench.py --extras 0 --cpu-frames 0 > $O/new_1200_$rep.json 2>> $O/new.err
done
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err
echo done
