#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r02l}
mkdir -p $O
cd $R
for rep in 1 2 3; do
  ( cd tools/ab/r01 && timeout 300 python bench.py --cpu-frames 0 ) > $O/r01_1200_$rep.json 2>> $O/r01.err
  timeout 300 python bench.py --extras 0 --cpu-frames 0 > $O/new_1200_$rep.json 2>> $O/new.err
  SSF_URGENT_FIRST=0 timeout 300 python bench.py --extras 0 --cpu-frames 0 > $O/flat_1200_$rep.json 2>> $O/new.err
  ( cd tools/ab/r01 && timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 ) > $O/r01_s20_$rep.json 2>> $O/r01.err
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --extras 0 --cpu-frames 0 > $O/new_s20_$rep.json 2>> $O/new.err
  SSF_URGENT_FIRST=0 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --extras 0 --cpu-frames 0 > $O/flat_s20_$rep.json 2>> $O/new.err
done
echo done
