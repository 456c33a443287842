#!/bin/bash
# round 2, GPU call C: regression check against the round-1 tree after the leaver fix, occupancy A/B of the RGB-D pass,
# ablation timing of the pass kernel.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02c
mkdir -p $O
cd $R
timeout 300 python tools/pass_probe.py > $O/pass_probe_w8.txt 2>&1
SSF_PASS_WAVES=6 timeout 300 python tools/pass_probe.py > $O/pass_probe_w6.txt 2>&1
for rep in 1 2; do
  ( cd tools/ab/r01 && timeout 300 python bench.py --cpu-frames 0 ) > $O/r01_1200_$rep.json 2>> $O/r01.err
  SSF_PASS_WAVES=6 timeout 300 python bench.py --extras 0 --cpu-frames 0 > $O/new_1200_w6_$rep.json 2>> $O/new.err
  SSF_PASS_WAVES=8 timeout 300 python bench.py --extras 0 --cpu-frames 0 > $O/new_1200_w8_$rep.json 2>> $O/new.err
  ( cd tools/ab/r01 && timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 ) > $O/r01_s20_$rep.json 2>> $O/r01.err
  SSF_PASS_WAVES=6 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --extras 0 --cpu-frames 0 > $O/new_s20_w6_$rep.json 2>> $O/new.err
  SSF_PASS_WAVES=8 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --extras 0 --cpu-frames 0 > $O/new_s20_w8_$rep.json 2>> $O/new.err
done
timeout 300 python bench.py --extras 0 --cpu-frames 0 --pipeline-depth 0 --extract-batch 1 --steps 200 > $O/new_latency.json 2>> $O/new.err
echo done
