R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r15c; mkdir -p $O; cd $R
for r in 1 2; do for cfg in "2 8" "3 8" "3 6" "3 5" "4 4"; do set -- $cfg
  v20=$(timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --extras 0 --cpu-frames 0 --pipeline-depth $1 --extract-batch $2 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['pipeline_fill']['frame_done_us'][0], [b['frames'] for b in d['pipeline_fill']['extract_batches_launched']])")
  v12=$(timeout 300 python bench.py --extras 0 --cpu-frames 0 --profile-frames 0 --pipeline-depth $1 --extract-batch $2 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1))")
  echo "depth $1 batch $2 run $r: 20-frame form $v20 | 1200 frames $v12" >> $O/depth_ab.txt
done; done
cat $O/depth_ab.txt
