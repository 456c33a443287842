#!/bin/bash
# peer-to-peer backend after a change: its tests, the probe, one-rank benches
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r02n2}
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_p2p_gpu.py tests/test_comm_gpu.py -q -m gpu > $O/test_p2p.log 2>&1
echo "p2p tests rc=$?" >> $O/test_p2p.log
timeout 600 python tools/p2p_probe.py --threads --ranks 1 2 4 2 1 > $O/p2p_probe_threads.txt 2>&1
for rep in 1 2; do
  timeout 300 python bench.py --extras 0 --cpu-frames 0 > $O/default_$rep.json 2>> $O/bench.err
  timeout 300 python bench.py --extras 0 --cpu-frames 0 --force-sharded --comm p2p > $O/one_rank_p2p_$rep.json 2>> $O/bench.err
done
tail -n 5 $O/test_p2p.log; cat $O/p2p_probe_threads.txt
for f in $O/*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(d["value"],1), d["stage_ms"], d["config"]["exchange"])
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
