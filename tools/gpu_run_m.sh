#!/bin/bash
# peer-to-peer exchange backend: tests, one-rank benches next to RCCL, and that the default path did not move
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r02m}
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_p2p_gpu.py -x -q -m gpu > $O/test_p2p.log 2>&1
echo "p2p tests rc=$?" >> $O/test_p2p.log
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_comm_gpu.py -x -q -m gpu > $O/test_parity.log 2>&1
echo "parity tests rc=$?" >> $O/test_parity.log
for rep in 1 2; do
  timeout 300 python bench.py --extras 0 --cpu-frames 0 > $O/default_$rep.json 2>> $O/bench.err
  timeout 300 python bench.py --extras 0 --cpu-frames 0 --force-sharded --comm rccl > $O/one_rank_rccl_$rep.json 2>> $O/bench.err
  timeout 300 python bench.py --extras 0 --cpu-frames 0 --force-sharded --comm p2p > $O/one_rank_p2p_$rep.json 2>> $O/bench.err
done
tail -5 $O/test_p2p.log $O/test_parity.log
for f in $O/*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(d["value"],1), d["stage_ms"], d["config"]["exchange"])
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
