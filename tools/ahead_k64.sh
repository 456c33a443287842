#!/bin/bash
# 64 timed frames (the longest run whose per-frame marks the library keeps) with and without the first ICP iteration fused into the
# row-move kernel (lab build: SSF_ICP_AHEAD=1|0), alternated: where in a sequence does which form win?
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:?outdir}; mkdir -p $O; cd $R
export SSF_PRODUCT_VARIANT=lab
for r in 1 2; do for m in 1 0; do
  SSF_ICP_AHEAD=$m timeout 300 python bench.py --gpus 1 --steps 64 --warmup 5 --extras 0 --cpu-frames 0 > $O/k64_ahead${m}_$r.json 2> $O/k64_ahead${m}_$r.err
done; done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/k64_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); pf=d["pipeline_fill"]; fd=pf["frame_done_us"]
    inc=[round(fd[0])]+[round(fd[i]-fd[i-1]) for i in range(1,len(fd))]
    te=pf["track_entered_us"]; fr=pf["first_icp_record_us"]; idn=pf["icp_done_us"]
    print(f.split("/")[-1], round(d["value"]), "region", pf["region_us"])
    print("  inc", inc)
    print("  entry->first", [round(fr[i]-te[i]) for i in range(len(fd))])
    print("  first->icpdone", [round(idn[i]-fr[i]) for i in range(len(fd))])
    print("  icpdone->done", [round(fd[i]-idn[i]) for i in range(len(fd))])
    print("  iters", pf["icp_iters"]); print("  batches", [(round(b["at_us"]),b["frames"],b["host_us"]) for b in pf["extract_batches_launched"]])
PY
