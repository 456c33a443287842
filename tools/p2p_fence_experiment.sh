#!/bin/bash
# Round 3 experiment: does real agent-scope release / acquire ordering around the arrival ticket
# (the SSF_ARRIVE_FENCED build: tools/build_variant.sh fenced -DSSF_EXPERIMENTS -DSSF_ARRIVE_FENCED) remove the first-frame divergence that
# four ranks of one map on one GPU showed with UNCACHED exchange regions in round 2?  Processes of the two arms interleaved.
#   bash tools/p2p_fence_experiment.sh <processes per arm> <cycles per process>
N=${1:-10}; C=${2:-100}
cd "$(dirname "$0")/.."
for i in $(seq 1 $N); do
  for arm in relaxed fenced; do
    if [ $arm = fenced ]; then export SSF_PRODUCT_VARIANT=fenced; else export SSF_PRODUCT_VARIANT=lab; fi     # (SSF_P2P_REGION_UNCACHED is a lab switch)
    out=$(SSF_P2P_REGION_UNCACHED=1 timeout 600 python tools/p2p_first_frame_stress.py --ranks 4 --frames 3 --cycles $C 2>/dev/null | tail -n 1 | cut -c1-400)
    echo "uncached regions, $arm arrival : $out"
  done
done
unset SSF_PRODUCT_VARIANT
for i in 1 2; do
  out=$(timeout 600 python tools/p2p_first_frame_stress.py --ranks 4 --frames 3 --cycles $C 2>/dev/null | tail -n 1 | cut -c1-200)
  echo "plain regions (default), relaxed arrival : $out"
done
