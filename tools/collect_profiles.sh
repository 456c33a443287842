#!/bin/bash
# copy what one run of tools/gpu_run_g.sh left under gpurun_out/<dir> into profiles/ (round-2 names)
set -eu
O=gpurun_out/${1:?run directory under gpurun_out/}
P=profiles
cp $O/pmc_r02.json $P/pmc_r02.json
SHA=$(python -c "import json;print(json.load(open('$O/pmc_r02.json'))['source_sha'])")
{ echo "# rocprofv3 --kernel-trace --pmc <counters> -- python bench.py --cpu-frames 0 --profile-frames 0 --extras 0 --steps 96 --warmup 8"
  echo "# (pipeline_depth 2, extract_batch 8; two passes of 8 SQ/GRBM counters; mean per launch; extract kernels: launches over the full batch of 8 frames only)"
  echo "# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles; kernel sources $SHA"
  echo; echo "## pass 1: occupancy, waits, LDS"; cat $O/pmc_sq.txt; echo; echo "## pass 2: instruction mix"; cat $O/pmc_sq2.txt; } > $P/pmc_r02_occupancy.txt
cp $O/rocprof_summary.txt $P/rocprof_r02.txt
cp $O/rocprof_distribution.txt $P/rocprof_r02_distribution.txt
cp $O/pytest_gpu.log $P/gputest_r02_full_suite.log
cp $O/pass_probe.txt $P/pass_probe_r02.txt
for n in default driver_form latency_mode config3 config4_1rank config5 one_rank_rccl one_rank_p2p; do
  [ -s $O/bench_$n.json ] && tail -n 1 $O/bench_$n.json > $P/bench_r02_$n.json
done
[ -s $O/r01_1200.json ] && tail -n 1 $O/r01_1200.json > $P/bench_r02_samebox_round1_tree.json
[ -s $O/r01_s20.json ] && tail -n 1 $O/r01_s20.json > $P/bench_r02_samebox_round1_tree_driver_form.json
[ -s $O/r01_latency.json ] && tail -n 1 $O/r01_latency.json > $P/bench_r02_samebox_round1_tree_latency_mode.json
echo "profiles/ <- $O (kernel sources $SHA)"
