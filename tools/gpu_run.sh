#!/bin/bash
# ONE parameterised script for everything that is run on the GPU box:
#   gpurun --timeout N -- 'bash tools/gpu_run.sh <outdir> <step> [<step> ...]'
# steps (each writes under gpurun_out/<outdir>/):
#   suite        pytest -m gpu (full), smoke
#   quick        pytest -m gpu on tests/test_parity_gpu.py only
#   bench        default bench line (1200 frames) + the driver's form (--steps 20 --warmup 5), twice
#   latency      one frame in flight (pipeline_depth 0, extract_batch 1)
#   config3|4|5  bench.py --config N
#   trace        rocprofv3 --kernel-trace --stats of the profile command
#   pmc          FETCH_SIZE / WRITE_SIZE passes (separate) + the two SQ passes, summarised
#   prof3        the profile command of the trace / pmc steps that follow becomes BASELINE config 3 (1280x960, 1 M rows in view,
#                10 forced ICP iterations; outputs tagged _config3)
#   prof5 / prof2  the same for BASELINE config 5 / back to the default workload
#   driver       the driver's exact command (python bench.py --gpus 1 --steps 20 --warmup 5, everything on)
#   gpus2        python bench.py --gpus 2 ... on this one-GPU box: must end in the JSON error record
#   sharded1     every multi-GPU exchange on one rank (RCCL, then the peer-to-peer regions)
#   env:K=V      export K=V for the steps that follow (A/B of kernel variants on the same box)
#   py:<file>    python tools/<file> (a probe), output to <file>.txt
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:?outdir}; shift
mkdir -p $O
cd $R
TAG=""
PROF="python $R/bench.py --cpu-frames 0 --profile-frames 0 --extras 0 --steps 96 --warmup 8"
PROFNOTE="bench.py --cpu-frames 0 --profile-frames 0 --extras 0 --steps 96 --warmup 8 (pipelined 2 x 8)"
lastline() { [ -s "$1" ] && tail -n 1 "$1" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$2', round(d['value'],1), d.get('unit'), 'frac', round(d['roofline']['frac'],4), d['roofline'].get('kernel'), 'seq_ms', d.get('sequential_ms_per_frame'))" 2>/dev/null >> $O/summary.txt; }
for step in "$@"; do
  case $step in
    prof3) PROF="python $R/bench.py --config 3 --cpu-frames 0 --profile-frames 0 --extras 0 --steps 32 --warmup 8"; PROFNOTE="bench.py --config 3 --cpu-frames 0 --profile-frames 0 --extras 0 --steps 32 --warmup 8 (1280x960, pipelined 2 x 4)"; export PMC_EXTRACT_BATCH=4; export PROFPIX=1228800; TAG="_config3";;
    prof5) PROF="python $R/bench.py --config 5 --cpu-frames 0 --profile-frames 0 --extras 0 --steps 96 --warmup 8"; PROFNOTE="bench.py --config 5 --cpu-frames 0 --profile-frames 0 --extras 0 --steps 96 --warmup 8 (TUM-shaped frames, pre-filter in the frame, pipelined 2 x 12)"; export PMC_EXTRACT_BATCH=12; export PROFPIX=307200; TAG="_config5";;
    prof2) PROF="python $R/bench.py --cpu-frames 0 --profile-frames 0 --extras 0 --steps 96 --warmup 8"; PROFNOTE="bench.py --cpu-frames 0 --profile-frames 0 --extras 0 --steps 96 --warmup 8 (pipelined 2 x 8)"; export PMC_EXTRACT_BATCH=8; export PROFPIX=307200; TAG="";;
    driver)
      timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver_command.err; lastline $O/bench_driver_command.json driver_command;;
    gpus2)
      timeout 300 python bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_gpus2.json 2> $O/bench_gpus2.err; echo "gpus2 rc=$? $(cat $O/bench_gpus2.json)" >> $O/summary.txt;;
    sharded1)
      timeout 400 python bench.py --force-sharded --extras 0 --cpu-frames 0 --steps 240 > $O/bench_one_rank_rccl.json 2> $O/bench_one_rank_rccl.err; lastline $O/bench_one_rank_rccl.json one_rank_rccl
      timeout 400 python bench.py --force-sharded --comm p2p --extras 0 --cpu-frames 0 --steps 240 > $O/bench_one_rank_p2p.json 2> $O/bench_one_rank_p2p.err; lastline $O/bench_one_rank_p2p.json one_rank_p2p
      timeout 400 python bench.py --force-sharded --extract dealt --extras 0 --cpu-frames 0 --steps 240 > $O/bench_one_rank_rccl_dealt.json 2> $O/bench_one_rank_rccl_dealt.err; lastline $O/bench_one_rank_rccl_dealt.json one_rank_rccl_dealt;;
    env:*) export "${step#env:}"; case "${step#env:}" in SSF_*) export SSF_PRODUCT_VARIANT=${SSF_PRODUCT_VARIANT:-lab};; esac;    # (the switches live in the lab build)
            TAG="${TAG}_$(echo ${step#env:} | tr -c 'A-Za-z0-9=\n' '_')";;
    suite)
      ( time timeout 2400 python -m pytest tests -m gpu -x -q --durations=15 ) > $O/pytest_gpu$TAG.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu$TAG.log
      timeout 600 python __graft_entry__.py smoke > $O/smoke$TAG.log 2>&1; tail -n 3 $O/pytest_gpu$TAG.log >> $O/summary.txt;;
    quick)
      ( time timeout 1200 python -m pytest tests/test_parity_gpu.py -m gpu -x -q ) > $O/pytest_quick$TAG.log 2>&1; tail -n 3 $O/pytest_quick$TAG.log >> $O/summary.txt;;
    bench)
      timeout 600 python bench.py > $O/bench_default$TAG.json 2> $O/bench_default$TAG.err; lastline $O/bench_default$TAG.json default$TAG
      for r in a b; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --extras 0 --cpu-frames 0 > $O/bench_s20_$r$TAG.json 2> $O/bench_s20_$r$TAG.err; lastline $O/bench_s20_$r$TAG.json s20_$r$TAG; done;;
    fast)
      timeout 300 python bench.py --extras 0 --cpu-frames 0 > $O/bench_fast$TAG.json 2> $O/bench_fast$TAG.err; lastline $O/bench_fast$TAG.json fast$TAG
      timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --extras 0 --cpu-frames 0 > $O/bench_s20$TAG.json 2> $O/bench_s20$TAG.err; lastline $O/bench_s20$TAG.json s20$TAG;;
    latency)
      timeout 300 python bench.py --extras 0 --cpu-frames 0 --pipeline-depth 0 --extract-batch 1 --steps 200 > $O/bench_latency$TAG.json 2> $O/bench_latency$TAG.err; lastline $O/bench_latency$TAG.json latency$TAG;;
    config3|config4|config5)
      timeout 400 python bench.py --config ${step#config} --extras 0 --cpu-frames 0 > $O/bench_$step$TAG.json 2> $O/bench_$step$TAG.err; lastline $O/bench_$step$TAG.json $step$TAG;;
    trace)
      ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace$TAG -o trace -- $PROF > $O/trace$TAG.log 2>&1 )
      DB=$(find $O/trace$TAG -name "*.db" | head -1)
      [ -n "$DB" ] && python tools/rocprof_summary.py $DB $O/rocprof_summary$TAG.txt "$PROFNOTE" $O/rocprof$TAG.json ${PROFPIX:-307200} > /dev/null 2>&1
      [ -n "$DB" ] && python tools/rocprof_dist.py $DB > $O/rocprof_distribution$TAG.txt 2>&1;;
    pmc)
      ( cd /tmp && export TMPDIR=/tmp
        timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch$TAG -o p -- $PROF > $O/pmc_fetch$TAG.log 2>&1
        timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write$TAG -o p -- $PROF > $O/pmc_write$TAG.log 2>&1
        timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/pmc_sq$TAG -o p -- $PROF > $O/pmc_sq$TAG.log 2>&1
        timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $O/pmc_sq2$TAG -o p -- $PROF > $O/pmc_sq2$TAG.log 2>&1 )
      python tools/pmc_summary.py $O/pmc_fetch$TAG $O/pmc_write$TAG $O/pmc$TAG.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of: $PROF" > $O/pmc_summary$TAG.txt 2>&1
      python tools/pmc_counters.py $O/pmc_sq$TAG > $O/pmc_sq$TAG.txt 2>&1
      python tools/pmc_counters.py $O/pmc_sq2$TAG > $O/pmc_sq2$TAG.txt 2>&1
      DBT=$(find $O/trace$TAG -name "*.db" 2>/dev/null | head -1)         # (the `trace` step before `pmc`: durations come from the unperturbed run)
      [ -n "$DBT" ] && python tools/pmc_issue.py $O/pmc_sq2$TAG $DBT $O/pmc_issue$TAG.json "$PROFNOTE" > $O/pmc_issue$TAG.txt 2>&1;;
    py:*)
      f=${step#py:}; timeout 900 python tools/$f > $O/${f%.py}$TAG.txt 2>&1;;
    sh:*)
      timeout 1800 bash -c "${step#sh:}" > $O/sh$TAG.txt 2>&1;;
    *) echo "unknown step $step" >> $O/summary.txt;;
  esac
done
# keep the merge-back small: the raw traces are large
find $O -name "*.db" -size +20M -delete; find $O -name "*.csv" -size +20M -delete
cat $O/summary.txt 2>/dev/null
echo done
