#!/bin/bash
# round 2, GPU call A: full GPU test suite, bench (1200-step and the driver's 20-step form), A/B of the relabelling
# tile width, rocprofv3 kernel trace + PMC passes (HBM traffic, occupancy / LDS conflicts).  Everything under gpurun_out/r02a.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02a
mkdir -p $O
cd $R
nproc > $O/nproc.txt; lscpu | head -20 >> $O/nproc.txt
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 ) > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 600 python __graft_entry__.py smoke > $O/smoke.log 2>&1
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --extras 0 --cpu-frames 0 > $O/bench_s20_a.json 2> $O/bench_s20_a.err
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --extras 0 --cpu-frames 0 > $O/bench_s20_b.json 2>> $O/bench_s20_a.err
for npx in 1 2; do
  SSF_PASS_NPX=$npx timeout 300 python bench.py --extras 0 --cpu-frames 0 > $O/bench_npx${npx}.json 2> $O/bench_npx${npx}.err
  SSF_PASS_NPX=$npx timeout 300 python bench.py --extras 0 --cpu-frames 0 --pipeline-depth 0 --extract-batch 1 --steps 200 > $O/bench_npx${npx}_latency.json 2>> $O/bench_npx${npx}.err
done
timeout 300 python bench.py --config 3 --extras 0 --cpu-frames 0 > $O/bench_config3.json 2> $O/bench_config3.err
cd /tmp && export TMPDIR=/tmp
PROF="python $R/bench.py --cpu-frames 0 --profile-frames 0 --extras 0 --steps 96 --warmup 8"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- $PROF > $O/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o p -- $PROF > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o p -- $PROF > $O/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/pmc_sq -o p -- $PROF > $O/pmc_sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $O/pmc_sq2 -o p -- $PROF > $O/pmc_sq2.log 2>&1
cd $R
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_r02.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of: python bench.py --cpu-frames 0 --profile-frames 0 --extras 0 --steps 96 --warmup 8" > $O/pmc_summary.txt 2>&1
python tools/pmc_counters.py $O/pmc_sq > $O/pmc_sq.txt 2>&1
python tools/pmc_counters.py $O/pmc_sq2 > $O/pmc_sq2.txt 2>&1
DB=$(find $O/trace -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB $O/rocprof_r02a.txt "bench.py --steps 96, pipelined 2 x 8" > /dev/null 2>&1
[ -n "$DB" ] && python tools/rocprof_dist.py $DB > $O/rocprof_r02a_distribution.txt 2>&1
# keep the merge-back small: the raw traces are large
find $O -name "*.db" -size +20M -delete; find $O -name "*.csv" -size +20M -delete
du -sh $O > $O/size.txt
echo done
