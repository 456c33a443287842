#!/bin/bash
# round 2, GPU call G: the complete GPU suite, the bench lines and the profiles that get committed under profiles/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r02g}
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 ) > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err
timeout 300 python bench.py --extras 0 --cpu-frames 0 --pipeline-depth 0 --extract-batch 1 --steps 200 > $O/bench_latency_mode.json 2> $O/bench_latency_mode.err
timeout 300 python bench.py --config 3 --extras 0 > $O/bench_config3.json 2> $O/bench_config3.err
timeout 300 python bench.py --config 4 --extras 0 --cpu-frames 0 > $O/bench_config4_1rank.json 2> $O/bench_config4.err
timeout 300 python bench.py --config 5 --extras 0 --cpu-frames 0 > $O/bench_config5.json 2> $O/bench_config5.err
timeout 300 python bench.py --force-sharded --extras 0 --cpu-frames 0 > $O/bench_one_rank_rccl.json 2> $O/bench_one_rank_rccl.err
timeout 300 python bench.py --force-sharded --comm p2p --extras 0 --cpu-frames 0 > $O/bench_one_rank_p2p.json 2> $O/bench_one_rank_p2p.err
( cd tools/ab/r01 && timeout 300 python bench.py --cpu-frames 0 ) > $O/r01_1200.json 2>> $O/r01.err
( cd tools/ab/r01 && timeout 300 python bench.py --cpu-frames 0 --pipeline-depth 0 --extract-batch 1 --steps 200 ) > $O/r01_latency.json 2>> $O/r01.err
( cd tools/ab/r01 && timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 ) > $O/r01_s20.json 2>> $O/r01.err
timeout 300 python tools/pass_probe.py > $O/pass_probe.txt 2>&1
cd /tmp && export TMPDIR=/tmp
PROF="python $R/bench.py --cpu-frames 0 --profile-frames 0 --extras 0 --steps 96 --warmup 8"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- $PROF > $O/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o p -- $PROF > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o p -- $PROF > $O/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/pmc_sq -o p -- $PROF > $O/pmc_sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU -d $O/pmc_sq2 -o p -- $PROF > $O/pmc_sq2.log 2>&1
cd $R
PMC_EXTRACT_BATCH=8 python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_r02.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of: python bench.py --cpu-frames 0 --profile-frames 0 --extras 0 --steps 96 --warmup 8 (pipeline_depth 2, extract_batch 8; extract kernels: only the launches over the full batch of 8 frames are averaged). Counters are KB; hbm_bytes_per_launch = 1024 x (2 x FETCH_SIZE + WRITE_SIZE): FETCH_SIZE reports half of a wide coalesced read on gfx950 (MI355X_MICROARCH.md, HBM section)" > $O/pmc_summary.txt 2>&1
python tools/pmc_counters.py $O/pmc_sq > $O/pmc_sq.txt 2>&1
python tools/pmc_counters.py $O/pmc_sq2 > $O/pmc_sq2.txt 2>&1
DB=$(find $O/trace -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB $O/rocprof_summary.txt "bench.py --cpu-frames 0 --profile-frames 0 --extras 0 --steps 96 --warmup 8 (pipeline_depth 2, extract_batch 8)" > /dev/null 2>&1
[ -n "$DB" ] && python tools/rocprof_dist.py $DB > $O/rocprof_distribution.txt 2>&1
find $O -name "*.db" -delete; find $O -name "*.csv" -size +5M -delete
echo done
