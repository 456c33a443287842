#!/bin/bash
# round 2, GPU call B: same-box A/B against the round-1 tree, host topology, PMC instruction / occupancy counters, the
# new pre-filter kernel, the OpenMP CPU baseline probe.  Everything under gpurun_out/r02b.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02b
mkdir -p $O
cd $R
( lscpu | grep -E "Model name|Socket|NUMA|Thread|Core" ; numactl -H 2>/dev/null | head -12; for d in /sys/class/drm/card*/device; do echo $d $(cat $d/numa_node 2>/dev/null) $(cat $d/local_cpulist 2>/dev/null); done; rocm-smi --showtoponuma 2>/dev/null | head -20 ) > $O/topology.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q -k "prefilter or replay or config5 or sequence_bit_exact" > $O/pytest_sel.log 2>&1
echo "pytest rc=$?" >> $O/pytest_sel.log
for rep in 1 2; do
  ( cd tools/ab/r01 && timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 ) > $O/r01_s20_$rep.json 2>> $O/r01.err
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --extras 0 --cpu-frames 0 --pin 0 > $O/new_s20_nopin_$rep.json 2>> $O/new.err
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --extras 0 --cpu-frames 0 > $O/new_s20_pin_$rep.json 2>> $O/new.err
  ( cd tools/ab/r01 && timeout 300 python bench.py --cpu-frames 0 ) > $O/r01_1200_$rep.json 2>> $O/r01.err
  timeout 300 python bench.py --extras 0 --cpu-frames 0 --pin 0 > $O/new_1200_nopin_$rep.json 2>> $O/new.err
  timeout 300 python bench.py --extras 0 --cpu-frames 0 > $O/new_1200_pin_$rep.json 2>> $O/new.err
done
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
cd /tmp && export TMPDIR=/tmp
PROF="python $R/bench.py --cpu-frames 0 --profile-frames 0 --extras 0 --steps 96 --warmup 8"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/pmc_sq -o p -- $PROF > $O/pmc_sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU -d $O/pmc_sq2 -o p -- $PROF > $O/pmc_sq2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA -d $O/pmc_sq3 -o p -- $PROF > $O/pmc_sq3.log 2>&1
cd $R
for d in pmc_sq pmc_sq2 pmc_sq3; do python tools/pmc_counters.py $O/$d > $O/$d.txt 2>&1; done
find $O -name "*.db" -delete; find $O -name "*.csv" -size +5M -delete
echo done
