#!/bin/bash
# SQ counter passes for one kernel of the bench step (run on the GPU box through gpurun):
#   tools/pmc_kernel.sh gpurun_out/<dir> <kernel name substring> [bench args]
# One rocprofv3 --pmc run per counter group (--kernel-trace only), summarised by tools/rocprof_summary.py.
R=$PWD; OUT=$R/$1; K=$2; shift; shift; mkdir -p $OUT
BENCH="python $R/bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 $@"
: > $OUT/pmc.txt
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU"; do
  cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pmc_g
  timeout 120 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmc_g -- $BENCH > /dev/null 2>&1
  cd $R; python tools/rocprof_summary.py pmc $(find /tmp/pmc_g -name "*.db" | head -1) 2>/dev/null | grep -i "$K\|counter" >> $OUT/pmc.txt; rm -rf /tmp/pmc_g
done
cat $OUT/pmc.txt
