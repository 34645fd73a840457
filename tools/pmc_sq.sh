#!/bin/bash
# SQ counters of the matching kernels (default bench = stream order, 3 steps), one rocprofv3 --pmc pass per counter group.
#   tools/pmc_sq.sh gpurun_out/<dir>
R=$PWD; OUT=$R/$1; mkdir -p $OUT; : > $OUT/pmc_sq.txt
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS" "SQ_WAVES SQ_INSTS_SALU SQ_ACTIVE_INST_ANY"; do
  cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pmc_g
  timeout 300 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmc_g -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  cd $R; python tools/rocprof_summary.py pmc $(find /tmp/pmc_g -name "*.db" | head -1) | grep -i "hamming\|counter" >> $OUT/pmc_sq.txt
done
cat $OUT/pmc_sq.txt
