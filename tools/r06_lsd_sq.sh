#!/bin/bash
# SQ counters of the one-wave-per-image growth kernel at 1024 images (is the batch issue-bound?):  gpurun --timeout 600 -- 'bash tools/r06_lsd_sq.sh'
R=$PWD; OUT=$R/gpurun_out/lsd_sq; mkdir -p $OUT; : > $OUT/pmc_sq.txt
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY" "SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_WR SQ_WAVES" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
  cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pmc_g
  timeout 120 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmc_g -- python $R/tools/lsd_probe.py --batch ${1:-1024} --iters 1 > /dev/null 2>&1
  cd $R; python tools/rocprof_summary.py pmc $(find /tmp/pmc_g -name "*.db" | head -1) 2>&1 | grep -i "lsd_grow\|counter" >> $OUT/pmc_sq.txt
done
cat $OUT/pmc_sq.txt
