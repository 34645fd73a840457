#!/bin/bash
# Instruction-fetch counters of the pose kernels over the bench workload (one pass per group):  tools/pmc_icache.sh <tag>
R=$PWD; T=$1; OUT=$R/gpurun_out/$T; mkdir -p $OUT; : > $OUT/pmc_icache.txt
PMCB="python $R/bench.py --no-cpu-baseline --no-extras --no-parity --no-clocks --steps 1 --warmup 1 --repeats 1"
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_IFETCH" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM"; do
  cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pmc_g
  timeout 300 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmc_g -- $PMCB > /dev/null 2>$OUT/pmc_icache.err; echo "$grp: exit $?" >> $OUT/pmc_icache.txt
  cd $R; python tools/rocprof_summary.py pmc $(find /tmp/pmc_g -name "*.db" | head -1) 2>/dev/null | grep -i "pose\|grid_points_fused\|counter" >> $OUT/pmc_icache.txt
done
cut -c1-200 $OUT/pmc_icache.txt
