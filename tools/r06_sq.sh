#!/bin/bash
# SQ counter passes over the bench workload; the key-line stage ahead off (its gate kernel waits for another stream's kernel: under --pmc every dispatch runs alone)
R=$PWD; OUT=$R/gpurun_out/r06; mkdir -p $OUT
PMCB="python $R/bench.py --no-cpu-baseline --no-extras --no-parity --no-clocks --steps 4 --warmup 1 --repeats 1"
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
rm -rf /tmp/pmc_x; S=$(date +%s)
STVO_LINES_AHEAD=$v timeout 90 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace -d /tmp/pmc_x -- $PMCB > /dev/null 2>&1; echo "LINES_AHEAD=$v sq: exit $? $(( $(date +%s) - S )) s"
if [ $v = 0 ]; then python $R/tools/rocprof_summary.py pmc $(find /tmp/pmc_x -name "*.db" | head -1) 2>/dev/null | grep -i "hamming_knn2_mfma\|pose\|grid_points_fused\|counter" > $OUT/pmc_sq.txt; fi
done
rm -rf /tmp/pmc_x; S=$(date +%s)
STVO_LINES_AHEAD=0 timeout 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d /tmp/pmc_x -- $PMCB > /dev/null 2>&1; echo "wave cycles: exit $? $(( $(date +%s) - S )) s"
python $R/tools/rocprof_summary.py pmc $(find /tmp/pmc_x -name "*.db" | head -1) 2>/dev/null | grep -i "hamming_knn2_mfma_kernel\|pose2c\|grid_points_fused\|counter" > $OUT/pmc_wave_cycles.txt
head -6 $OUT/pmc_sq.txt | cut -c1-150
