#!/bin/bash
R=$PWD; OUT=$R/gpurun_out/r06; mkdir -p $OUT
PMCB="python $R/bench.py --no-cpu-baseline --no-extras --no-parity --no-clocks --steps 4 --warmup 1 --repeats 1"
cd /tmp && export TMPDIR=/tmp
for attempt in 1 2 3; do
rm -rf /tmp/pmc_x; S=$(date +%s)
STVO_LINES_AHEAD=0 timeout 60 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d /tmp/pmc_x -- $PMCB > /dev/null 2>&1; rc=$?; echo "wave cycles attempt $attempt: exit $rc $(( $(date +%s) - S )) s"
if [ $rc -eq 0 ]; then python $R/tools/rocprof_summary.py pmc $(find /tmp/pmc_x -name "*.db" | head -1) 2>/dev/null | grep -i "hamming_knn2_mfma_kernel\|pose2c\|grid_points_fused\|counter" > $OUT/pmc_wave_cycles.txt; break; fi
done
head -4 $OUT/pmc_wave_cycles.txt | cut -c1-150
