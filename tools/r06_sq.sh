#!/bin/bash
# SQ counter passes over the bench workload (final code; the streams from a cache written by an unprofiled run: no generator processes are
# forked under the profiler — that, not any kernel, is what made counter passes hang on this pool):  gpurun --timeout 600 -- 'bash tools/r06_sq.sh'
R=$PWD; OUT=$R/gpurun_out/r06; mkdir -p $OUT
PMCB="python $R/bench.py --no-cpu-baseline --no-extras --no-parity --no-clocks --steps 4 --warmup 1 --repeats 1 --streams-cache /tmp/streams3072.pkl"
cd /tmp && export TMPDIR=/tmp
[ -f /tmp/streams3072.pkl ] || $PMCB > /dev/null 2>&1
pass() {  # <file tag> <grep pattern> <counters...>
  local tag=$1 pat=$2; shift 2
  for attempt in 1 2; do
    rm -rf /tmp/pmc_x; local S=$(date +%s)
    timeout 60 rocprofv3 --pmc "$@" --kernel-trace -d /tmp/pmc_x -- $PMCB > /dev/null 2>&1; local rc=$?
    echo "$tag attempt $attempt: exit $rc, $(( $(date +%s) - S )) s"
    if [ $rc -eq 0 ]; then python $R/tools/rocprof_summary.py pmc $(find /tmp/pmc_x -name "*.db" | head -1) 2>/dev/null | grep -i "$pat" > $OUT/pmc_$tag.txt; return 0; fi
  done
}
pass sq "hamming_knn2_mfma\|pose\|grid_points_fused\|counter" SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES
pass wave_cycles "hamming_knn2_mfma_kernel\|pose2c\|grid_points_fused\|counter" SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS
head -5 $OUT/pmc_sq.txt | cut -c1-140; grep "pose2c" $OUT/pmc_wave_cycles.txt | cut -c1-120
