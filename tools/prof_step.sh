#!/bin/bash
# Kernel-level profile of the bench step (run on the GPU box through gpurun):  tools/prof_step.sh gpurun_out/<dir> [bench args]
R=$PWD; OUT=$R/$1; shift; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/bench.py --no-cpu-baseline --no-extras "$@" > $OUT/bench_profiled.json 2>/dev/null
cd $R; python tools/rocprof_summary.py stats $(find /tmp/kt -name "*.db" | head -1) > $OUT/kernel_stats.txt; rm -rf /tmp/kt
head -40 $OUT/kernel_stats.txt
