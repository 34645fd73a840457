#!/bin/bash
# kernel times of the images -> poses leg with key-lines.  gpurun --timeout 400 -- 'bash tools/r06_images_prof.sh [streams]'
R=$PWD; OUT=$R/gpurun_out/img_prof; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/img_o
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/img_o -- python $R/tools/r06_images_leg.py ${1:-2048} 1 3 > $OUT/trace.out 2>&1
cd $R; tail -1 $OUT/trace.out | cut -c1-400; python tools/rocprof_summary.py stats $(find /tmp/img_o -name "*.db" | head -1) > $OUT/kernel_stats.txt 2>&1; head -34 $OUT/kernel_stats.txt | cut -c1-150
