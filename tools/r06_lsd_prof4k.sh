#!/bin/bash
# kernel times of the LSD detector at 4096 images per launch (bench.py's leg).  gpurun --timeout 300 -- 'bash tools/r06_lsd_prof4k.sh'
R=$PWD; OUT=$R/gpurun_out/lsd_prof; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/lsd_o
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/lsd_o -- python $R/tools/r06_lsd_leg.py ${1:-4096} > $OUT/trace4k.out 2>&1
cd $R; tail -1 $OUT/trace4k.out | cut -c1-300; python tools/rocprof_summary.py stats $(find /tmp/lsd_o -name "*.db" | head -1) > $OUT/kernel_stats4k.txt 2>&1; head -24 $OUT/kernel_stats4k.txt | cut -c1-170
