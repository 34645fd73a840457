#!/bin/bash
# kernel times of ONE stereo pair with key-lines per step (images -> pose).  gpurun --timeout 300 -- 'bash tools/r06_pair_prof.sh'
R=$PWD; OUT=$R/gpurun_out/pair_prof; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pair_o
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pair_o -- python $R/tools/r06_images_leg.py 1 1 12 > $OUT/trace.out 2>&1
cd $R; grep stereo_pairs_per_s $OUT/trace.out | cut -c1-120; python tools/rocprof_summary.py stats $(find /tmp/pair_o -name "*.db" | head -1) > $OUT/kernel_stats.txt 2>&1; head -40 $OUT/kernel_stats.txt | cut -c1-150
