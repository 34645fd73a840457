#!/bin/bash
# per-kernel times of the f2f match on clustered descriptors (512 frame pairs):  gpurun --timeout 200 -- 'bash tools/corr_prof.sh'
R=$PWD; OUT=$R/gpurun_out/corr; mkdir -p $OUT
timeout 120 python tools/corr_one.py 512 > $OUT/plain.txt 2>&1; cat $OUT/plain.txt
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/corr_o
timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/corr_o -- python $R/tools/corr_one.py 512 > $OUT/trace.err 2>&1
cd $R; python tools/rocprof_summary.py stats $(find /tmp/corr_o -name "*.db" | head -1) > $OUT/kernel_stats.txt 2>&1; head -30 $OUT/kernel_stats.txt | cut -c1-200
