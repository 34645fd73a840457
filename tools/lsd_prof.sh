#!/bin/bash
# per-kernel times of the LSD detector on ONE KITTI-size image:  gpurun --timeout 200 -- 'bash tools/lsd_prof.sh'
R=$PWD; OUT=$R/gpurun_out/lsd_prof; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/lsd_o
timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/lsd_o -- python $R/tools/lsd_probe.py --batch ${1:-1} --iters 3 > $OUT/trace.out 2>&1
cd $R; tail -2 $OUT/trace.out; python tools/rocprof_summary.py stats $(find /tmp/lsd_o -name "*.db" | head -1) > $OUT/kernel_stats.txt 2>&1; head -16 $OUT/kernel_stats.txt | cut -c1-150
