#!/bin/bash
# kernel timeline of the last steps of the device pipeline (both streams):  tools/timeline.sh <tag> [N] [extra bench_pipeline args]
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; N=${2:-70}
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/tl
timeout 300 rocprofv3 --kernel-trace -d /tmp/tl -- python $R/tools/bench_pipeline.py --batch 1024 --steps 6 --warmup 4 --cpu-frames 0 ${@:3} > /dev/null 2>&1
cd $R; python tools/rocprof_summary.py timeline $(find /tmp/tl -name "*.db" | head -1) $N > $OUT/timeline_$1.txt; rm -rf /tmp/tl
cat $OUT/timeline_$1.txt
