#!/bin/bash
R=$PWD; OUT=$R/gpurun_out/r06_ahead; mkdir -p $OUT
Q="--no-cpu-baseline --no-extras --no-clocks --repeats 3 --no-parity"
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
rm -rf /tmp/kt
STVO_LINES_AHEAD=$v timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/bench.py $Q > /dev/null 2>&1
echo "# LINES_AHEAD=$v"; python $R/tools/rocprof_summary.py stats $(find /tmp/kt -name "*.db" | head -1) | head -10 | cut -c1-120
python $R/tools/rocprof_summary.py timeline $(find /tmp/kt -name "*.db" | head -1) 10 -300 | cut -c1-130
done
