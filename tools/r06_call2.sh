#!/bin/bash
# Round 6, second call: the light pass after the marker reorder — (1) shard tests, (2) rocprofv3 kernel trace of the bench without extras, with a timeline of
# the timed region AND of the light pass (do the three kernels have the same neighbours and durations in both?)
R=$PWD; OUT=$R/gpurun_out/r06_call2; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_shard.py -x -q > $OUT/tests.txt 2>&1; echo "tests rc $?"; tail -3 $OUT/tests.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/bench.py --no-cpu-baseline --no-extras --no-parity --no-clocks > $OUT/bench_profiled.json 2>/dev/null
DB=$(find /tmp/kt -name "*.db" | head -1)
cd $R
cat $OUT/bench_profiled.json | cut -c1-3000
{ python tools/rocprof_summary.py stats $DB; } > $OUT/kernel_stats.txt
python tools/rocprof_summary.py timeline $DB 700 > $OUT/timeline_all.txt
head -14 $OUT/kernel_stats.txt | cut -c1-150
