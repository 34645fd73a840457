#!/bin/bash
# Round 6, first GPU call: (1) the new bench-line / 8-rank tests and the matcher tests (QB = 1 fence), (2) the default `python bench.py` exactly as the
# driver runs it (short line + bench_extras.json), (3) rocprofv3 --kernel-trace --stats of the bench without extras: does the light pass's avg_launch_ms
# agree with the tracer's view of the TIMED region?
#   gpurun --timeout 900 -- 'bash tools/r06_first_call.sh'
R=$PWD; OUT=$R/gpurun_out/r06_first; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_shard.py tests/test_gpu_match.py -x -q > $OUT/tests.txt 2>&1; echo "tests rc $?"; tail -3 $OUT/tests.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc $?"
wc -c $OUT/bench_default.json; cat $OUT/bench_default.json; cp bench_extras.json $OUT/bench_extras_default.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/bench.py --no-cpu-baseline --no-extras > $OUT/bench_profiled.json 2>/dev/null
DB=$(find /tmp/kt -name "*.db" | head -1)
cd $R
{ echo "# bench line of the profiled run:"; cat $OUT/bench_profiled.json; echo; python tools/rocprof_summary.py stats $DB; echo; python tools/rocprof_summary.py split $DB; } > $OUT/kernel_stats.txt
{ python tools/rocprof_summary.py timeline $DB 48 -400; echo; echo "# light pass + parity (end of run)"; python tools/rocprof_summary.py timeline $DB 60; } > $OUT/timeline.txt
head -30 $OUT/kernel_stats.txt | cut -c1-200
