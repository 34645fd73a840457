#!/bin/bash
# Round-4 evidence, run on the GPU box through gpurun:  tools/collect_r04.sh <tag>
#   1. bench line of the default command (bench_default.json);
#   2. rocprofv3 --kernel-trace --stats of `bench.py --no-cpu-baseline --no-extras`, summarised per kernel and per launch shape;
#   3. the timeline of the last steps of the same workload on both streams;
#   4. PMC passes over the BENCH WORKLOAD ITSELF (three cameras, four resident slots): FETCH_SIZE, WRITE_SIZE (separate passes, as
#      the HBM section of MI355X_MICROARCH.md prescribes) and the matrix-pipe / CU busy cycles — `bench.py --no-extras --no-cpu-baseline
#      --no-parity --steps 1 --warmup 1 --repeats 1` keeps the counter passes to a few hundred dispatches;
#   5. single-stream latency: handler (both modes) vs stvo_seq_push, and the kernel chain of one frame (seq_push and handler).
R=$PWD; T=$1; OUT=$R/gpurun_out/$T; mkdir -p $OUT
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --no-cpu-baseline --no-extras"
rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -- $BENCH > $OUT/bench_profiled.json 2>/dev/null
DB=$(find /tmp/kt -name "*.db" | head -1)
cd $R
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-extras (tools/collect_r04.sh); bench line of the profiled run:"; cat $OUT/bench_profiled.json; echo;
  python tools/rocprof_summary.py stats $DB; echo; echo "# the same dispatches split by launch shape (grid_x in work-items x workgroup size): one line per problem size of a kernel";
  python tools/rocprof_summary.py split $DB; } > $OUT/kernel_stats.txt
{ echo "# The last steps of 'python bench.py --no-cpu-baseline --no-extras' under rocprofv3 --kernel-trace (tools/collect_r04.sh): every dispatch in start order,";
  echo "# queue = HIP stream (the point stream and the key-line stream forked at the start of a step)";
  python tools/rocprof_summary.py timeline $DB 60; } > $OUT/timeline.txt
rm -rf /tmp/kt
PMCB="$BENCH --no-parity --no-clocks --steps 1 --warmup 1 --repeats 1"
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp; rm -rf /tmp/pmc_$c
  timeout 120 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -- $PMCB > $OUT/pmc_$c.bench.json 2>/dev/null
  cd $R; python tools/rocprof_summary.py pmc $(find /tmp/pmc_$c -name "*.db" | head -1) $c > $OUT/pmc_$c.txt 2>/dev/null; rm -rf /tmp/pmc_$c
done
cd /tmp; rm -rf /tmp/pmc_g
timeout 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace -d /tmp/pmc_g -- $PMCB > /dev/null 2>&1
cd $R; python tools/rocprof_summary.py pmc $(find /tmp/pmc_g -name "*.db" | head -1) 2>/dev/null | grep -i "hamming_knn2_mfma_kernel<2, 0>\|pose\|grid_points_fused\|counter" > $OUT/pmc_sq.txt; rm -rf /tmp/pmc_g
tools/latency.sh gpurun_out/$T/latency.txt > /dev/null 2>&1
bash tools/trace_latency.sh gpurun_out/$T/lat 100 > /dev/null 2>&1
TRACE_MODE=" " bash tools/trace_latency.sh gpurun_out/$T/lat_handler 100 > /dev/null 2>&1
#   6. the LSD line detector: kernel statistics of 1024 KITTI-size images per launch (tools/lsd_probe.py)
cd /tmp; rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/tools/lsd_probe.py --batch 1024 --iters 2 > $OUT/lsd_probe.txt 2>/dev/null
cd $R; { echo "# rocprofv3 --kernel-trace --stats -- python tools/lsd_probe.py --batch 1024 --iters 2 (3 launches incl. the parity pass); its output:"; tail -4 $OUT/lsd_probe.txt; echo;
  python tools/rocprof_summary.py stats $(find /tmp/kt -name "*.db" | head -1) | head -12; } > $OUT/lsd_kernel_stats.txt; rm -rf /tmp/kt
ls -la $OUT; head -c 600 $OUT/bench_default.json; echo; tail -3 $OUT/bench_default.err; head -20 $OUT/pmc_FETCH_SIZE.txt
