#!/bin/bash
# Round-3 evidence, run on the GPU box through gpurun:  tools/collect_r03.sh <tag> [pmc]
#   bench line (default command), rocprofv3 --kernel-trace --stats of the same workload summarised per kernel AND per launch shape
#   (tools/rocprof_summary.py split: K1m's key-point launch alone), optional PMC passes (FETCH_SIZE, WRITE_SIZE, matrix-pipe busy).
R=$PWD; OUT=$R/gpurun_out/$1; mkdir -p $OUT
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --no-cpu-baseline --no-extras"
rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -- $BENCH > $OUT/bench_profiled.json 2>/dev/null
DB=$(find /tmp/kt -name "*.db" | head -1)
cd $R
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-extras (tools/collect_r03.sh); bench line of the profiled run:"; cat $OUT/bench_profiled.json; echo;
  python tools/rocprof_summary.py stats $DB; echo; echo "# the same dispatches split by launch shape (grid_x in work-items x workgroup size): one line per problem size of a kernel";
  python tools/rocprof_summary.py split $DB; } > $OUT/kernel_stats.txt
rm -rf /tmp/kt
if [ "$2" = "pmc" ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    cd /tmp; rm -rf /tmp/pmc_$c
    timeout 420 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -- $BENCH --steps 1 --warmup 1 --repeats 1 --no-parity > /dev/null 2>&1
    cd $R; python tools/rocprof_summary.py pmc $(find /tmp/pmc_$c -name "*.db" | head -1) $c > $OUT/pmc_$c.txt 2>/dev/null; rm -rf /tmp/pmc_$c
  done
  cd /tmp; rm -rf /tmp/pmc_g
  timeout 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace -d /tmp/pmc_g -- $BENCH --steps 1 --warmup 1 --repeats 1 --no-parity > /dev/null 2>&1
  cd $R; python tools/rocprof_summary.py pmc $(find /tmp/pmc_g -name "*.db" | head -1) 2>/dev/null | grep -i "hamming_knn2_mfma_kernel<2, 0>\|pose\|grid_points_fused\|counter" > $OUT/pmc_sq.txt; rm -rf /tmp/pmc_g
fi
ls -la $OUT; head -c 1500 $OUT/bench_default.json; echo; tail -5 $OUT/bench_default.err
