#!/bin/bash
R=$PWD; OUT=$R/gpurun_out/r02_final3; mkdir -p $OUT
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats -d /tmp/kt -- $BENCH > $OUT/bench_profiled.json 2>/dev/null
cd $R; python tools/rocprof_summary.py stats $(find /tmp/kt -name "*.db" | head -1) > $OUT/kernel_stats.txt; rm -rf /tmp/kt
cd /tmp; rm -rf /tmp/pmc_g
timeout 170 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace -d /tmp/pmc_g -- $BENCH --steps 1 --warmup 1 > /dev/null 2>&1
cd $R; python tools/rocprof_summary.py pmc $(find /tmp/pmc_g -name "*.db" | head -1) 2>/dev/null | grep -i "hamming_knn2_mfma_kernel<2, 0>\|pose\|grid_points_fused\|counter" > $OUT/pmc_sq.txt; rm -rf /tmp/pmc_g
tools/latency.sh gpurun_out/r02_final3/latency.txt > /dev/null 2>&1
python tools/cpu_pipeline_time.py >> $OUT/latency.txt 2>/dev/null
ls -la $OUT; cat $OUT/pmc_sq.txt | cut -c1-150
