#!/bin/bash
# Collects the evidence kept under profiles/ (run on the GPU box through gpurun):
#   tools/collect_profiles.sh gpurun_out/<dir>
R=$PWD; OUT=$R/$1; mkdir -p $OUT
python bench.py > $OUT/bench_default.json 2>/dev/null
python bench.py --overlap --no-cpu-baseline > $OUT/bench_overlap.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/bench.py > $OUT/bench_profiled.json 2>/dev/null
cd $R; python tools/rocprof_summary.py stats $(find /tmp/kt -name "*.db" | head -1) > $OUT/kernel_stats.txt; rm -rf /tmp/kt
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp; rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  cd $R; python tools/rocprof_summary.py pmc $(find /tmp/pmc_$c -name "*.db" | head -1) $c > $OUT/pmc_$c.txt; rm -rf /tmp/pmc_$c
done
tools/latency.sh $1/latency.txt > /dev/null
python tools/cpu_pipeline_time.py >> $OUT/latency.txt 2>/dev/null
python tools/bench_pipeline.py --batch 512 > $OUT/pipeline_b512.json 2>/dev/null
python tools/bench_pipeline.py --batch 512 --lines 0 > $OUT/pipeline_b512_points.json 2>/dev/null
python tools/bench_pipeline.py --batch 256 > $OUT/pipeline_b256.json 2>/dev/null
ls -la $OUT
