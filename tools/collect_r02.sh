#!/bin/bash
# Collects the round-2 evidence kept under profiles/ (run on the GPU box through gpurun):
#   tools/collect_r02.sh gpurun_out/<dir>
# bench line, rocprofv3 kernel stats of the same command, HBM traffic counters (one --pmc pass per counter; WRITE_SIZE passes
# take minutes), SQ counters of the matching kernel, single-stream latency, pose-kernel variants, ORB front-end.
R=$PWD; OUT=$R/$1; mkdir -p $OUT
BENCH="python $R/bench.py --no-cpu-baseline --no-extras"
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/kt -- $BENCH > $OUT/bench_profiled.json 2>/dev/null
cd $R; python tools/rocprof_summary.py stats $(find /tmp/kt -name "*.db" | head -1) > $OUT/kernel_stats.txt; rm -rf /tmp/kt
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp; rm -rf /tmp/pmc_$c
  timeout 360 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -- $BENCH --steps 1 --warmup 1 > /dev/null 2>&1
  cd $R; python tools/rocprof_summary.py pmc $(find /tmp/pmc_$c -name "*.db" | head -1) $c | head -40 > $OUT/pmc_$c.txt; rm -rf /tmp/pmc_$c
done
: > $OUT/pmc_sq.txt
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVES"; do
  cd /tmp; rm -rf /tmp/pmc_g
  timeout 150 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmc_g -- $BENCH --steps 1 --warmup 1 > /dev/null 2>&1
  cd $R; python tools/rocprof_summary.py pmc $(find /tmp/pmc_g -name "*.db" | head -1) | grep -i "hamming_knn2_mfma_kernel<2, 0>\|pose\|grid_points_fused\|counter" >> $OUT/pmc_sq.txt; rm -rf /tmp/pmc_g
done
tools/latency.sh $1/latency.txt > /dev/null 2>&1
python tools/cpu_pipeline_time.py >> $OUT/latency.txt 2>/dev/null
{ for cfg in "1 0 1" "1 0 1024" "2 40 1" "2 40 512" "2 40 1024" "2 8 1024"; do set -- $cfg
    echo "STVO_POSE_KERNEL=$1 STVO_POSE2_NW=$2 batch $3: $(STVO_POSE_KERNEL=$1 STVO_POSE2_NW=$2 python tools/pose_probe.py --batch $3 --iters 5 2>/dev/null | tail -1)"; done; } > $OUT/pose_variants.txt
python tools/bench_orb.py --batch 256 > $OUT/orb_bench.json 2>/dev/null
cd /tmp; rocprofv3 --kernel-trace --stats -d /tmp/kto -- python $R/tools/bench_orb.py --batch 256 --iters 5 > /dev/null 2>&1
cd $R; python tools/rocprof_summary.py stats $(find /tmp/kto -name "*.db" | head -1) | head -12 > $OUT/orb_kernel_stats.txt; rm -rf /tmp/kto
ls -la $OUT
