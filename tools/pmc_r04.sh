#!/bin/bash
# The counter passes of tools/collect_r04.sh alone (FETCH_SIZE, WRITE_SIZE, matrix pipe) over the bench workload:  tools/pmc_r04.sh <tag>
R=$PWD; T=$1; OUT=$R/gpurun_out/$T; mkdir -p $OUT
PMCB="python $R/bench.py --no-cpu-baseline --no-extras --no-parity --no-clocks --steps 1 --warmup 1 --repeats 1"
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pmc_$c; S=$(date +%s)
  timeout 400 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -- $PMCB > $OUT/pmc_$c.bench.json 2>/dev/null; echo "$c: exit $?, $(( $(date +%s) - S )) s"
  cd $R; python tools/rocprof_summary.py pmc $(find /tmp/pmc_$c -name "*.db" | head -1) $c > $OUT/pmc_$c.txt 2>/dev/null; rm -rf /tmp/pmc_$c
done
cd /tmp; rm -rf /tmp/pmc_g
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace -d /tmp/pmc_g -- $PMCB > /dev/null 2>&1
cd $R; python tools/rocprof_summary.py pmc $(find /tmp/pmc_g -name "*.db" | head -1) 2>/dev/null | grep -i "hamming_knn2_mfma_kernel<2, 0>\|pose\|grid_points_fused\|counter" > $OUT/pmc_sq.txt; rm -rf /tmp/pmc_g
head -14 $OUT/pmc_FETCH_SIZE.txt | cut -c1-160; head -8 $OUT/pmc_WRITE_SIZE.txt | cut -c1-160; cat $OUT/pmc_sq.txt | cut -c1-160
