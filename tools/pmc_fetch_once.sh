#!/bin/bash
# One FETCH_SIZE pass over the bench workload with a short leash (the pass of tools/collect_r04.sh hung once):  tools/pmc_fetch_once.sh <tag> [counter]
R=$PWD; T=$1; C=${2:-FETCH_SIZE}; OUT=$R/gpurun_out/$T; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pmc_$C; S=$(date +%s)
timeout 45 rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc_$C -- python $R/bench.py --no-cpu-baseline --no-extras --no-parity --no-clocks --steps 1 --warmup 1 --repeats 1 > $OUT/pmc_$C.bench.json 2>/dev/null
echo "$C: exit $?, $(( $(date +%s) - S )) s"
cd $R; python tools/rocprof_summary.py pmc $(find /tmp/pmc_$C -name "*.db" | head -1) $C > $OUT/pmc_$C.txt 2>/dev/null; rm -rf /tmp/pmc_$C
head -12 $OUT/pmc_$C.txt | cut -c1-160
