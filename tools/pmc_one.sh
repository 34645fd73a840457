#!/bin/bash
# One rocprofv3 --pmc pass over the default bench workload (1 step):  tools/pmc_one.sh <tag> <COUNTER> [timeout_s]
R=$PWD; OUT=$R/gpurun_out/$1; mkdir -p $OUT; C=$2; T=${3:-900}
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pmc_$C
timeout $T rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc_$C -- python $R/bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 --repeats 1 --no-parity > /dev/null 2>&1
cd $R; python tools/rocprof_summary.py pmc $(find /tmp/pmc_$C -name "*.db" | head -1) $C > $OUT/pmc_$C.txt 2>/dev/null; rm -rf /tmp/pmc_$C
head -25 $OUT/pmc_$C.txt | cut -c1-150
