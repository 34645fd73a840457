#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the ORB front-end's kernels: one rocprofv3 --pmc pass per counter over tools/bench_orb.py (2 iterations).
#   tools/pmc_orb.sh <tag> [timeout_s]
R=$PWD; OUT=$R/gpurun_out/$1; mkdir -p $OUT; T=${2:-180}
for C in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pmc_$C
  timeout $T rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc_$C -- python $R/tools/bench_orb.py --iters 2 > $OUT/pmc_orb_$C.err 2>&1
  cd $R; python tools/rocprof_summary.py pmc $(find /tmp/pmc_$C -name "*.db" 2>/dev/null | head -1) $C > $OUT/pmc_orb_$C.txt 2>/dev/null; rm -rf /tmp/pmc_$C
  tail -2 $OUT/pmc_orb_$C.err; head -8 $OUT/pmc_orb_$C.txt | cut -c1-140
done
