#!/bin/bash
# One counter pass over the bench workload with a retry (the passes hang now and then on this pool):  tools/pmc_r05.sh <tag> <COUNTER...>
R=$PWD; T=$1; shift; OUT=$R/gpurun_out/$T; mkdir -p $OUT
PMCB="python $R/bench.py --no-cpu-baseline --no-extras --no-parity --no-clocks --steps 1 --warmup 1 --repeats 1"
NAME=$(echo "$@" | tr ' ' '_')
for attempt in 1 2 3; do
  cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pmc_x; S=$(date +%s)
  timeout 75 rocprofv3 --pmc "$@" --kernel-trace -d /tmp/pmc_x -- $PMCB > /dev/null 2>&1; rc=$?
  echo "$NAME attempt $attempt: exit $rc, $(( $(date +%s) - S )) s"
  cd $R
  if [ $rc -eq 0 ]; then python tools/rocprof_summary.py pmc $(find /tmp/pmc_x -name "*.db" | head -1) "$@" > $OUT/pmc_$NAME.txt 2>/dev/null; rm -rf /tmp/pmc_x; head -12 $OUT/pmc_$NAME.txt | cut -c1-150; break; fi
done
