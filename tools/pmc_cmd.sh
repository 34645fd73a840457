#!/bin/bash
# One rocprofv3 --pmc pass per counter over an arbitrary command:  tools/pmc_cmd.sh <tag> <name> <timeout_s> <COUNTER[,COUNTER..]> -- <command...>
R=$PWD; OUT=$R/gpurun_out/$1; NAME=$2; T=$3; CS=$4; shift 5; mkdir -p $OUT
for C in ${CS//,/ }; do
  cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pmc_x
  S=$(date +%s); timeout $T rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc_x -- "$@" > $OUT/pmc_${NAME}_$C.err 2>&1; E=$?
  cd $R; python tools/rocprof_summary.py pmc $(find /tmp/pmc_x -name "*.db" 2>/dev/null | head -1) $C > $OUT/pmc_${NAME}_$C.txt 2>/dev/null; rm -rf /tmp/pmc_x
  echo "== $NAME $C: exit $E, $(( $(date +%s) - S )) s"; head -6 $OUT/pmc_${NAME}_$C.txt | cut -c1-150
done
