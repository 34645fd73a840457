#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of tools/hbm_calib (known byte counts), one --pmc pass per counter:  gpurun --timeout 300 -- 'bash tools/hbm_calib.sh'
R=$PWD; OUT=$R/gpurun_out/hbm_calib; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/cal_$c
  timeout 120 rocprofv3 --pmc $c --kernel-trace -d /tmp/cal_$c -- $R/tools/hbm_calib > $OUT/run_$c.txt 2>&1; echo "$c: exit $?"
  cd $R; python tools/rocprof_summary.py pmc $(find /tmp/cal_$c -name "*.db" | head -1) $c > $OUT/pmc_$c.txt 2>/dev/null; rm -rf /tmp/cal_$c
  grep calib $OUT/pmc_$c.txt | cut -c1-140
done
tail -1 $OUT/run_FETCH_SIZE.txt
