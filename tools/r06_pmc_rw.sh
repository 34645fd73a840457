#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes over the bench workload (final code), with retries:  gpurun --timeout 900 -- 'bash tools/r06_pmc_rw.sh'
R=$PWD; OUT=$R/gpurun_out/r06; mkdir -p $OUT
export STVO_LINES_AHEAD=0
PMCB="python $R/bench.py --no-cpu-baseline --no-extras --no-parity --no-clocks --steps 4 --warmup 1 --repeats 1 --streams-cache /tmp/streams3072.pkl"
cd /tmp && export TMPDIR=/tmp
$PMCB > /dev/null 2>&1; ls -la /tmp/streams3072.pkl   # unprofiled: generates the streams once (forks its workers outside the profiler)
for c in FETCH_SIZE WRITE_SIZE; do
  for attempt in 1 2 3; do
    rm -rf /tmp/pmc_x; S=$(date +%s)
    timeout 60 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_x -- $PMCB > /dev/null 2>&1; rc=$?; echo "$c attempt $attempt: exit $rc $(( $(date +%s) - S )) s"
    if [ $rc -eq 0 ]; then python $R/tools/rocprof_summary.py pmc $(find /tmp/pmc_x -name "*.db" | head -1) $c > $OUT/pmc_$c.txt 2>/dev/null; break; fi
  done
done
bash $R/tools/hbm_calib.sh > $OUT/hbm_calib.txt 2>&1
grep "pose2c\|knn2_mfma_kernel\|grid_points" $OUT/pmc_FETCH_SIZE.txt $OUT/pmc_WRITE_SIZE.txt | cut -c1-170
