#!/bin/bash
# Round-5 evidence, run on the GPU box through gpurun:  gpurun --timeout 900 -- 'bash tools/collect_r05.sh r05'
#   1. bench line of the default command (bench_default.json);
#   2. rocprofv3 --kernel-trace --stats of `bench.py --no-cpu-baseline --no-extras`, per kernel and per launch shape, + the timeline of the last steps;
#   3. the counter calibration (tools/hbm_calib: known byte counts) and the PMC passes over THE BENCH WORKLOAD ITSELF: FETCH_SIZE, WRITE_SIZE
#      (separate passes, as the HBM section of MI355X_MICROARCH.md prescribes), matrix-pipe / CU busy cycles;
#   4. the f2f match on clustered descriptors: kernel stats (tools/corr_prof.sh);
#   5. single-stream latency (handler in both modes, stvo_seq_push) and the LSD detector (one image, 1024 images).
R=$PWD; T=${1:-r05}; OUT=$R/gpurun_out/$T; mkdir -p $OUT
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --no-cpu-baseline --no-extras"
rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -- $BENCH > $OUT/bench_profiled.json 2>/dev/null
DB=$(find /tmp/kt -name "*.db" | head -1)
cd $R
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-extras (tools/collect_r05.sh); bench line of the profiled run:"; cat $OUT/bench_profiled.json; echo;
  python tools/rocprof_summary.py stats $DB; echo; echo "# the same dispatches split by launch shape (grid_x in work-items x workgroup size): one line per problem size of a kernel";
  python tools/rocprof_summary.py split $DB; } > $OUT/kernel_stats.txt
{ echo "# 'python bench.py --no-cpu-baseline --no-extras' under rocprofv3 --kernel-trace (tools/collect_r05.sh): every dispatch in start order, queue = HIP stream";
  echo "# (1 = the point stream, 3 = the key-line stream: forked in front of the point matcher, and the builder of the NEXT frame's grid — point_cells_kernel — once its key-line work is done).  (a) steps of the TIMED region (dispatches 400 .. 447 of the run: steps back";
  echo "# to back, no markers, no reads):";
  python tools/rocprof_summary.py timeline $DB 48 -400;
  echo; echo "# (b) the last dispatches of the run (the pass with an event pair around every kernel, then the parity sample): the markers change what runs beside what";
  python tools/rocprof_summary.py timeline $DB 36; } > $OUT/timeline.txt
rm -rf /tmp/kt
bash tools/hbm_calib.sh > $OUT/hbm_calib.txt 2>&1
PMCB="$BENCH --no-parity --no-clocks --steps 1 --warmup 1 --repeats 1"
for c in ${PMC_PASSES:-FETCH_SIZE WRITE_SIZE}; do   # (PMC_PASSES=WRITE_SIZE: skip a pass that keeps hanging)
  cd /tmp; rm -rf /tmp/pmc_$c
  timeout 150 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -- $PMCB > $OUT/pmc_$c.bench.json 2>/dev/null; echo "$c pass: exit $?"
  cd $R; python tools/rocprof_summary.py pmc $(find /tmp/pmc_$c -name "*.db" | head -1) $c > $OUT/pmc_$c.txt 2>/dev/null; rm -rf /tmp/pmc_$c
done
cd /tmp; rm -rf /tmp/pmc_g
timeout 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace -d /tmp/pmc_g -- $PMCB > /dev/null 2>&1
cd $R; python tools/rocprof_summary.py pmc $(find /tmp/pmc_g -name "*.db" | head -1) 2>/dev/null | grep -i "hamming_knn2_mfma\|pose\|grid_points_fused\|counter" > $OUT/pmc_sq.txt; rm -rf /tmp/pmc_g
# where a wave's cycles go (quad-cycles: active / issue-stalled / parked), the forward scan and the pose kernel
cd /tmp; rm -rf /tmp/pmc_w
timeout 60 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d /tmp/pmc_w -- $PMCB > /dev/null 2>&1; echo "wave-cycle pass: exit $?"
cd $R; python tools/rocprof_summary.py pmc $(find /tmp/pmc_w -name "*.db" | head -1) 2>/dev/null | grep -i "hamming_knn2_mfma_kernel\|pose2c\|counter" > $OUT/pmc_wave_cycles.txt; rm -rf /tmp/pmc_w
bash tools/corr_prof.sh > $OUT/clustered_match.txt 2>&1; cp gpurun_out/corr/kernel_stats.txt $OUT/clustered_match_kernel_stats.txt 2>/dev/null
tools/latency.sh gpurun_out/$T/latency.txt > /dev/null 2>&1
cd /tmp; rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/tools/lsd_probe.py --batch 1024 --iters 2 > $OUT/lsd_probe.txt 2>/dev/null
cd $R; { echo "# rocprofv3 --kernel-trace --stats -- python tools/lsd_probe.py --batch 1024 --iters 2 (3 launches incl. the parity pass); its output:"; tail -4 $OUT/lsd_probe.txt; echo;
  python tools/rocprof_summary.py stats $(find /tmp/kt -name "*.db" | head -1) | head -12; } > $OUT/lsd_kernel_stats.txt; rm -rf /tmp/kt
{ for b in 1 2 8; do timeout 60 python tools/lsd_probe.py --batch $b --iters 3 2>&1 | grep -E "rows differ|images:|committer"; done; } > $OUT/lsd_small_batches.txt 2>&1
bash tools/lsd_prof.sh 1 > /dev/null 2>&1; cp gpurun_out/lsd_prof/kernel_stats.txt $OUT/lsd_one_image_kernel_stats.txt 2>/dev/null
ls -la $OUT; head -c 400 $OUT/bench_default.json; echo; tail -2 $OUT/bench_default.err; head -8 $OUT/pmc_FETCH_SIZE.txt | cut -c1-150; cat $OUT/pmc_sq.txt | cut -c1-150
