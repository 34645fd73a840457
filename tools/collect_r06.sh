#!/bin/bash
# Round-6 evidence, run on the GPU box:  gpurun --timeout 1500 -- 'bash tools/collect_r06.sh r06'
#   1. the default command exactly as the driver runs it (short line -> bench_default.json, full record -> bench_extras_default.json);
#   2. rocprofv3 --kernel-trace --stats of `bench.py --no-cpu-baseline --no-extras`: per kernel, per launch shape, the timeline of the TIMED region and of
#      the LIGHT pass (the one bench.py's roofline objects are timed in) side by side;
#   3. counter passes over the bench workload itself, >= 10 dispatches per kernel (--steps 4: 1 warm-up + 4 timed + 4 counting + 4 light), separate
#      passes for FETCH_SIZE and WRITE_SIZE (MI355X_MICROARCH.md), each with up to 3 attempts (a pass hangs now and then on this pool), + the calibration
#      of the two counters on known byte counts (tools/hbm_calib), + matrix-pipe busy and the wave-cycle counters of K1m / pose;
#   4. clustered-descriptor match, single-stream latency (+ kernel chain of one frame), LSD one image / 1024 images.
R=$PWD; T=${1:-r06}; OUT=$R/gpurun_out/$T; mkdir -p $OUT
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc $? $(wc -c < $OUT/bench_default.json) bytes"
cp bench_extras.json $OUT/bench_extras_default.json
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --no-cpu-baseline --no-extras"
rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -- $BENCH --no-parity --no-clocks > $OUT/bench_profiled.json 2>/dev/null
DB=$(find /tmp/kt -name "*.db" | head -1)
cd $R
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-extras --no-parity --no-clocks (tools/collect_r06.sh): 6 warm-up + 5 x 24 timed + 24 counting-pass"
  echo "# + 24 light-pass steps; bench line of the profiled run:"; cat $OUT/bench_profiled.json; echo;
  python tools/rocprof_summary.py stats $DB; echo; echo "# the same dispatches split by launch shape (grid_x in work-items x workgroup size)";
  python tools/rocprof_summary.py split $DB; } > $OUT/kernel_stats.txt
{ echo "# 'python bench.py --no-cpu-baseline --no-extras --no-parity --no-clocks' under rocprofv3 --kernel-trace: every dispatch in start order, queue = HIP stream (1 = the point"
  echo "# stream, 3 = the key-line stream, which also builds the NEXT frame's grid — point_cells_kernel — once its key-line work is done)."
  echo "# (a) steps of the TIMED region (no markers, no reads):"
  python tools/rocprof_summary.py timeline $DB 40 -600;
  echo; echo "# (b) the last steps of the run = the LIGHT pass (event pairs around grid_points_fused / hamming_knn2_mfma<2, 0> / pose2c only; what roofline* is timed in):"
  python tools/rocprof_summary.py timeline $DB 40; } > $OUT/timeline.txt
rm -rf /tmp/kt
bash tools/hbm_calib.sh > $OUT/hbm_calib.txt 2>&1
# (the streams come from a cache written by an unprofiled run: under rocprofv3 the generator's forked worker processes are what made
#  counter passes hang on this pool — with the cache every pass of round 6's last day ran at the first attempt in 3 - 4 s)
python $R/bench.py --no-cpu-baseline --no-extras --no-parity --no-clocks --steps 1 --warmup 1 --repeats 1 --streams-cache /tmp/streams3072.pkl > /dev/null 2>&1
PMCB="$BENCH --no-parity --no-clocks --steps 4 --warmup 1 --repeats 1 --streams-cache /tmp/streams3072.pkl"
pmc_pass() {  # <file tag> <grep pattern or -> <counters...>
  local tag=$1 pat=$2; shift 2
  for attempt in 1 2 3; do
    cd /tmp; rm -rf /tmp/pmc_x; local S=$(date +%s)
    timeout 60 rocprofv3 --pmc "$@" --kernel-trace -d /tmp/pmc_x -- $PMCB > /dev/null 2>&1; local rc=$?
    echo "$tag attempt $attempt: exit $rc, $(( $(date +%s) - S )) s"
    cd $R
    if [ $rc -eq 0 ]; then
      if [ "$pat" = "-" ]; then python tools/rocprof_summary.py pmc $(find /tmp/pmc_x -name "*.db" | head -1) > $OUT/pmc_$tag.txt 2>/dev/null
      else python tools/rocprof_summary.py pmc $(find /tmp/pmc_x -name "*.db" | head -1) 2>/dev/null | grep -i "$pat" > $OUT/pmc_$tag.txt; fi
      rm -rf /tmp/pmc_x; return 0
    fi
  done
  return 1
}
pmc_pass FETCH_SIZE - FETCH_SIZE
pmc_pass WRITE_SIZE - WRITE_SIZE
pmc_pass sq "hamming_knn2_mfma\|pose\|grid_points_fused\|counter" SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES
pmc_pass wave_cycles "hamming_knn2_mfma_kernel\|pose2c\|grid_points_fused\|counter" SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS
bash tools/corr_prof.sh > $OUT/clustered_match.txt 2>&1; cp gpurun_out/corr/kernel_stats.txt $OUT/clustered_match_kernel_stats.txt 2>/dev/null
tools/latency.sh gpurun_out/$T/latency.txt > /dev/null 2>&1
bash tools/trace_latency.sh gpurun_out/$T 100 > /dev/null 2>&1
{ for b in 1 2 8; do timeout 60 python tools/lsd_probe.py --batch $b --iters 3 2>&1 | grep -E "rows differ|images:|committer"; done; } > $OUT/lsd_small_batches.txt 2>&1
bash tools/lsd_prof.sh 1 > /dev/null 2>&1; cp gpurun_out/lsd_prof/kernel_stats.txt $OUT/lsd_one_image_kernel_stats.txt 2>/dev/null
ls -la $OUT; head -c 600 $OUT/bench_default.json; echo; tail -2 $OUT/bench_default.err; head -8 $OUT/pmc_FETCH_SIZE.txt | cut -c1-150; head -8 $OUT/pmc_WRITE_SIZE.txt | cut -c1-150; cat $OUT/pmc_sq.txt | cut -c1-150
