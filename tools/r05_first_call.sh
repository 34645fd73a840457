#!/bin/bash
# What round 5 should measure FIRST, in one gpurun call (~4 GPU-minutes: every step under its own timeout, outputs under gpurun_out/r05_first/):
#   1. the LSD kernels after the end-of-round-4 work, and the 16-waves-per-image kernel that has never run (parity test + two-image time);
#   2. the ORB front-end: ms per launch (no PyTorch), then SQ counters of its kernels over that command — instruction mix and busy cycles
#      of orb_fast_nms_kernel / orb_describe_kernel (only counters that are known to work on this pool: NOTES.md lists the ones that hang);
#   3. the headline bench line without the extras.
#   gpurun --timeout 420 -- 'bash tools/r05_first_call.sh'
R=$PWD; OUT=$R/gpurun_out/r05_first; mkdir -p $OUT
bash tools/lsd_check.sh > $OUT/lsd_check.txt 2>&1
timeout 30 python tools/orb_probe.py > $OUT/orb_probe.txt 2>&1
# the blur that requests rows ahead (never run): parity, then the same probe with it
STVO_TEST_BLUR_AHEAD=1 timeout 60 python -m pytest tests/test_gpu_orb.py -x -q -k rows_ahead > $OUT/blur_ahead_test.txt 2>&1
STVO_BLUR_AHEAD=1 timeout 30 python tools/orb_probe.py > $OUT/orb_probe_blur_ahead.txt 2>&1
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_BUSY_CU_CYCLES SQ_WAVES"; do
    tag=$(echo $grp | tr ' ' '_')
    cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pmc_o
    timeout 60 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmc_o -- python $R/tools/orb_probe.py --iters 2 > $OUT/pmc_$tag.err 2>&1
    cd $R; python tools/rocprof_summary.py pmc $(find /tmp/pmc_o -name "*.db" 2>/dev/null | head -1) > $OUT/pmc_$tag.txt 2>/dev/null; rm -rf /tmp/pmc_o
done
timeout 150 python bench.py --no-extras --no-cpu-baseline > $OUT/bench_quick.json 2> $OUT/bench_quick.err
tail -12 $OUT/lsd_check.txt; cat $OUT/orb_probe.txt; tail -1 $OUT/blur_ahead_test.txt; cat $OUT/orb_probe_blur_ahead.txt; grep -h "orb_" $OUT/pmc_*.txt | cut -c1-150 | head -16; cut -c1-300 $OUT/bench_quick.json
