#!/bin/bash
# Round-3 evidence, second half of the round (after the point-matcher / key-line work):  tools/collect_r03b.sh <tag>
#   bench line, kernel-trace stats (per kernel and per launch shape), the timeline of the last steps on both streams, the PMC passes
#   (FETCH_SIZE, WRITE_SIZE, matrix-pipe busy) over tools/bench_pipeline.py at the headline batch size, single-stream latency.
R=$PWD; T=$1
tools/collect_r03.sh $T > $R/gpurun_out/collect_$T.log 2>&1
tools/timeline.sh $T 36 > /dev/null 2>&1; mv $R/gpurun_out/timeline_$T.txt $R/gpurun_out/$T/timeline.txt
BP="python $R/tools/bench_pipeline.py --batch 1024 --steps 1 --warmup 1 --cpu-frames 0"
tools/pmc_cmd.sh $T pipe 200 FETCH_SIZE,WRITE_SIZE -- $BP > $R/gpurun_out/$T/pmc.log 2>&1
tools/pmc_cmd.sh $T pipe 200 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" -- $BP >> $R/gpurun_out/$T/pmc.log 2>&1
tools/latency.sh gpurun_out/$T/latency.txt > /dev/null 2>&1
timeout 200 python tools/latency_variants.py > $R/gpurun_out/$T/latency_variants.txt 2>&1
ls $R/gpurun_out/$T; head -c 400 $R/gpurun_out/$T/bench_default.json
