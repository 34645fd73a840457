#!/bin/bash
# Round 6, second session (the LSD work): evidence on the GPU box.   gpurun --timeout 1500 -- 'bash tools/collect_r06b.sh'
#   1. the driver's command (short line + extras);  2. LSD: one / two / eight images per call with the committer's, the speculating waves' and the
#   dispatcher's counters, mid-size batches against one wave per image, kernel stats of one image and of the batched leg (8192 + 4096 per launch),
#   SQ counters of the one-wave growth kernel at 1024 images;  3. kernel stats of images -> poses with key-lines (2048 streams).
R=$PWD; OUT=$R/gpurun_out/r06b; mkdir -p $OUT
S=$(date +%s); timeout 800 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc $? $(wc -c < $OUT/bench_default.json) bytes, $(( $(date +%s) - S )) s"
cp bench_extras.json $OUT/bench_extras_default.json
{ for b in 1 2 8; do timeout 60 python tools/lsd_probe.py --batch $b --iters 7 2>&1 | grep -E "rows differ|committer|feeder|speculating|dispatcher|per call|images:"; done; } > $OUT/lsd_small_batches.txt 2>&1
BL="16 32 64 128" bash tools/r06_lsd_mid.sh > /dev/null 2>&1; cp gpurun_out/lsd_all/mid.txt $OUT/lsd_mid_batches.txt
bash tools/lsd_prof.sh 1 > /dev/null 2>&1; cp gpurun_out/lsd_prof/kernel_stats.txt $OUT/lsd_one_image_kernel_stats.txt 2>/dev/null
bash tools/r06_lsd_prof4k.sh > /dev/null 2>&1; cp gpurun_out/lsd_prof/kernel_stats4k.txt $OUT/lsd_batched_kernel_stats.txt; grep images_per_s gpurun_out/lsd_prof/trace4k.out > $OUT/lsd_batched_leg.json
bash tools/r06_lsd_sq.sh 1024 > /dev/null 2>&1; cp gpurun_out/lsd_sq/pmc_sq.txt $OUT/lsd_one_wave_sq_counters.txt
bash tools/r06_images_prof.sh 2048 > /dev/null 2>&1; cp gpurun_out/img_prof/kernel_stats.txt $OUT/images_with_lines_kernel_stats.txt; grep stereo_pairs_per_s gpurun_out/img_prof/trace.out > $OUT/images_with_lines_leg.json
ls -la $OUT; head -c 300 $OUT/bench_default.json; echo; tail -2 $OUT/bench_default.err; tail -3 $OUT/lsd_small_batches.txt
