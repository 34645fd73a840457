#!/bin/bash
# lsd_grow_xcd_kernel: one image per call by the developer knobs.   gpurun --timeout 300 -- 'bash tools/r06_lsd_sweep.sh "STVO_LSD_FEED_AHEAD=128" "STVO_LSD_SEP=32 STVO_LSD_AHEAD=8192" ...'
mkdir -p gpurun_out/lsd_xcd
for cfg in "$@"; do
  echo "== $cfg" | tee -a gpurun_out/lsd_xcd/sweep.txt
  env $cfg timeout 60 python tools/lsd_probe.py --batch ${BATCH:-1} --iters 10 2>&1 | grep -E "rows differ|committer|images:|feeder|speculating|dispatcher" | tee -a gpurun_out/lsd_xcd/sweep.txt
done
