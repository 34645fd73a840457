#!/bin/bash
# mid-size LSD batches: the many-waves form (several images per XCD) against one wave per image.  gpurun --timeout 600 -- 'bash tools/r06_lsd_mid.sh'
mkdir -p gpurun_out/lsd_all; O=gpurun_out/lsd_all/mid.txt; : > $O
for B in ${BL:-12 16 32 64 128}; do
  echo "== $B images, many waves" | tee -a $O
  timeout 90 python tools/lsd_probe.py --batch $B --iters 5 2>&1 | grep -E "rows differ|images:" | tee -a $O
  echo "== $B images, one wave each" | tee -a $O
  STVO_LSD_WAVES=0 timeout 90 python tools/lsd_probe.py --batch $B --iters 3 2>&1 | grep -E "images:" | tee -a $O
done
