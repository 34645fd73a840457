#!/bin/bash
# lsd_grow_xcd_kernel (round 6): parity tests, then one / two / eight images per call by the number of speculating workgroups per image.
#   gpurun --timeout 600 -- 'bash tools/r06_lsd_xcd.sh'
mkdir -p gpurun_out/lsd_xcd
O=gpurun_out/lsd_xcd
timeout 300 python -m pytest tests/test_gpu_lsd.py -x -q > $O/tests.txt 2>&1; echo "exit $?" >> $O/tests.txt
tail -3 $O/tests.txt
for nsb in ${NSB_LIST:-0 1 2 4 8 16 31}; do
  echo "== STVO_LSD_XCD_BLOCKS=$nsb" | tee -a $O/probe.txt
  STVO_LSD_XCD_BLOCKS=$nsb timeout 60 python tools/lsd_probe.py --batch 1 --iters 5 2>&1 | grep -E "rows differ|committer|images:|feeder|speculating|dispatcher" | tee -a $O/probe.txt
done
for B in 2 8; do
  echo "== batch $B" | tee -a $O/probe.txt
  timeout 60 python tools/lsd_probe.py --batch $B --iters 5 2>&1 | grep -E "rows differ|images:" | tee -a $O/probe.txt
done
echo "== one wave per image" | tee -a $O/probe.txt
STVO_LSD_WAVES=0 timeout 60 python tools/lsd_probe.py --batch 1 --iters 5 2>&1 | grep -E "rows differ|committer|images:|feeder|speculating|dispatcher" | tee -a $O/probe.txt
