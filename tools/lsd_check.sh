#!/bin/bash
# LSD on the GPU box after a change: the parity tests (both forms of the growth kernel), the ORB tests (shared fastAtan2 / blur /
# resize), per-phase cycles of one image for both forms, batch timings.   gpurun --timeout 280 -- 'bash tools/lsd_check.sh'
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_gpu_lsd.py tests/test_gpu_orb.py -x -q > gpurun_out/lsd_orb_tests.txt 2>&1; echo "exit $?" >> gpurun_out/lsd_orb_tests.txt
tail -3 gpurun_out/lsd_orb_tests.txt
timeout 40 python tools/lsd_probe.py --batch 2 --iters 1 2>&1 | grep -E "rounds:|cycles|rows differ"
STVO_LSD_GROW=0 timeout 40 python tools/lsd_probe.py --batch 2 --iters 1 2>&1 | grep -E "rounds:|cycles"
timeout 60 python tools/lsd_probe.py --batch 1024 --iters 3 2>&1 | tail -1
# the many-waves kernel of the small batches (one image per XCD): its parity tests and the time of one / two / eight images
timeout 120 python -m pytest tests/test_gpu_lsd.py -x -q -k "many_waves or hostile" 2>&1 | tail -2
for B in 1 2 8; do timeout 60 python tools/lsd_probe.py --batch $B --iters 5 2>&1 | grep -E "rows differ|committer|feeder|images:"; done
