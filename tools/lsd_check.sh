#!/bin/bash
# LSD on the GPU box after a change: the parity tests (candidate variants included), the handler / image-pipeline tests that
# use key-lines from images, per-phase cycles of one image, batch timings.   gpurun --timeout 280 -- 'bash tools/lsd_check.sh'
mkdir -p gpurun_out
export STVO_TEST_CANDIDATES=1
timeout 200 python -m pytest tests/test_gpu_lsd.py tests/test_gpu_orb.py -x -q > gpurun_out/lsd_orb_tests.txt 2>&1; echo "exit $?" >> gpurun_out/lsd_orb_tests.txt
tail -3 gpurun_out/lsd_orb_tests.txt
timeout 40 python tools/lsd_probe.py --batch 2 --iters 1 2>&1 | grep -E "rounds:|cycles|rows differ"
timeout 60 python tools/lsd_probe.py --batch 1024 --iters 3 2>&1 | tail -1
STVO_LSD_SORT_FULL=1 timeout 60 python tools/lsd_probe.py --batch 1024 --iters 3 2>&1 | tail -1
