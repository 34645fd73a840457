#!/bin/bash
# LSD after a change of the growth kernels: parity tests, one / two / eight images per call (many-waves form), 1024 / 4096 images (one wave each).
#   gpurun --timeout 900 -- 'bash tools/r06_lsd_all.sh'
mkdir -p gpurun_out/lsd_all; O=gpurun_out/lsd_all
timeout 300 python -m pytest tests/test_gpu_lsd.py -x -q > $O/tests.txt 2>&1; echo "exit $?" >> $O/tests.txt; tail -3 $O/tests.txt
: > $O/probe.txt
for B in 1 2 8; do timeout 60 python tools/lsd_probe.py --batch $B --iters 7 2>&1 | grep -E "rows differ|committer|feeder|images:" | tee -a $O/probe.txt; done
STVO_LSD_WAVES=0 timeout 60 python tools/lsd_probe.py --batch 1 --iters 2 2>&1 | grep -E "rounds:|cycles|images:" | tee -a $O/probe.txt
for B in 1024 4096; do timeout 120 python tools/lsd_probe.py --batch $B --iters 3 2>&1 | grep -E "rows differ|images:" | tee -a $O/probe.txt; done
