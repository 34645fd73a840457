#!/bin/bash
# Parity and cost of the lsd_grow_kernel variants (STVO_LSD_GROW, debug_switches.h) on the GPU box; output under gpurun_out/.
#   gpurun --timeout 200 -- 'bash tools/lsd_candidates.sh'
mkdir -p gpurun_out
export STVO_TEST_CANDIDATES=1
timeout 150 python -m pytest tests/test_gpu_lsd.py -x -q -k candidates > gpurun_out/lsd_cand_test.txt 2>&1
echo "candidate test exit $?" >> gpurun_out/lsd_cand_test.txt
for v in 7 0 2 3 6; do
    STVO_LSD_GROW=$v timeout 40 python tools/lsd_probe.py --batch 2 --iters 1 > gpurun_out/lsd_cand_probe_$v.txt 2>&1
done
for v in 7 0 6; do
    STVO_LSD_GROW=$v timeout 60 python tools/lsd_probe.py --batch 1024 --iters 2 2>&1 | tail -2 > gpurun_out/lsd_cand_b1024_$v.txt
done
tail -3 gpurun_out/lsd_cand_test.txt
grep -h "rows differ\|cycles" gpurun_out/lsd_cand_probe_*.txt
cat gpurun_out/lsd_cand_b1024_*.txt
