#!/bin/bash
# the whole GPU suite as the driver runs it, then smoke()
R=$PWD; OUT=$R/gpurun_out/r06_tests; mkdir -p $OUT
timeout 1500 python -m pytest tests/ -x -q -m gpu > $OUT/gpu_tests.txt 2>&1; echo "gpu tests rc $?"; tail -4 $OUT/gpu_tests.txt
timeout 120 python __graft_entry__.py smoke 2>&1 | tail -2
