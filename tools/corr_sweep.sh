#!/bin/bash
timeout 200 python -m pytest tests/test_gpu_match.py -x -q 2>&1 | tail -2
timeout 100 python tools/corr_one.py 512 2>&1 | tail -2
for sl in 2 3 8; do echo "slots $sl"; STVO_TMP_REV_SLOTS=$sl timeout 100 python tools/corr_one.py 512 2>&1 | tail -1; done
