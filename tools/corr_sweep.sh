#!/bin/bash
timeout 200 python -m pytest tests/test_gpu_match.py -x -q 2>&1 | tail -2
timeout 100 python tools/corr_one.py 512 2>&1 | tail -2
timeout 100 python tools/corr_one.py 1024 2>&1 | tail -1
