#!/bin/bash
# round 5, call 2: the new tests (motion model on the pipeline and through the handler, bench --gpus 2 over gloo, LSD / ORB after the clean-up)
R=$PWD; OUT=$R/gpurun_out/r05_c2; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_seq.py -x -q -k motion_model > $OUT/seq_motion.txt 2>&1; tail -5 $OUT/seq_motion.txt
timeout 300 python -m pytest tests/test_gpu_handler.py -x -q -k motion_model > $OUT/handler_motion.txt 2>&1; tail -5 $OUT/handler_motion.txt
timeout 400 python -m pytest tests/test_gpu_shard.py -x -q > $OUT/shard.txt 2>&1; tail -15 $OUT/shard.txt
timeout 300 python -m pytest tests/test_gpu_lsd.py tests/test_gpu_orb.py -x -q > $OUT/lsd_orb.txt 2>&1; tail -5 $OUT/lsd_orb.txt
timeout 60 python tools/lsd_probe.py --batch 1 --iters 3 2>&1 | grep -E "rows differ|images:"
timeout 60 python tools/lsd_probe.py --batch 2 --iters 3 2>&1 | grep -E "rows differ|images:"
timeout 60 python tools/lsd_probe.py --batch 8 --iters 3 2>&1 | grep -E "rows differ|images:"
