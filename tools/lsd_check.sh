mkdir -p gpurun_out
export STVO_TEST_CANDIDATES=1
timeout 170 python -m pytest tests/test_gpu_lsd.py tests/test_gpu_orb.py -x -q > gpurun_out/lsd_orb_tests.txt 2>&1; echo "exit $?" >> gpurun_out/lsd_orb_tests.txt
tail -3 gpurun_out/lsd_orb_tests.txt
timeout 40 python tools/lsd_probe.py --batch 2 --iters 1 2>&1 | grep -E "rounds:|cycles|rows differ"
timeout 60 python tools/lsd_probe.py --batch 1024 --iters 2 2>&1 | tail -1
