# The LSD tests N times, each in a fresh process (the empty-first-image failure of round 6 showed in about every second process on
# some boxes), then the whole GPU suite once.  Run from the repo root on a GPU box: bash tools/r06_repro.sh [N]
N=${1:-8}
mkdir -p gpurun_out/repro; : > gpurun_out/repro/log.txt
for i in $(seq 1 $N); do timeout 300 python -m pytest tests/test_gpu_lsd.py tests/test_gpu_images.py -x -q 2>&1 | grep -E "^E  |^tests/|passed|failed|assert" | cut -c1-220 | head -12 >> gpurun_out/repro/log.txt; done
echo "runs with 'passed':" $(grep -c passed gpurun_out/repro/log.txt) "of $N; failures:"; grep -E "failed|^E" gpurun_out/repro/log.txt | head -20
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/repro/full.txt; tail -5 gpurun_out/repro/full.txt
