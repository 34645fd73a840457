# The whole GPU suite N times, each in a fresh process, failures collected (round 6: one flaky failure showed only on some boxes of the pool).
# Run from the repo root on a GPU box: bash tools/soak_gpu_tests.sh [N]
N=${1:-5}
mkdir -p gpurun_out/soak; : > gpurun_out/soak/summary.txt
for i in $(seq 1 $N); do
  timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/soak/run_$i.txt 2>&1
  echo "run $i: $(tail -1 gpurun_out/soak/run_$i.txt)" >> gpurun_out/soak/summary.txt
  grep -E "^FAILED|^E  " gpurun_out/soak/run_$i.txt | head -5 | cut -c1-200 >> gpurun_out/soak/summary.txt
done
cat gpurun_out/soak/summary.txt
