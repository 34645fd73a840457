#!/bin/bash
# pipelined steps: parity at the headline shape, then A/B of the step time, then the timeline of the pipelined timed region
R=$PWD; OUT=$R/gpurun_out/r06_pipe; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_seq.py tests/test_gpu_grid.py -x -q > $OUT/tests.txt 2>&1; echo "tests rc $?"; tail -3 $OUT/tests.txt
Q="--no-cpu-baseline --no-extras --no-clocks --repeats 3"
run() {  # <tag> <env...>
  local tag=$1; shift
  env "$@" timeout 300 python bench.py $Q > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err; echo "$tag ($*) rc $?"
  python - $OUT/bench_$tag.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("  value %.0f  ms/step %.4f  parity %s  K1m %.3f pose %.3f grid %.3f" % (d["value"], d["ms_per_step"], d["parity_sampled"].get("ok"), d["roofline"]["avg_launch_ms"], d["roofline_pose"]["avg_launch_ms"], d["roofline_grid_scan"]["avg_launch_ms"]))
PY
}
run pipe0 STVO_SEQ_PIPE=0
run pipe1 STVO_SEQ_PIPE=1
run pipe1_static STVO_SEQ_PIPE=1 STVO_GRID_DYN=0
run pipe1_late STVO_SEQ_PIPE=1 STVO_LINE_FORK=late
run pipe1_nogate STVO_SEQ_PIPE=2
cd /tmp && export TMPDIR=/tmp
for v in "" "STVO_LINE_FORK=late"; do
rm -rf /tmp/kt
env $v timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -- python $R/bench.py $Q --no-parity > /dev/null 2>&1
echo "# timeline piped $v"; python $R/tools/rocprof_summary.py timeline $(find /tmp/kt -name "*.db" | head -1) 20 -500 | cut -c1-130
done > $OUT/timeline_piped.txt; cat $OUT/timeline_piped.txt
