#!/bin/bash
R=$PWD; OUT=$R/gpurun_out/r06_ahead; mkdir -p $OUT
Q="--no-cpu-baseline --no-extras --no-clocks --repeats 5"
run() {  # <tag> <env...>
  local tag=$1; shift
  env "$@" timeout 300 python bench.py $Q > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err; echo "$tag ($*) rc $?"
  python - $OUT/bench_$tag.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("  value %.0f  ms/step %.4f  parity %s  K1m %.3f pose %.3f grid %.3f" % (d["value"], d["ms_per_step"], d["parity_sampled"].get("ok"), d["roofline"]["avg_launch_ms"], d["roofline_pose"]["avg_launch_ms"], d["roofline_grid_scan"]["avg_launch_ms"]))
PY
}
run off1 STVO_LINES_AHEAD=0
run on1 STVO_LINES_AHEAD=1
run gc1 STVO_LINES_AHEAD=2
run on2 STVO_LINES_AHEAD=1
run gc2 STVO_LINES_AHEAD=2
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/kt
STVO_LINES_AHEAD=2 timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -- python $R/bench.py $Q --no-parity > /dev/null 2>&1
python $R/tools/rocprof_summary.py timeline $(find /tmp/kt -name "*.db" | head -1) 20 -500 | cut -c1-130 > $OUT/timeline2.txt; cat $OUT/timeline2.txt
