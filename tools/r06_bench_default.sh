#!/bin/bash
# the driver's command, timed, + the LSD evidence of round 6's second session.  gpurun --timeout 900 -- 'bash tools/r06_bench_default.sh'
R=$PWD; OUT=$R/gpurun_out/r06b; mkdir -p $OUT
S=$(date +%s); timeout 800 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc $? $(wc -c < $OUT/bench_default.json) bytes, $(( $(date +%s) - S )) s"
cp bench_extras.json $OUT/bench_extras_default.json
head -c 400 $OUT/bench_default.json; echo; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06b/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "roofline.frac", d["roofline"]["frac"]); print(json.dumps(d["legs"]))
PY
tail -3 $OUT/bench_default.err
