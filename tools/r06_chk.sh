#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_seq.py -x -q -k "back_to_back" 2>&1 | tail -5
echo "default euroc 512: $(python tools/bench_pipeline.py --preset euroc --mode 0 --batch 512 --points 660 --lines 250 --cpu-frames 0 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')"
for v in 0 x; do
  if [ $v = x ]; then unset STVO_LINES_AHEAD; else export STVO_LINES_AHEAD=$v; fi
  python bench.py --no-cpu-baseline --no-extras --no-clocks --repeats 5 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("LINES_AHEAD='$v'", d["value"], d["ms_per_step"], d["parity_sampled"])'
done
