#!/bin/bash
# EuRoC-shaped batches (line-heavy) and mid batches with / without the key-line stage ahead; then the whole GPU suite
for v in 0 1 0 1; do
  echo "LINES_AHEAD=$v euroc 512: $(STVO_LINES_AHEAD=$v python tools/bench_pipeline.py --preset euroc --mode 0 --batch 512 --points 660 --lines 250 --cpu-frames 0 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in d if k in ("value","ms_per_step","frame_pairs_per_s")})')"
done
for v in 0 1; do
  echo "LINES_AHEAD=$v kitti 256: $(STVO_LINES_AHEAD=$v python tools/bench_pipeline.py --batch 256 --cpu-frames 0 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in d if k in ("value","ms_per_step","frame_pairs_per_s")})')"
done
timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -3
