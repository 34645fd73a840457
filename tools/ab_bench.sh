#!/bin/bash
# A/B of library builds (tools/ab/<name>.so, git-ignored; select one with STVO_LIB) and developer switches on the bench headline:
#   tools/ab_bench.sh name[:VAR=VAL[,VAR=VAL...]] ...     (AB_ARGS: extra bench.py arguments, e.g. --lines 0)
# One line per variant: value (median of the repeats), min / max of the repeats, ms per step, the three kernels' light-pass times, the parity sample.
for spec in "$@"; do
  lib=${spec%%:*}; envs=""
  if [[ "$spec" == *:* ]]; then envs=$(echo "${spec#*:}" | tr ',' ' '); fi
  echo "== $spec"
  env $envs STVO_LIB=$PWD/tools/ab/$lib.so timeout 300 python bench.py --no-extras --no-cpu-baseline --no-clocks ${AB_ARGS:-} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['repeats']
print(round(d['value']), r['value_min'], r['value_max'], d['ms_per_step'], 'K1m', d['roofline']['avg_launch_ms'], 'pose', d['roofline_pose']['avg_launch_ms'], 'grid', d['roofline_grid_scan']['avg_launch_ms'], 'parity', d['parity_sampled'].get('ok'))"
done
