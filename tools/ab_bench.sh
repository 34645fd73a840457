#!/bin/bash
# A/B of library builds (tools/ab/<name>.so, git-ignored; select one with STVO_LIB) and developer switches on the bench headline:
#   tools/ab_bench.sh name[:VAR=VAL[,VAR=VAL...]] ...     (AB_ARGS: extra bench.py arguments, e.g. --lines 0)
# One line per variant: value, ms per step (median and every repeat), the parity sample (8 streams vs the oracle after the timed region).
for spec in "$@"; do
  lib=${spec%%:*}; envs=""
  if [[ "$spec" == *:* ]]; then envs=$(echo "${spec#*:}" | tr ',' ' '); fi
  echo "== $spec"
  env $envs STVO_LIB=$PWD/tools/ab/$lib.so timeout 300 python bench.py --no-extras --no-cpu-baseline --no-clocks ${AB_ARGS:-} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(round(d['value']), d['ms_per_step'], [round(x,4) for x in d['repeats']['ms_per_step_all']], 'parity', d['parity_sampled'].get('ok'), d['parity_sampled'].get('mismatches'))"
done
