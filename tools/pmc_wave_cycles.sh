#!/bin/bash
# wave-cycle decomposition of the bench step's kernels, one rocprofv3 --pmc pass:  tools/pmc_wave_cycles.sh "<counters>" [grep pattern]
R=$PWD; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pmc_x
timeout ${PMC_LEASH:-90} rocprofv3 --pmc $1 --kernel-trace -d /tmp/pmc_x -- python $R/bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 --repeats 1 --no-parity --no-clocks > /dev/null 2>&1; echo "exit $?"
cd $R; python tools/rocprof_summary.py pmc $(find /tmp/pmc_x -name "*.db" | head -1) 2>/dev/null | grep -i "${2:-hamming_knn2_mfma_kernel\|counter}" | cut -c1-150; rm -rf /tmp/pmc_x
