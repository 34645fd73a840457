#!/bin/bash
# dispatches of the TIMED region of bench.py (steps back to back, no markers) under rocprofv3 --kernel-trace:  tools/timed_timeline.sh <lib> [skip]
R=$PWD; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/kt
STVO_LIB=$R/tools/ab/$1.so timeout 200 rocprofv3 --kernel-trace -d /tmp/kt -- python $R/bench.py --no-cpu-baseline --no-extras --no-parity --no-clocks --repeats 1 > /tmp/b.json 2>/dev/null
cd $R; python tools/rocprof_summary.py timeline $(find /tmp/kt -name "*.db" | head -1) 48 ${2:-330} | cut -c1-110
