#!/bin/bash
# Kernel-level trace of the single-stream device pipeline (one sequence, one frame at a time).
#   tools/trace_latency.sh <outdir> [lines]
set -e
R=$PWD; OUT=$R/$1; L=${2:-0}; mkdir -p $OUT
python tools/make_sequence.py /tmp/seq.bin --frames 31 --lines $L > /dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/lat_kt -o lat -- $R/stvo-pl_amd/bin/imagesStVO_synth /tmp/seq.bin /tmp/res.bin --preset kitti ${TRACE_MODE---device-pipeline} > $OUT/run_l$L.txt 2>&1 || true
cd $R
DB=$(find /tmp/lat_kt -name "*.db" | head -1)
python - "$DB" > $OUT/trace_l$L.txt <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("kernels")][0] if any(t.startswith("kernels") for t in tabs) else None
rows = con.execute("select name, start, end from kernels order by start").fetchall()
# last frame = last occurrence of pose kernel; print the dispatches between the previous pose kernel and it
idx = [i for i, r in enumerate(rows) if "pose_kernel" in r[0]]
a, b = idx[-2] + 1, idx[-1]
t0 = rows[a][1]
prev_end = None
for name, s, e in rows[a:b + 1]:
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{(s - t0) / 1e3:9.1f} us  +{gap:6.1f} gap  {(e - s) / 1e3:8.1f} us  {name[:70]}")
    prev_end = e
print(f"GPU span of one frame: {(rows[b][2] - t0) / 1e3:.1f} us; frame-to-frame period: {(rows[idx[-1]][1] - rows[idx[-2]][1]) / 1e3:.1f} us")
PY
rm -rf /tmp/lat_kt
tail -3 $OUT/run_l$L.txt; cat $OUT/trace_l$L.txt
