R=$PWD; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/tools/bench_pipeline.py --preset euroc --batch 512 --points 800 --lines 250 --steps 8 --warmup 4 --cpu-frames 0 > $R/gpurun_out/r04_euroc_bench.json 2>/dev/null
DB=$(find /tmp/kt -name "*.db" | head -1); cd $R
python tools/rocprof_summary.py split $DB | cut -c1-160 | head -24
python tools/rocprof_summary.py timeline $DB 24 | cut -c1-150
cat gpurun_out/r04_euroc_bench.json | cut -c1-400
