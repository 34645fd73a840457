#!/bin/bash
# BASELINE configs[3] (EuRoC-shaped 752x480, ~800 key-points over 4 octaves, ~300 key-lines, 40 % point outliers; optimiser modes
# 0 GN, 1 robust GN, 2 LM): single-stream latency through the handler mirror and stvo_seq_push, full-pipeline throughput, oracle time.
#   tools/config4.sh <outfile>
R=$PWD; OUT=$R/$1; mkdir -p $(dirname $OUT)
python tools/make_sequence.py /tmp/seq_e.bin --frames 41 --points 660 --lines 250 --cam euroc --config4 > /dev/null
{
for m in 0 1 2; do
  echo "# optimiser mode $m: handler (device pipeline) | stvo_seq_push directly"
  $R/stvo-pl_amd/bin/imagesStVO_synth /tmp/seq_e.bin /tmp/res_e.bin --preset euroc --mode $m | tail -1
  $R/stvo-pl_amd/bin/imagesStVO_synth /tmp/seq_e.bin /tmp/res_e.bin --preset euroc --mode $m --device-pipeline | tail -1
  python tools/bench_pipeline.py --preset euroc --mode $m --batch 512 --points 660 --lines 250 --cpu-frames 6 2>/dev/null
done
} > $OUT 2>&1
cat $OUT
