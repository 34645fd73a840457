#!/bin/bash
# Single-stream latency of the per-frame pipeline (unprofiled): handler API (both modes) vs stvo_seq_push, points only and points + lines.
#   tools/latency.sh <outfile>
R=$PWD; OUT=$R/$1; mkdir -p $(dirname $OUT)
python tools/make_sequence.py /tmp/seq_p.bin --frames 51 > /dev/null
python tools/make_sequence.py /tmp/seq_l.bin --frames 51 --lines 100 > /dev/null
{
for s in p l; do
  echo "# $([ $s = p ] && echo 'points only' || echo 'points + ~100 key-lines per image'): handler (default: on the device pipeline) | handler, one synchronous call per stage | stvo_seq_push directly"
  $R/stvo-pl_amd/bin/imagesStVO_synth /tmp/seq_$s.bin /tmp/res_h$s.bin --preset kitti | tail -1
  STVO_HANDLER_PIPELINE=0 $R/stvo-pl_amd/bin/imagesStVO_synth /tmp/seq_$s.bin /tmp/res_g$s.bin --preset kitti | tail -1
  $R/stvo-pl_amd/bin/imagesStVO_synth /tmp/seq_$s.bin /tmp/res_d$s.bin --preset kitti --device-pipeline | tail -1
done
} > $OUT 2>&1
cat $OUT
