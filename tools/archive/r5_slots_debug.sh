#!/bin/bash
# per-piece read times of a one-turn stream with 2, 3, 4 and 8 slots (1, 2, 3, 7 loader threads)
mkdir -p gpurun_out/r5
for sl in 2 3 4 8; do
  DFDB_STREAM_DEBUG=1 timeout 200 python tools/r4_stream_timeline.py --rows 1e9 --slots $sl --readers 1 > /dev/null 2> gpurun_out/r5/slots_$sl.err
  echo "slots $sl: $(grep seconds gpurun_out/r5/slots_$sl.err | tail -1)"
done
