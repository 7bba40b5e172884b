#!/bin/bash
# interleaved A/B of the x-load placement (DMEL_XTOP) and the weight prefetch distance of the fp16-split kernel, one device
for round in 1 2; do
  for lib in dmel_codec_amd/libdmel_hip.so tools/ab/xtop0.so tools/ab/xtop1_pd3.so; do
    echo "== $lib (round $round)"
    DMEL_LIB=$PWD/$lib timeout -k 10 200 python tools/bench_conv.py --iters 10 --precision 3 --check 2>/dev/null
  done
done
