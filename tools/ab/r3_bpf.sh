#!/bin/bash
# interleaved A/B of the B-fragment prefetch (DMEL_BPF), one device
for round in 1 2; do
  for lib in dmel_codec_amd/libdmel_hip.so tools/ab/bpf1.so; do
    echo "== $lib (round $round)"
    DMEL_LIB=$PWD/$lib timeout -k 10 200 python tools/bench_conv.py --iters 10 --precision 3 --check 2>/dev/null
  done
done
for lib in dmel_codec_amd/libdmel_hip.so tools/ab/bpf1.so; do
  echo "== $lib six-product split"
  DMEL_LIB=$PWD/$lib timeout -k 10 200 python tools/bench_conv.py --iters 10 --precision 0 --check --only wn 2>/dev/null
done
