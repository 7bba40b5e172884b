#!/bin/bash
# interleaved A/B of STFT kernel variants (waves per workgroup, tables in registers or from L1)
for round in 1 2; do
  for lib in dmel_codec_amd/libdmel_hip.so tools/ab/stft_f64.so tools/ab/stft_f16.so tools/ab/stft_f64w8.so; do
    echo "== $lib (round $round)"
    DMEL_LIB=$PWD/$lib timeout -k 10 100 python tools/bench_stft.py 2>/dev/null | tail -2
  done
done
