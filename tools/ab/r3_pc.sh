#!/bin/bash
for round in 1 2; do
  for lib in "$@"; do
    echo "== $lib (round $round)"
    DMEL_LIB=$PWD/$lib timeout -k 10 200 python tools/bench_conv.py --iters 10 --precision 3 2>/dev/null | grep -E "bv1|wn_dec|conv_pre"
  done
done
