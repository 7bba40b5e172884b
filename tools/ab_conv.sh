#!/bin/bash
# interleaved A/B of conv kernel builds on ONE device: tools/ab_conv.sh "<shapes regex>" lib1.so lib2.so ...
PAT=$1; shift
for round in 1 2; do
  for lib in "$@"; do
    echo "== $lib (round $round)"
    DMEL_LIB=$PWD/$lib python tools/bench_conv.py --iters 10 2>/dev/null | grep -E "$PAT"
  done
done
