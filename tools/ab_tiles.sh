#!/bin/bash
# sweep conv tile ids on one device: tools/ab_tiles.sh "<shapes regex>" id id ...
PAT=$1; shift
for t in "$@"; do
  echo "== tile $t"
  DMEL_CONV_TILE=$t python tools/bench_conv.py --iters 10 2>/dev/null | grep -E "$PAT"
done
