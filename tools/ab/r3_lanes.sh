#!/bin/bash
# lanes x vocoder-streams sweep of bench.py, two interleaved rounds
for round in 1 2; do
  for lanes in 1 2 3 4; do
    for st in 1 3; do
      echo -n "round $round lanes $lanes streams $st: "
      timeout -k 10 120 python bench.py --steps 20 --warmup 5 --lanes $lanes --streams $st --cpu-budget 0 --median-steps 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
    done
  done
done
