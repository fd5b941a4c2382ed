#!/bin/bash
# usage: ab_env.sh <workload> <rounds> VAR   -> alternates VAR=0 / default
w=$1; rounds=$2; var=$3
for r in $(seq 1 $rounds); do
  for v in 0 1; do
    out=$(env $var=$v timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --no-small-batch --no-e2e --no-roofline 2>/dev/null | tail -1)
    echo "$w round $r $var=$v: $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
  done
done
