#!/bin/bash
# Same-box A/B of library builds: tools/ab_run.sh <workload> <rounds> <lib or "base"> ...   (prints ms_per_step per run)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
w=$1; rounds=$2; shift 2
for r in $(seq 1 $rounds); do
  for lib in "$@"; do
    if [ "$lib" = base ]; then unset UNETK_AB_LIB; else export UNETK_AB_LIB=$lib; fi
    out=$(timeout -k 10 300 python tools/ab_bench.py --workload $w --no-cpu-baseline --no-small-batch --no-e2e --no-roofline 2>/dev/null | tail -1)
    echo "$w round $r $lib: $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
  done
done
