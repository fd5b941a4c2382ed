#!/usr/bin/env python
"""Per-kernel averages of the counters in a rocprofv3 --pmc counter_collection.csv.

    python tools/pmc_table.py <counter_collection.csv> [kernel substring]
"""
import csv, sys, collections
sys.path.insert(0, __file__.rsplit("/", 1)[0])
from prof_summary import short

acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.Counter())
dur = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    k = short(r["Kernel_Name"]) + f" grid={r['Grid_Size']}"
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k][r["Counter_Name"]] += 1
    dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k in acc:
    n = max(cnt[k].values())
    ncounters = len(acc[k])
    line = f"{k} launches={n} dur_us={dur[k]/n/ncounters/1e3:.1f} "
    line += " ".join(f"{c}={v/cnt[k][c]:.4g}" for c, v in sorted(acc[k].items()))
    print(line)
