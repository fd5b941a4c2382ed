#!/usr/bin/env python
"""Sum the counters of a rocprofv3 --pmc run per kernel: tools/pmc_kernel.py DIR [name-substring]."""
import csv, glob, sys, collections, re
d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:60]
        if sub and sub not in k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            calls[k] += 1
for k, c in acc.items():
    print(k, "calls", calls[k])
    for n, v in sorted(c.items()):
        print(f"   {n:32s} {v / calls[k]:16.0f} per call")
