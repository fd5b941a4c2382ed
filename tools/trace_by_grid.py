#!/usr/bin/env python
"""Per-dispatch view of a rocprofv3 kernel trace: groups the dispatches of kernels whose name matches a
pattern by (kernel, grid size) and prints count, average and total duration.

    python tools/trace_by_grid.py <dir with *_kernel_trace.csv> <name regex> [skip_first_n_steps_fraction]
"""
import collections
import csv
import glob
import re
import sys


def main():
    d, pat = sys.argv[1], re.compile(sys.argv[2])
    files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    rows = []
    for f in files:
        rows += list(csv.DictReader(open(f)))
    groups = collections.defaultdict(list)
    for r in rows:
        name = r.get("Kernel_Name") or r.get("Name")
        if not pat.search(name):
            continue
        grid = r.get("Grid_Size_X") or r.get("Grid_Size")
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        short = re.sub(r"\(anonymous namespace\)::", "", name)[:60]
        groups[(short, int(grid))].append(dur)
    tot = 0.0
    for (name, grid), ds in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        ds_sorted = sorted(ds)
        med = ds_sorted[len(ds) // 2]
        print(f"{name:60s} grid {grid:9d} n {len(ds):5d} med {med:8.1f} us avg {sum(ds)/len(ds):8.1f} us total {sum(ds)/1e3:8.3f} ms")
        tot += sum(ds)
    print(f"total {tot/1e3:.3f} ms")


if __name__ == "__main__":
    main()
