#!/usr/bin/env python
"""Idle time between kernels of a rocprofv3 --kernel-trace run: tools/trace_gaps.py <kernel_trace.csv> [skip-first-fraction]

Prints, for the second half of the trace (steady state), the span, the sum of kernel durations, the union of busy
intervals (kernels of several streams overlap) and the histogram of the gaps between consecutive busy intervals."""
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
rows = rows[int(len(rows) * float(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2):]
span = rows[-1][1] - rows[0][0]
dur = sum(e - s for s, e, _ in rows)
busy, gaps, end = 0, [], rows[0][0]
for s, e, _ in rows:
    if s > end:
        gaps.append(s - end); busy += e - s; end = e
    elif e > end:
        busy += e - end; end = e
print(f"kernels {len(rows)}  span {span/1e6:.2f} ms  sum of durations {dur/1e6:.2f} ms  busy (union) {busy/1e6:.2f} ms  idle {(span-busy)/1e6:.2f} ms in {len(gaps)} gaps")
for lo, hi in ((0, 1), (1, 2), (2, 4), (4, 8), (8, 16), (16, 1e9)):
    g = [x for x in gaps if lo * 1e3 <= x < hi * 1e3]
    print(f"   gaps {lo:>2}-{hi if hi < 1e9 else 'inf':>3} us: {len(g):6d}  total {sum(g)/1e6:7.3f} ms")
