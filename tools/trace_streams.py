#!/usr/bin/env python
"""Per-stream summary of a compact kernel trace (tools/trace_step.sh): for the last step in the file, the busy time
of each stream, the wall time of the step and the kernels of the busiest stream by total duration.

    python tools/trace_streams.py gpurun_out/<tag>_<w>/trace.csv [steps in the file]
"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
for r in rows:
    r["start"], r["end"] = int(r["start"]), int(r["end"])
rows.sort(key=lambda r: r["start"])
# the step boundary: the optimizer's update kernel closes a step
ends = [i for i, r in enumerate(rows) if "adamw_kernel" in r["kernel"]]
if len(ends) < 2:
    sys.exit("no step boundaries found")
lo, hi = ends[-2] + 1, ends[-1] + 1
step = rows[lo:hi]
wall = (step[-1]["end"] - step[0]["start"]) / 1e6
by_stream = collections.defaultdict(list)
for r in step:
    by_stream[(r["queue"], r["stream"])].append(r)
print(f"last step: {len(step)} launches, wall {wall:.3f} ms")
for key, rs in sorted(by_stream.items(), key=lambda kv: -sum(r['end'] - r['start'] for r in kv[1])):
    busy = sum(r["end"] - r["start"] for r in rs) / 1e6
    print(f"  stream {key}: {len(rs)} launches, busy {busy:.3f} ms")
main = max(by_stream.values(), key=lambda rs: sum(r["end"] - r["start"] for r in rs))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in main:
    k = re.sub(r"\(anonymous namespace\)::", "", r["kernel"])[:48]
    agg[k][0] += 1
    agg[k][1] += (r["end"] - r["start"]) / 1e6
gaps = sum(max(0, b["start"] - a["end"]) for a, b in zip(main, main[1:])) / 1e6
print(f"busiest stream: idle between its launches {gaps:.3f} ms")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"  {t:7.3f} ms {n:4d} x {t / n * 1e3:7.1f} us  {k}")
