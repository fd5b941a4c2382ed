#!/usr/bin/env python
"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, separate runs).

    python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [<mfma counter_collection.csv>]

With the fourth file (a pass of SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE) every kernel also gets
"mfma_util" = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs), as tools/roofline_table.py.

Both counters are in KiB.  On gfx950 FETCH_SIZE tallies the 128-byte requests of wide coalesced
reads at 64 bytes, so the read side is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores.
"""
import csv, json, sys
from collections import defaultdict
sys.path.insert(0, __file__.rsplit("/", 1)[0])
from prof_summary import short


def per_kernel(path, counter):
    tot, n = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        tot[k] += float(r["Counter_Value"])
        n[k] += 1
    return tot, n


fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
write, nw = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(fetch, key=lambda k: -fetch[k]):
    if nf[k] == 0 or nw.get(k, 0) == 0:
        continue
    rd = 2.0 * fetch[k] / nf[k] * 1024
    wr = write[k] / nw[k] * 1024
    out[k] = {"launches_sampled": nf[k], "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
              "hbm_bytes_per_launch": round(rd + wr)}
# Instantiations that differ only in bool template flags ("name<...>+01") are also summed under the bare
# name when there are several of them: bench.py prices a kernel family as one (all its launches per step).
fam = defaultdict(list)
for k in out:
    fam[k.split("+")[0]].append(k)
for base, ks in fam.items():
    if len(ks) > 1:
        n = sum(out[k]["launches_sampled"] for k in ks)
        rd = sum(out[k]["read_bytes_per_launch"] * out[k]["launches_sampled"] for k in ks) / n
        wr = sum(out[k]["write_bytes_per_launch"] * out[k]["launches_sampled"] for k in ks) / n
        if base in out:
            out[base + "+plain"] = out[base]
        out[base] = {"launches_sampled": n, "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
                     "hbm_bytes_per_launch": round(rd + wr), "variants": sorted(k for k in ks if k != base) + [base + "+plain"]}
if len(sys.argv) > 4:
    busy, _ = per_kernel(sys.argv[4], "SQ_VALU_MFMA_BUSY_CYCLES")
    act, _ = per_kernel(sys.argv[4], "GRBM_GUI_ACTIVE")
    for k in out:
        base = k.split("+")[0]
        ks = [k] if k in act else [x for x in act if x.split("+")[0] == base]
        a = sum(act[x] for x in ks)
        if a > 0:
            out[k]["mfma_util"] = round(sum(busy[x] for x in ks) / (a / 8.0 * 1024.0), 4)
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in list(out.items())[:12]:
    print(k, v)
