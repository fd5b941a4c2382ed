#!/usr/bin/env python
"""Per-kernel roofline table from three separate rocprofv3 --pmc passes of the same command.

    python tools/roofline_table.py <fetch.csv> <write.csv> <mfma.csv> <out.csv> "<title>"

HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KiB (gfx950 tallies the 128-byte requests of wide
reads at 64 bytes); achieved TB/s = bytes / the kernel's average duration in the FETCH_SIZE pass,
against 8 TB/s.  MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs):
the share of SIMD-cycles in which the matrix pipe is busy, at the clock the chip actually held.
clock_GHz = GRBM_GUI_ACTIVE / 8 / the kernel's duration in that pass: the clock the chip held under this
kernel's load (it reads high on launches much shorter than 0.3 ms); the dense peaks assume 2.4 GHz.
"""
import csv, sys, collections
sys.path.insert(0, __file__.rsplit("/", 1)[0])
from prof_summary import short


def load(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(lambda: collections.Counter())
    dur = collections.defaultdict(float)
    seen = set()
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k][r["Counter_Name"]] += 1
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return acc, n, dur


fe, nf, df = load(sys.argv[1])
wr, nw, _ = load(sys.argv[2])
mf, nm, dm = load(sys.argv[3])
rows = []
for k in fe:
    c = nf[k]["FETCH_SIZE"]
    if not c or not nw[k]["WRITE_SIZE"]:
        continue
    rd = 2.0 * fe[k]["FETCH_SIZE"] / c * 1024
    w = wr[k]["WRITE_SIZE"] / nw[k]["WRITE_SIZE"] * 1024
    us = df[k] / c / 1e3
    util = clk = None
    if nm[k]["GRBM_GUI_ACTIVE"] and mf[k]["GRBM_GUI_ACTIVE"] > 0:
        util = mf[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / (mf[k]["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        if dm[k] > 0:
            clk = mf[k]["GRBM_GUI_ACTIVE"] / 8.0 / dm[k]          # cycles per ns = GHz (sums over the same launches)
    rows.append((df[k], k, c, us, rd, w, (rd + w) / (us * 1e-6) / 1e12, util, clk))
rows.sort(reverse=True)
out = [f"# {sys.argv[5]}", "kernel,launches,avg_us,hbm_read_MB_per_launch,hbm_write_MB_per_launch,achieved_TB_per_s,frac_of_8TBps,mfma_util,clock_GHz"]
for _, k, c, us, rd, w, tb, util, clk in rows[:28]:
    out.append(f"{k},{c},{us:.1f},{rd/1e6:.1f},{w/1e6:.1f},{tb:.2f},{tb/8:.3f},{'' if util is None else f'{util:.3f}'},"
               f"{'' if clk is None else f'{clk:.2f}'}")
open(sys.argv[4], "w").write("\n".join(out) + "\n")
print("\n".join(out))
