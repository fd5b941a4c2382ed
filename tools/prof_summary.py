#!/usr/bin/env python
"""Condense a rocprofv3 *_kernel_stats.csv into the short per-kernel table kept under profiles/.

    python tools/prof_summary.py <kernel_stats.csv> <out.csv> "<title>" <steps>
"""
import csv
import os
import re
import sys


def short(n):
    """Demangled-ish short name: kernel<dtype,int template args>."""
    n = n.replace('(anonymous namespace)::', '')
    m = re.match(r'_ZN12_GLOBAL__N_1\d+([a-z0-9_]+?)I(DF16b|f)(.*)E+v', n)
    if m:
        ints = re.findall(r'Li(\d+)E', m.group(3))
        flags = re.findall(r'Lb([01])E', m.group(3))       # bool template arguments: shown only when one is set
        tail = ("+" + "".join(flags)) if "1" in flags else ""
        return f"{m.group(1)}<{'bf16' if m.group(2) == 'DF16b' else 'f32'}{''.join(',' + i for i in ints)}>{tail}"
    m = re.match(r'(?:void )?([a-z0-9_]+)<(__hip_bfloat16|float)((?:, \d+)*)(?:, (?:true|false))*>', n)
    if m:
        return f"{m.group(1)}<{'bf16' if m.group(2) != 'float' else 'f32'}{m.group(3).replace(' ', '')}>"
    return n.split('(')[0][:70]


def main():
    src, dst, title, steps = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
    rows = list(csv.DictReader(open(src)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    out = [f"# {title}", "kernel,calls,total_ms,avg_us,percent,ms_per_step"]
    for r in rows[:(400 if os.environ.get("PROF_ALL") else 32)]:      # PROF_ALL=1: every kernel, not the top 32
        t = float(r['TotalDurationNs'])
        out.append(f"{short(r['Name'])},{r['Calls']},{t/1e6:.3f},{float(r['AverageNs'])/1e3:.1f},"
                   f"{t/tot*100:.2f},{t/1e6/steps:.3f}")
    out.append(f"# total GPU kernel time {tot/1e6:.1f} ms over {steps:g} steps = {tot/1e6/steps:.2f} ms/step")
    open(dst, 'w').write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main()
