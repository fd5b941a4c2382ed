#!/usr/bin/env python
"""Register / LDS / spill usage per kernel of one HIP source of unet_amd/csrc.
   python tools/kernel_regs.py conv_wgrad.hip [substring of the demangled kernel name] [extra hipcc flags]"""
import os, re, subprocess, sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = sys.argv[3:]
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "unet_amd", "csrc")
cmd = ["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-result", "-c", src, "-o",
       "/tmp/kernel_regs.o", "-Rpass-analysis=kernel-resource-usage"] + extra
out = subprocess.run(cmd, cwd=csrc, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: +Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark: +([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = name.replace("(anonymous namespace)::", "")
    if flt and flt not in name:
        continue
    g = lambda k: r.get(k, 0)
    print("%-88s VGPR %3d AGPR %3d spill %3d scratch %4d LDS %6d occ %d" %
          (name[:88], g("VGPRs"), g("AGPRs"), g("VGPRs Spill"), g("ScratchSize"), g("LDS Size"), g("Occupancy")))
