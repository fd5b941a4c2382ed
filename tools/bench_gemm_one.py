#!/usr/bin/env python
"""One token GEMM shape, many launches (for counter passes): tools/bench_gemm_one.py SIDE CI CO [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_amd import ops, _lib
if os.environ.get("UNETK_AB_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["UNETK_AB_LIB"])
side, ci, co = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 64
x = torch.randn(B, side, side, ci, device="cuda").to(torch.bfloat16)
w = torch.randn(co, ci, 1, 1, device="cuda") * 0.05
pw = ops.PackedConv(w, torch.bfloat16)
for _ in range(3):
    ops.linear_fwd(x, pw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.linear_fwd(x, pw)
e1.record()
torch.cuda.synchronize()
print(f"{side}^2 {ci}->{co}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
