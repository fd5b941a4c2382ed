#!/usr/bin/env python
"""One conv launch of a given shape (for in-kernel timing builds): exp_one.py HW CI CO [stats]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_amd import ops, _lib
if os.environ.get("UNETK_AB_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["UNETK_AB_LIB"])
hw, ci, co = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
stats = len(sys.argv) > 4
x = torch.randn(64, hw, hw, ci, device="cuda").to(torch.bfloat16)
w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
pw = ops.PackedConv(w, torch.bfloat16)
for _ in range(2):
    ops.conv_fwd(x, pw, want_stats=stats)
    torch.cuda.synchronize()
