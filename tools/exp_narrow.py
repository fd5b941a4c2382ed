#!/usr/bin/env python
"""Timing experiments on the full-resolution narrow conv layers (what bounds the 48-channel kernels)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_amd import ops, _lib
if os.environ.get("UNETK_AB_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["UNETK_AB_LIB"])
dt = torch.bfloat16
B = 64


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for hw, ci, co in [(256, 48, 48), (256, 32, 48), (256, 64, 48), (256, 96, 48), (256, 144, 48), (256, 48, 32), (256, 48, 96),
                   (256, 32, 32), (256, 64, 64), (128, 96, 96), (128, 48, 96), (128, 96, 48)]:
    x = torch.randn(B, hw, hw, ci, device="cuda").to(dt)
    w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    pw = ops.PackedConv(w, dt)
    mb = B * hw * hw * (ci + co) * 2 / 1e6
    fl = 2.0 * B * hw * hw * ci * co * 9
    t_s = timeit(lambda: ops.conv_fwd(x, pw, want_stats=True))
    t_n = timeit(lambda: ops.conv_fwd(x, pw))
    print(f"{hw}^2 {ci:4d}->{co:3d} bn={pw.bn}: stats {t_s*1e3:6.0f} us, plain {t_n*1e3:6.0f} us = {mb/t_n/1e6:5.2f} TB/s, "
          f"{fl/t_n/1e9:5.0f} TF/s ({mb:.0f} MB)", flush=True)
