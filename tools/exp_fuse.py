#!/usr/bin/env python
"""Timing of the eval-mode fused epilogues of the 48 -> 48 conv at the predictor's batch (21 tiles of 256^2)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_amd import ops, _lib
if os.environ.get("UNETK_AB_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["UNETK_AB_LIB"])
dt = torch.bfloat16


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B in (21, 64):
    x = torch.randn(B, 256, 256, 48, device="cuda").to(dt)
    w = torch.randn(48, 48, 3, 3, device="cuda") * 0.05
    pw = ops.PackedConv(w, dt)
    sc, sh = torch.rand(48, device="cuda") + 0.5, torch.randn(48, device="cuda")
    cat = torch.empty(B, 256, 256, 144, device="cuda", dtype=dt)
    hw, hb = torch.randn(1, 48, 1, 1, device="cuda"), torch.randn(1, device="cuda")
    print(f"B={B}: plain {timeit(lambda: ops.conv_fwd(x, pw)):.0f} us, affine+relu {timeit(lambda: ops.conv_fwd(x, pw, scale=sc, shift=sh, act=1)):.0f} us, "
          f"into concat slice {timeit(lambda: ops.conv_fwd(x, pw, scale=sc, shift=sh, act=1, out=cat[..., 96:])):.0f} us, "
          f"+pool {timeit(lambda: ops.conv_fwd(x, pw, scale=sc, shift=sh, act=1, out=cat[..., 96:], pool=True)):.0f} us, "
          f"head only {timeit(lambda: ops.conv_fwd(x, pw, scale=sc, shift=sh, act=1, head=(hw, hb), write_out=False)):.0f} us", flush=True)
