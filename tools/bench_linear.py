#!/usr/bin/env python
"""Timing of the 1x1 (linear-layer) path on the Swin-UNet shapes at batch 64: forward, data gradient, weight gradient."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_amd import ops, _lib
if os.environ.get("UNETK_AB_LIB"):      # A/B a kernel variant built to another file
    _lib.LIB_PATH = os.path.abspath(os.environ["UNETK_AB_LIB"])

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dt = torch.bfloat16


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


tot = [0, 0, 0, 0, 0, 0]
for side, C in ((64, 96), (32, 192), (16, 384), (8, 768)):
    for name, ci, co in (("qkv", C, 3 * C), ("proj", C, C), ("fc1", C, 4 * C), ("fc2", 4 * C, C)):
        x = torch.randn(B, side, side, ci, device="cuda").to(dt)
        dy = torch.randn(B, side, side, co, device="cuda").to(dt)
        w = torch.randn(co, ci, 1, 1, device="cuda") * 0.05
        pw, pwt = ops.PackedConv(w, dt), ops.PackedConv(w, dt, transpose=True)
        ws = torch.empty(ops._lib.load().unetk_conv_wgrad_workspace_bytes(B, side, side, ci, co, 1, 1), dtype=torch.uint8, device="cuda")
        tf = timeit(lambda: ops.conv_fwd(x, pw))
        td = timeit(lambda: ops.conv_fwd(dy, pwt))
        gf = timeit(lambda: ops.linear_fwd(x, pw)) if ops.linear_supported(x, pw) else float("nan")
        gd = timeit(lambda: ops.linear_fwd(dy, pwt)) if ops.linear_supported(dy, pwt) else float("nan")
        gg = timeit(lambda: ops.linear_fwd(x, pw, act=2, out2=True)) if ops.linear_supported(x, pw) else float("nan")
        ga = timeit(lambda: ops.linear_fwd(dy, pwt, gelu_grad_of=x)) if ops.linear_supported(dy, pwt) else float("nan")
        tot[3] += gf; tot[4] += gd
        tw = timeit(lambda: ops.conv_wgrad(x, dy, ci, co, 1, workspace=ws))
        tg = timeit(lambda: ops.token_wgrad(x, dy, ci, co))
        M = B * side * side
        fl = 2.0 * M * ci * co
        by = 2.0 * M * (ci + co)
        tot[0] += tf; tot[1] += td; tot[2] += tw; tot[5] += tg
        print(f"{side:3d}^2 {name:5s} {ci:5d}->{co:5d}  GEMM fwd {gf*1e3:6.1f} us {fl/gf/1e9:5.0f} TF/s {by/gf/1e9:5.2f} TB/s dgrad {gd*1e3:6.1f} | gelu2 {gg*1e3:6.1f} gelu' {ga*1e3:6.1f} || conv1x1 fwd {tf*1e3:7.1f} us {fl/tf/1e9:6.0f} TF/s {by/tf/1e9:5.2f} TB/s | "
              f"dgrad {td*1e3:7.1f} us {fl/td/1e9:6.0f} TF/s | wgrad {tw*1e3:7.1f} us {fl/tw/1e9:6.0f} TF/s {by/tw/1e9:5.2f} TB/s | NEW wgrad {tg*1e3:7.1f} us {fl/tg/1e9:6.0f} TF/s ", flush=True)
print("token_wgrad sum ms", round(tot[5], 3))
print("sum ms", [round(t, 3) for t in tot])
