#!/usr/bin/env python
"""Timing and effective HBM bandwidth of the window-attention kernels on the four Swin stages (b64, 256^2 input).
Algorithmic bytes: forward reads q,k,v and writes out (4 tensors of tokens x C); backward reads q,k,v,dout and
writes dq,dk,dv (7 tensors)."""
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
    return e0.elapsed_time(e1) / n * 1e3


tot = [0.0, 0.0]
for hw, C, heads, blocks in [(64, 96, 3, 2), (32, 192, 6, 2), (16, 384, 12, 6), (8, 768, 24, 2)]:
    qkv = torch.randn(B, hw, hw, 3 * C, device="cuda").to(dt)
    dout = torch.randn(B, hw, hw, C, device="cuda").to(dt)
    table = torch.randn(225, heads, device="cuda") * 0.02
    for shift, mask in ((0, False), (4, True)):
        if hw == 8 and shift:
            continue
        tf = timeit(lambda: ops.window_attn_fwd(qkv, table, heads, shift, mask))
        tb = timeit(lambda: ops.window_attn_bwd(qkv, dout, table, heads, shift, mask))
        nb = B * hw * hw * C * 2
        flop_f = 4.0 * 64 * 64 * 32 * B * (hw // 8) ** 2 * heads
        print(f"{hw:3d}^2 C={C:4d} heads={heads:2d} shift={shift}  fwd {tf:7.1f} us {4*nb/tf/1e6:5.2f} TB/s {flop_f/tf/1e6:6.1f} TF/s"
              f" | bwd {tb:7.1f} us {7*nb/tb/1e6:5.2f} TB/s {2.5*flop_f/tb/1e6:6.1f} TF/s", flush=True)
        w = blocks / 2 if hw != 8 else blocks
        tot[0] += tf * w; tot[1] += tb * w
print(f"per step (depths 2-2-6-2): fwd {tot[0]/1e3:.3f} ms, bwd {tot[1]/1e3:.3f} ms")
