#!/usr/bin/env python
"""ops.linear_wgrad (weight + bias gradient of a token linear) on the Swin shapes at batch 64: time, TB/s of operand bytes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_amd import ops, _lib
if os.environ.get("UNETK_AB_LIB"):      # A/B a kernel variant built to another file
    _lib.LIB_PATH = os.path.abspath(os.environ["UNETK_AB_LIB"])

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


tot = 0.0
for side, C in ((64, 96), (32, 192), (16, 384), (8, 768)):
    for name, ci, co in (("qkv", C, 3 * C), ("proj", C, C), ("fc1", C, 4 * C), ("fc2", 4 * C, C)):
        M = B * side * side
        x = torch.randn(M, ci, device="cuda").to(torch.bfloat16)
        dy = torch.randn(M, co, device="cuda").to(torch.bfloat16)
        t = timeit(lambda: ops.linear_wgrad(x, dy, ci, co))
        tot += t
        print(f"{side:3d}^2 {name:5s} {ci:5d}->{co:5d}  {t*1e3:7.1f} us  {2.0*M*(ci+co)/t/1e9:5.2f} TB/s  {2.0*M*ci*co/t/1e9:6.0f} TF/s", flush=True)
print(f"sum {tot:.3f} ms")
