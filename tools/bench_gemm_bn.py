#!/usr/bin/env python
"""Token GEMM (96-channel blocks) on the stage-2/3 shapes of the Swin encoder at batch 64, checked against torch:
tools/bench_gemm_bn.py   (UNETK_AB_LIB=<other build> to time a variant)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_amd import ops, _lib
from unet_amd._lib import call, ptr, dtype_code, stream
if os.environ.get("UNETK_AB_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["UNETK_AB_LIB"])
B, dt = 64, torch.bfloat16


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def pack(w, bn):
    Co, Ci = w.shape
    n = _lib.load().unetk_conv_packed_elems(Co, Ci, 1, bn, 1)
    d = torch.empty(n, dtype=dt, device="cuda")
    call("unetk_conv_pack_weight", ptr(w), ptr(d), Co, Ci, 1, bn, 1, 0, stream())
    return d


def run(x, d, N, bn, out):
    M, K = x.shape[0] * x.shape[1] * x.shape[2], x.shape[3]
    call("unetk_linear_fwd", ptr(x), K, ptr(d), ptr(out), N, None, 0, M, K, N, bn, 1, None, 0, None, 0, None, 1, None, 0, stream())


for side, C in ((16, 384), (8, 768)):
    for name, ci, co in (("qkv", C, 3 * C), ("proj", C, C), ("fc1", C, 4 * C), ("fc2", 4 * C, C)):
        x = torch.randn(B, side, side, ci, device="cuda").to(dt)
        w = (torch.randn(co, ci, device="cuda") * 0.05).contiguous()
        outs = {}
        ts = {}
        for bn in (96,):
            d = pack(w, bn)
            out = torch.empty(B, side, side, co, dtype=dt, device="cuda")
            ts[bn] = timeit(lambda: run(x, d, co, bn, out))
            outs[bn] = out
        ref = x.float().view(-1, ci) @ w.to(dt).float().t()
        err = ((outs[96].float().view(-1, co) - ref).norm() / ref.norm()).item()
        print(f"{side}^2 {name} {ci}->{co}: {ts[96]:.1f} us  rel err vs torch {err:.1e}", flush=True)
