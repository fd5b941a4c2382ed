#!/usr/bin/env python
"""The one-launch MLP branch against the four launches it replaces, at the batch-64 shapes of Swin stages 0 and 1.
Operands are rotated through a pool larger than the Infinity Cache, so that every call reads cold tensors as in the step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_amd import ops

dt = torch.bfloat16
for C, side in ((96, 64), (192, 32)):
    B = 64
    NP = 6
    xs = [torch.randn(B, side, side, C, device="cuda").to(dt) for _ in range(NP)]
    gamma, beta = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1
    w1, b1 = torch.randn(4 * C, C, device="cuda") * C ** -0.5, torch.randn(4 * C, device="cuda") * 0.1
    w2, b2 = torch.randn(C, 4 * C, device="cuda") * (4 * C) ** -0.5, torch.randn(C, device="cuda") * 0.1
    keep = (torch.rand(B, device="cuda") > 0.1).float() / 0.9
    p1, p2 = ops.pack_many([(w1, "bn64"), (w2, False)], dt)
    pw1, pw2 = ops.PackedConv(w1, dt), ops.PackedConv(w2, dt)

    def sep(x):
        ln, m, r = ops.layernorm_fwd(x, gamma, beta, 1e-5)
        h = ops.linear_fwd(ln, pw1, bias=b1)
        a = ops.gelu_fwd(h)
        return ops.linear_fwd(a, pw2, bias=b2, resid=x, resid_keep=keep)

    def timeit(fn, n=12):
        for i in range(3):
            fn(xs[i % NP])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            fn(xs[i % NP])
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    M = B * side * side
    t_sep = timeit(sep)
    t_f = timeit(lambda x: ops.swin_mlp_fwd(x, gamma, beta, 1e-5, p1, b1, p2, b2, keep=keep))
    t_fn = timeit(lambda x: ops.swin_mlp_fwd(x, gamma, beta, 1e-5, p1, b1, p2, b2, keep=keep, want_act=False))
    by_f = M * (C * 2 * 4 + 4 * C * 2 * 2)
    by_fn = M * (C * 2 * 4 + 4 * C * 2)
    print(f"C={C} tokens {M}: separate {t_sep:7.1f} us | fused {t_f:7.1f} us ({by_f / t_f / 1e6:5.2f} TB/s) | "
          f"fused, GELU not stored {t_fn:7.1f} us ({by_fn / t_fn / 1e6:5.2f} TB/s)")
