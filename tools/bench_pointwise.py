#!/usr/bin/env python
"""Timing and effective bandwidth of the HBM-bound passes on the UNet shapes of the headline step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_amd import ops, _lib
if os.environ.get("UNETK_AB_LIB"):      # A/B a kernel variant built to another file
    _lib.LIB_PATH = os.path.abspath(os.environ["UNETK_AB_LIB"])

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dt = torch.bfloat16
f = 48
levels = [(256, f), (128, 2 * f), (64, 4 * f), (32, 8 * f)]


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


# what a plain library copy / read reaches on this card at the same sizes (the practical ceiling)
for mb in (100, 403, 1600):
    n = mb * 1000 * 1000 // 2
    src = torch.randn(n, device="cuda").to(dt)
    dst = torch.empty_like(src)
    t_c = timeit(lambda: dst.copy_(src))
    t_r = timeit(lambda: torch.sum(src.view(torch.int16), dtype=torch.int32))
    print(f"ceiling {mb:5d} MB: torch copy {2 * mb / 1e3 / t_c:5.2f} TB/s, torch read-sum {mb / 1e3 / t_r:5.2f} TB/s", flush=True)
    del src, dst

tot = {}
for hw, C in levels:
    S = B * hw * hw * C * 2 / 1e9     # GB of one activation tensor
    y = torch.randn(B, hw, hw, C, device="cuda").to(dt)
    g = torch.randn(B, hw, hw, C, device="cuda").to(dt)
    keep = (torch.rand(B, C, device="cuda") > 0.05).float()
    stats = torch.stack([y.float().sum((0, 1, 2)), (y.float() ** 2).sum((0, 1, 2))]).view(1, 2, C).contiguous()
    mean, invstd, sc, sh = ops.bn_finalize(stats.clone(), B * hw * hw, torch.ones(C, device="cuda"), torch.zeros(C, device="cuda"))
    res = {}
    res["bn_act_fwd+pool"] = (timeit(lambda: ops.bn_act_fwd(y, sc, sh, keep, relu=True, pool=True)), 2.25 * S)
    res["bn_act_fwd"] = (timeit(lambda: ops.bn_act_fwd(y, sc, sh, None, relu=True, pool=False)), 2 * S)
    res["bn_act_bwd"] = (timeit(lambda: ops.bn_act_bwd(g, y, sc, sh, mean, invstd, keep, relu=True)), 5 * S)
    a, pooled = ops.bn_act_fwd(y, sc, sh, keep, relu=True, pool=True)
    dp = torch.randn_like(pooled)
    dskip = torch.randn(B, hw, hw, C, device="cuda").to(dt)
    dup = torch.randn(B, hw, hw, 2 * C, device="cuda").to(dt)
    res["pool_bwd"] = (timeit(lambda: ops.pool_bwd(a, dp, dskip)), 3.25 * S)
    low = torch.randn(B, hw // 2, hw // 2, 2 * C, device="cuda").to(dt)
    res["upsample_concat"] = (timeit(lambda: ops.upsample_concat(low, a)), (0.5 + 1 + 3) * S)
    cat = torch.empty(B, hw, hw, 3 * C, device="cuda", dtype=dt)
    res["upsample_into"] = (timeit(lambda: ops.upsample_into(low, cat)), (0.5 + 2) * S)
    res["upsample_bwd"] = (timeit(lambda: ops.upsample_bwd(dup, 2 * C, hw // 2, hw // 2)), 2.5 * S)
    # the last BatchNorm + ReLU of the block below and its resize to this level: two passes / one
    ylow = torch.randn(B, hw // 2, hw // 2, 2 * C, device="cuda").to(dt)
    sc2, sh2 = torch.rand(2 * C, device="cuda") + 0.5, torch.randn(2 * C, device="cuda") * 0.3
    res["bn+upsample 2 passes"] = (timeit(lambda: ops.upsample(ops.bn_act_fwd(ylow, sc2, sh2, None, relu=True)[0], hw, hw)), (0.5 + 0.5 + 0.5 + 2) * S)
    if ops.bn_act_upsample_supported(ylow):
        res["bn+upsample fused"] = (timeit(lambda: ops.bn_act_upsample(ylow, sc2, sh2, None)), (0.5 + 2) * S)
    line = f"{hw:4d}^2 C={C:4d} S={S*1e3:6.0f} MB |"
    for k, (t, gb) in res.items():
        line += f" {k} {t*1e3:6.0f} us {gb/t:5.2f} TB/s |"
        tot[k] = tot.get(k, 0) + t
    print(line, flush=True)
print({k: round(v, 3) for k, v in tot.items()})
