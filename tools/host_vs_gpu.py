#!/usr/bin/env python
"""Host enqueue time against GPU time of one training step: is the step bound by the launch queue?

    python tools/host_vs_gpu.py swin|unet [batch]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

which = sys.argv[1] if len(sys.argv) > 1 else "swin"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dt = torch.bfloat16
from unet_amd import ops
from unet_amd.losses import MultiLossFunction
from unet_amd.optim import FusedAdamW
from oracle import weights as W   # seeded inputs only

if which == "swin":
    from oracle import swin_oracle
    from unet_amd.swin import SwinUNet
    cfg = dict(swin_oracle.DEFAULT_CONFIG)
    model = SwinUNet(cfg, compute_dtype=dt)
    model.load_state_dict(W.fill_state_dict(swin_oracle.swin_template(cfg), seed=42))
    crit = MultiLossFunction(bench.SWIN_LOSS_CFG).to(dev)
    opt_args = dict(lr=1e-4, weight_decay=0.05)
else:
    from unet_amd import OptimizedUNet
    model = OptimizedUNet(4, 1, 48, compute_dtype=dt)
    model.load_state_dict(W.fill_state_dict(W.unet_template(4, 1, 48), seed=42))
    crit = MultiLossFunction(bench.LOSS_CFG).to(dev)
    opt_args = dict(lr=5e-5, weight_decay=5e-5)
model = model.to(dev).train()
x, y, w = [t.to(dev) for t in W.synthetic_batch(B, 4, 256, 256, seed=42)]
opt = FusedAdamW(model.parameters(), max_grad_norm=1.0, **opt_args)


def step():
    for p in model.parameters():
        p.grad = None
    out = model(x)
    if out.shape[-2:] != y.shape[-2:]:
        out = ops.interpolate_bilinear(out, y.shape[-2:])
    loss, _ = crit(out, y, w)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
host, total = [], []
for _ in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3)
    total.append((t2 - t0) * 1e3)
print(f"{which} b{B}: host enqueue {sorted(host)[4]:.2f} ms, step (from an idle GPU) {sorted(total)[4]:.2f} ms")
# back-to-back steps
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step()
th = time.perf_counter() - t0
torch.cuda.synchronize()
tt = time.perf_counter() - t0
print(f"10 steps back to back: host {th*100:.2f} ms/step, wall {tt*100:.2f} ms/step")
