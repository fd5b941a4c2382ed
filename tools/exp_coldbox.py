#!/usr/bin/env python
"""Step time and host enqueue time of the UNet training step, group by group, from process start (cold box study)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import weights as W
from unet_amd import OptimizedUNet
from unet_amd.losses import MultiLossFunction
from unet_amd.optim import FusedAdamW
import bench
dev = torch.device("cuda:0")
model = OptimizedUNet(4, 1, 48, compute_dtype=torch.bfloat16)
model.load_state_dict(W.fill_state_dict(W.unet_template(4, 1, 48), seed=42))
model = model.to(dev).train()
crit = MultiLossFunction(bench.LOSS_CFG).to(dev)
x, y, w = [t.to(dev) for t in W.synthetic_batch(64, 4, 256, 256, seed=42)]
opt = FusedAdamW(model.parameters(), lr=5e-5, weight_decay=5e-5, max_grad_norm=1.0)
def step():
    for p in model.parameters():
        p.grad = None
    loss, _ = crit(model(x), y, w)
    loss.backward()
    opt.step()
t_start = time.perf_counter()
while time.perf_counter() - t_start < float(sys.argv[1]) if len(sys.argv) > 1 else 14:
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"t={t0 - t_start:6.2f}s  step {1e2 * (t2 - t0):6.2f} ms  host enqueue {1e2 * (t1 - t0):6.2f} ms", flush=True)
