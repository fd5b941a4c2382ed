#!/usr/bin/env python
"""Sensitivity of the bf16-vs-fp32 loss-curve comparison to the learning rate and to the fp32 path's own rounding."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from unet_amd import OptimizedUNet
from unet_amd.optim import FusedAdamW

torch.manual_seed(7)
B, H, Wd = 8, 64, 64
x = F.avg_pool2d(torch.randn(B, 4, H + 4, Wd + 4), 5, 1)
x = x / x.std()
y = torch.tanh(0.8 * x[:, :1] - 0.6 * x[:, 1:2] * x[:, 2:3]) + 0.3 * F.avg_pool2d(F.pad(x[:, 3:4], (1, 1, 1, 1)), 3, 1)
init = OptimizedUNet(4, 1, 16).state_dict()


def run(dt, lr, steps, eps=0.0):
    m = OptimizedUNet(4, 1, 16, compute_dtype=dt)
    m.load_state_dict(init)
    m = m.cuda().train()
    m.dropout_masks_override = [None] * 9
    opt = FusedAdamW(m.parameters(), lr=lr, weight_decay=5e-5, max_grad_norm=1.0)
    xd, yd = (x * (1 + eps * torch.randn_like(x))).cuda(), y.cuda()
    out = []
    for _ in range(steps):
        opt.zero_grad()
        loss = ((m(xd) - yd) ** 2).mean()
        loss.backward()
        opt.step()
        out.append(loss.item())
    return np.array(out)


for lr, steps in ((1e-3, 20), (5e-4, 30), (3e-4, 40)):
    f = run(torch.float32, lr, steps)
    f2 = run(torch.float32, lr, steps, 1e-6)
    b = run(torch.bfloat16, lr, steps)
    dev, self_dev = np.abs(b - f) / f, np.abs(f2 - f) / f
    print(f"lr {lr} steps {steps}: final/initial {f[-1]/f[0]:.4f}; bf16 dev max {dev.max():.4f} mean {dev.mean():.4f}; "
          f"fp32 self (1e-6 input noise) dev max {self_dev.max():.4f} mean {self_dev.mean():.4f}")
    print("   dev      ", np.round(dev, 3).tolist())
    print("   self dev ", np.round(self_dev, 3).tolist())
