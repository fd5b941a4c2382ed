"""Swin loss curves, fp32 / bf16 / fp32 with 1e-6 input noise, for several learning rates: picks the regime in which
tests/test_gpu_swin.py::test_swin_bf16_loss_curve_tracks_fp32 compares precision and not optimiser noise."""
import sys
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from oracle import swin_oracle
from unet_amd.optim import FusedAdamW
from unet_amd.swin import SwinUNet

cfg = dict(swin_oracle.DEFAULT_CONFIG)
torch.manual_seed(7)
B, H, Wd, steps = 8, 64, 64, 40
x = F.avg_pool2d(torch.randn(B, 4, H + 4, Wd + 4), 5, 1)
x = x / x.std()
yf = torch.tanh(0.8 * x[:, :1] - 0.6 * x[:, 1:2] * x[:, 2:3]) + 0.3 * F.avg_pool2d(F.pad(x[:, 3:4], (1, 1, 1, 1)), 3, 1)
y = F.avg_pool2d(yf, 2)
init = SwinUNet(cfg).state_dict()
noise = torch.randn_like(x)
draws = [swin_oracle.draw_drop_path(cfg, B) for _ in range(steps)]


def run(dt, lr, eps=0.0):
    m = SwinUNet(cfg, compute_dtype=dt)
    m.load_state_dict(init)
    m = m.cuda().train()
    opt = FusedAdamW(m.parameters(), lr=lr, weight_decay=0.05, max_grad_norm=1.0)
    xd, yd = (x * (1 + eps * noise)).cuda(), y.cuda()
    losses = []
    for s in range(steps):
        m.drop_path_override = draws[s]
        opt.zero_grad()
        loss = ((m(xd) - yd) ** 2).mean()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    return np.array(losses)


for lr in (2e-5, 5e-5, 1e-4):
    f32, b16, fe = run(torch.float32, lr), run(torch.bfloat16, lr), run(torch.float32, lr, 1e-6)
    dev, sd = np.abs(b16 - f32) / f32, np.abs(fe - f32) / f32
    print("lr", lr, "fp32", np.round(f32, 3).tolist())
    print("  bf16 dev", np.round(dev, 3).tolist())
    print("  self dev", np.round(sd, 3).tolist())
    for n in (20, 30, 40):
        print(f"  first {n}: bf16 max {dev[:n].max():.3f} mean {dev[:n].mean():.3f}; self max {sd[:n].max():.3f}")
