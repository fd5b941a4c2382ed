#!/usr/bin/env python
"""Which torch operators (and how many device copies / fills) one eager trainer step still launches beside the
library's own kernels: the headline UNet step or the Swin step under torch.profiler, operators by call count.

    python tools/torch_ops_step.py train|swin [batch] [regex: also list every operator / kernel whose name matches]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity

which = sys.argv[1] if len(sys.argv) > 1 else "train"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda:0")
dt = torch.bfloat16
from oracle import weights as W          # synthetic weights and batch only (a bench-side helper, not the product path)
import bench
from unet_amd import _lib
if os.environ.get("UNETK_AB_LIB"):      # A/B a kernel variant built to another file
    _lib.LIB_PATH = os.path.abspath(os.environ["UNETK_AB_LIB"])

if which == "train":
    from unet_amd import OptimizedUNet
    from unet_amd.losses import MultiLossFunction
    from unet_amd.optim import FusedAdamW
    model = OptimizedUNet(4, 1, 48, compute_dtype=dt)
    model.load_state_dict(W.fill_state_dict(W.unet_template(4, 1, 48), seed=42))
    model = model.to(dev).train()
    crit = MultiLossFunction(bench.LOSS_CFG).to(dev)
    x, y, w = (t.to(dev) for t in W.synthetic_batch(B, 4, 256, 256, seed=42))
    opt = FusedAdamW(model.parameters(), lr=5e-5, weight_decay=5e-5, max_grad_norm=1.0)
    params = list(model.parameters())

    def step():
        for p in params:
            p.grad = None
        loss, _ = crit(model(x), y, w)
        loss.backward()
        opt.step()
else:
    from oracle import swin_oracle
    from unet_amd import ops
    from unet_amd.losses import MultiLossFunction
    from unet_amd.optim import FusedAdamW
    from unet_amd.swin import SwinUNet
    cfg = dict(swin_oracle.DEFAULT_CONFIG)
    model = SwinUNet(cfg, compute_dtype=dt)
    model.load_state_dict(W.fill_state_dict(swin_oracle.swin_template(cfg), seed=42))
    model = model.to(dev).train()
    x, y, w = (t.to(dev) for t in W.synthetic_batch(B, 4, 256, 256, seed=42))
    crit = MultiLossFunction(bench.SWIN_LOSS_CFG).to(dev)
    opt = FusedAdamW(model.parameters(), lr=1e-4, weight_decay=0.05, max_grad_norm=1.0)
    params = list(model.parameters())

    def step():
        for p in params:
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
N = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(N):
        step()
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.count)
print(f"{'operator':60s} {'calls/step':>10s} {'cpu us/call':>12s} {'gpu us/call':>12s}")
import re
flt = re.compile(sys.argv[3]) if len(sys.argv) > 3 else None
for i, e in enumerate(rows):
    if i >= 60 and not (flt and flt.search(e.key)):
        continue
    gpu = getattr(e, "device_time_total", 0.0) / max(e.count, 1)
    print(f"{e.key[:60]:60s} {e.count / N:10.1f} {e.cpu_time_total / max(e.count, 1):12.1f} {gpu:12.1f}")
