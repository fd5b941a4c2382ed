#!/usr/bin/env python
"""Time of the one-launch weight packing of a training step (UNet(4,1,48) and the Swin-UNet, both forms, bf16)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_amd import ops
from unet_amd.unet import OptimizedUNet
from unet_amd.swin import SwinUNet

def timeit(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

from oracle import swin_oracle          # (tooling only: the default Swin-UNet configuration)
for name, model in (("unet", OptimizedUNet(4, 1, 48)), ("swin", SwinUNet(dict(swin_oracle.DEFAULT_CONFIG)))):
    model = model.cuda()
    ws = [m.weight for m in model.modules() if isinstance(m, (torch.nn.Linear, torch.nn.Conv2d))
          and (m.weight.dim() == 2 or (m.weight.shape[2] == m.weight.shape[3] and m.weight.shape[2] in (1, 3)))]
    n = sum(w.numel() for w in ws)
    for forms in ((False,), (True,), (False, True)):
        jobs = [(w, tr) for tr in forms for w in ws]
        t = timeit(lambda: ops.pack_many(jobs, torch.bfloat16))
        print(f"{name} {len(ws)} weights {n/1e6:.1f} M elements forms {forms}: {t:.1f} us  ({n*len(forms)*6/t/1e6:.2f} TB/s of 4 B in + 2 B out)", flush=True)
