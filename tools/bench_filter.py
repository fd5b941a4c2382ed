#!/usr/bin/env python
"""Timing of the float64 Gaussian low-pass on an 8192 x 8192 grid (sigma 31 pixels = 135 km at 1 arc-minute)
next to scipy.ndimage.gaussian_filter on the host (a 2048 x 2048 corner, scaled)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from unet_amd import prep

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
sig = float(sys.argv[2]) if len(sys.argv) > 2 else 31.0
x = torch.randn((n, n), dtype=torch.float64, device="cuda")
prep.gaussian_filter_device(x, sig, sig); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): prep.gaussian_filter_device(x, sig, sig)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
r = int(4 * sig + 0.5)
flops = 2.0 * 2 * (2 * r + 1) * n * n
print(f"GPU  {n}x{n} sigma {sig}: {ms:.3f} ms  {n*n/ms/1e3:.1f} Mpix/s  {flops/ms/1e9:.1f} TFLOP/s f64 (two passes of {2*r+1} taps), "
      f"{4*16*n*n/ms/1e9:.2f} TB/s if it were bandwidth (4 passes r+w)")
from scipy.ndimage import gaussian_filter
m = 2048
h = np.random.default_rng(0).standard_normal((m, m))
t0 = time.perf_counter(); gaussian_filter(h, sigma=[sig, sig], mode="reflect"); t1 = time.perf_counter()
print(f"CPU  scipy {m}x{m}: {(t1-t0)*1e3:.1f} ms  {m*m/(t1-t0)/1e6:.2f} Mpix/s (1 thread)")
