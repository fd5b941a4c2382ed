#!/usr/bin/env python
"""Patch extraction throughput: quality scan + gather over a synthetic grid, GPU vs the CPU oracle."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import patch_oracle
from unet_amd.data import EnhancedDataPreparation

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
P, S, MV = 128, 32, 0.25     # enhanced_training_multi_loss.py:88-90
feats, labels, weights = patch_oracle.synthetic_fields(G, G, 1)
cfg = type("C", (), {"OUTPUT_DIR": tempfile.mkdtemp(), "PATCH_SIZE": P, "STRIDE": S, "MIN_VALID_RATIO": MV})
prep = EnhancedDataPreparation(cfg)
d = torch.device("cuda")
f = torch.from_numpy(np.stack([feats[v] for v in patch_oracle.FEATURE_VARS])).to(d)
l, w = torch.from_numpy(labels).to(d), torch.from_numpy(weights).to(d)
for _ in range(2):
    pos = prep.scan(f, l, w)
torch.cuda.synchronize()
t0 = time.perf_counter()
pos = prep.scan(f, l, w)
torch.cuda.synchronize()
t_scan = time.perf_counter() - t0
ncand = len(patch_oracle.candidate_starts(G, P, S)) ** 2
t0 = time.perf_counter()
n = 0
for s in range(0, pos.shape[0], 1024):
    out = prep.gather(f, l, w, pos[s:s + 1024])
    n += out[0].shape[0]
torch.cuda.synchronize()
t_gather = time.perf_counter() - t0
print(f"grid {G}^2: {ncand} candidates, {n} accepted; scan {t_scan*1e3:.2f} ms "
      f"({ncand * 6 * P * P * 4 / t_scan / 1e12:.2f} TB/s of window reads), gather {t_gather*1e3:.2f} ms "
      f"({n * 6 * P * P * 4 * 2 / t_gather / 1e12:.2f} TB/s)")
sub = 640
fs = {k: v[:sub, :sub] for k, v in feats.items()}
t0 = time.perf_counter()
ref = patch_oracle.extract_patches(fs, labels[:sub, :sub], weights[:sub, :sub], P, S, MV)
t_cpu = time.perf_counter() - t0
nc = len(patch_oracle.candidate_starts(sub, P, S)) ** 2
print(f"CPU oracle on a {sub}^2 corner: {nc} candidates in {t_cpu*1e3:.1f} ms = {nc / t_cpu:.0f} candidates/s; "
      f"GPU {ncand / (t_scan + t_gather):.0f} candidates/s")
