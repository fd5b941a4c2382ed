#!/usr/bin/env python
"""Which bf16-stored tensor class costs what (CPU, oracle only; not collected by pytest).

    python tests/ablate_bf16_storage.py [case ...]

Runs the CPU oracle's train-mode forward + backward with ONE class of stored tensors rounded to
bfloat16 at a time (packed weights, raw conv outputs, activations, conv-output gradients, activation
gradients), then with all of them, and prints the distance of outputs and parameter gradients from
the fp32 run next to the reference's own bf16-autocast distance frozen in tests/golden/bf16.npz.
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import unet_oracle  # noqa: E402
from oracle import weights as W  # noqa: E402

CASES = {"c1": (1, 48, 4, 128, 128), "small": (4, 8, 2, 32, 32), "ragged": (4, 8, 2, 40, 24),
         "odd": (4, 16, 1, 50, 36), "f48": (4, 48, 1, 64, 64)}


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def main():
    gold = np.load(os.path.join(REPO, "tests", "golden", "bf16.npz"), allow_pickle=False)
    for name in (sys.argv[1:] or list(CASES)):
        cin, f, B, H, Wd = CASES[name]
        sd = W.fill_state_dict(W.unet_template(cin, 1, f), seed=42)
        x, y, _ = W.synthetic_batch(B, cin, H, Wd, seed=42)
        lf = lambda o: ((o - y) ** 2).mean()
        out32, _, g32, _ = unet_oracle.unet_fwd_bwd(sd, x, lf)
        names = [k for k in g32 if k != "__input__"]
        flat = lambda g: torch.cat([g[k].double().reshape(-1) for k in names])
        ref = flat(g32)
        print(f"{name}: reference autocast out {gold[f'unet.{name}.out_rel']:.3f} grad {gold[f'unet.{name}.grad_rel']:.3f} "
              f"cos {gold[f'unet.{name}.grad_cos']:.3f}")
        for classes in (["w"], ["y"], ["a"], ["dy"], ["da"], ["w", "a"], ["w", "a", "dy", "da"], ["w", "y", "a", "dy", "da"]):
            o, _, g, _ = unet_oracle.unet_fwd_bwd(sd, x, lf, None, set(classes))
            gf = flat(g)
            cos = float(torch.dot(gf, ref) / (gf.norm() * ref.norm()))
            print(f"   rounded {'+'.join(classes):14s} out {rel(o, out32):.4f}  grad {rel(gf, ref):.4f}  cos {cos:.4f}")


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()
