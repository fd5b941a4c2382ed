"""TEST INFRASTRUCTURE ONLY: CPU restatement of one optimiser step of the reference trainer.

zero_grad -> loss.backward() -> clip_grad_norm_(max_norm=1.0) -> AdamW.step()
(enhanced_training_multi_loss.py:819-829; AdamW built at :758-763 with lr 5e-5, weight decay 5e-5).
Pinned by tests/golden/step.npz, which tests/golden/make_golden.py froze from the reference's own
modules and torch.optim.AdamW.  Only tests may import this.
"""
from typing import Dict, Tuple

import torch

import torch.nn.functional as F

from . import loss_oracle, swin_oracle, unet_oracle


def clip_coefficient(grads: Dict[str, torch.Tensor], max_norm: float) -> Tuple[float, float]:
    """torch.nn.utils.clip_grad_norm_: norm of the per-tensor L2 norms; coef = min(1, max/(norm+1e-6))."""
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g) for g in grads.values()]))
    coef = float(max_norm) / (float(total) + 1e-6)
    return float(total), min(coef, 1.0)


def adamw_update(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-2):
    """torch.optim.AdamW, single-tensor form, fp32.  Returns (p, m, v) after step number `step` (1-based)."""
    p = p * (1.0 - lr * weight_decay)
    m = m + (g - m) * (1.0 - beta1)
    v = v * beta2 + g * g * (1.0 - beta2)
    bc1 = 1.0 - beta1 ** step
    bc2_sqrt = (1.0 - beta2 ** step) ** 0.5
    denom = v.sqrt() / bc2_sqrt + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v


def training_step(sd, x, y, w, loss_cfg, lr=5e-5, weight_decay=5e-5, max_norm=1.0):
    """First step from fresh optimiser state, Dropout2d off.  Returns (loss parts, grad norm, new params)."""
    parts_box = {}

    def loss_fn(out):
        total, parts = loss_oracle.multi_loss(loss_cfg, out, y, w)
        parts_box.update(parts)
        return total

    _, _, grads, _ = unet_oracle.unet_fwd_bwd(sd, x, loss_fn)
    grads.pop("__input__")
    norm, coef = clip_coefficient(grads, max_norm)
    new = {}
    for k, g in grads.items():
        g = g * coef if coef < 1.0 else g
        new[k], _, _ = adamw_update(sd[k], g, torch.zeros_like(g), torch.zeros_like(g), 1, lr,
                                    weight_decay=weight_decay)
    return parts_box, norm, new


def swin_training_step(sd, x, y, w, cfg, loss_cfg, lr=1e-4, weight_decay=0.05, max_norm=1.0):
    """First step of the Swin trainer (swin_unet_model.py:809-823; AdamW at :759-763 with the config's
    LEARNING_RATE 1e-4 and WEIGHT_DECAY 0.05, :730-731): forward, bilinear resize of the half-resolution
    output to the label size, multi-loss, backward, clip_grad_norm_(1.0), AdamW.  DropPath off.
    Returns (loss parts, grad norm, new params, grads)."""
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()
              if v.is_floating_point() and "running_" not in k}
    full = dict(sd)
    full.update(params)
    out, _ = swin_oracle.swin_unet_forward(full, x, cfg, True, None)
    if out.shape[-2:] != y.shape[-2:]:
        out = F.interpolate(out, size=y.shape[-2:], mode="bilinear", align_corners=False)
    total, parts = loss_oracle.multi_loss(loss_cfg, out, y, w)
    total.backward()
    grads = {k: p.grad for k, p in params.items()}
    norm, coef = clip_coefficient(grads, max_norm)
    new = {}
    for k, g in grads.items():
        g = g * coef if coef < 1.0 else g
        new[k], _, _ = adamw_update(sd[k], g, torch.zeros_like(g), torch.zeros_like(g), 1, lr,
                                    weight_decay=weight_decay)
    return parts, norm, new, grads
