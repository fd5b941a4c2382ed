"""CPU oracle (test infrastructure): the multi-loss arithmetic.

Restates reference enhanced_training_multi_loss.py:143-416 (byte-identical copy at
training.py:158-365) with torch-CPU fp32 ops, as pure functions.
"""
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

_SOBEL_X = torch.tensor([[-1., 0., 1.], [-2., 0., 2.], [-1., 0., 1.]]).view(1, 1, 3, 3)
_SOBEL_Y = torch.tensor([[-1., -2., -1.], [0., 0., 0.], [1., 2., 1.]]).view(1, 1, 3, 3)


def _masked_mean(weighted_map, plain_map, w):
    # mean over the pixels with w > 0, dividing by their COUNT (not by sum of w);
    # plain mean of the unweighted map when no pixel has w > 0.
    # reference enhanced_training_multi_loss.py:174-180, 261-267
    valid = w > 0
    if valid.sum() > 0:
        return weighted_map[valid].mean()
    return plain_map.mean()


def pixel_loss(pred, target, w=None, loss_type="mse", use_tid_weights=True, scale=1.0):
    """reference enhanced_training_multi_loss.py:159-184"""
    if loss_type == "mse":
        lm = (pred - target) ** 2
    elif loss_type == "mae":
        lm = (pred - target).abs()
    else:
        raise ValueError(f"unsupported loss type: {loss_type}")
    if use_tid_weights and w is not None:
        return _masked_mean(lm * (w * scale), lm, w)
    return lm.mean()


def gradient_loss(pred, target, w=None, method="sobel"):
    """reference enhanced_training_multi_loss.py:202-269"""
    if method == "sobel":
        gx = (F.conv2d(pred, _SOBEL_X, padding=1) - F.conv2d(target, _SOBEL_X, padding=1)) ** 2
        gy = (F.conv2d(pred, _SOBEL_Y, padding=1) - F.conv2d(target, _SOBEL_Y, padding=1)) ** 2
        g = gx + gy
        if w is not None:
            return _masked_mean(g * w, g, w)
        return g.mean()
    # forward differences; x and y terms are averaged separately with shifted weights
    dx = ((pred[..., :, 1:] - pred[..., :, :-1]) - (target[..., :, 1:] - target[..., :, :-1])) ** 2
    dy = ((pred[..., 1:, :] - pred[..., :-1, :]) - (target[..., 1:, :] - target[..., :-1, :])) ** 2
    if w is not None:
        wx, wy = w[..., :, 1:], w[..., 1:, :]
        return (_masked_mean(dx * wx, dx, wx) + _masked_mean(dy * wy, dy, wy)) / 2
    return (dx.mean() + dy.mean()) / 2


def spectral_weight(H, W, factor):
    """1 + dist((y,x),(H//2,W//2)) / |(H//2,W//2)| * (factor-1) on UN-shifted FFT bins.

    reference enhanced_training_multi_loss.py:313-326.  The distance is taken to
    the index (H//2, W//2) of the unshifted transform, so the weight is largest at
    the DC bin and exactly 1 at the Nyquist bin.
    """
    ch, cw = H // 2, W // 2
    y = torch.arange(H).view(H, 1)
    x = torch.arange(W).view(1, W)
    dist = torch.sqrt(((y - ch) ** 2 + (x - cw) ** 2).to(torch.float32))
    dmax = torch.sqrt(torch.tensor(float(ch ** 2 + cw ** 2)))
    return 1.0 + (dist / dmax) * (factor - 1.0)


def spectral_loss(pred, target, w=None, focus_high_freq=True, factor=2.0):
    """reference enhanced_training_multi_loss.py:280-344"""
    if not focus_high_freq:
        # the reference reads H, W unbound in this branch (:338) and raises NameError
        raise NameError("name 'H' is not defined")
    B = pred.shape[0]
    H, W = pred.shape[-2:]
    fw = spectral_weight(H, W, factor)
    per_sample = []
    for i in range(B):
        lp = torch.log(torch.abs(torch.fft.fft2(pred[i, 0])) + 1e-8)
        lt = torch.log(torch.abs(torch.fft.fft2(target[i, 0])) + 1e-8)
        d = (lp - lt) ** 2 * fw
        if w is not None:
            d = d * w[i, 0].mean()
        per_sample.append(d.mean() / (H * W))
    return torch.stack(per_sample).mean() * 0.01


def multi_loss(cfg: Dict, pred, target, w: Optional[torch.Tensor] = None
               ) -> Tuple[torch.Tensor, Dict[str, float]]:
    """reference enhanced_training_multi_loss.py:376-416"""
    comb = cfg["loss_combination"]
    parts: Dict[str, float] = {}
    pw = None if comb == "baseline" else w
    lp = pixel_loss(pred, target, pw, cfg["pixel_loss_type"], cfg["use_tid_weights"],
                    cfg["tid_weight_scale"])
    parts["pixel"] = float(lp.detach())
    total = cfg["pixel_loss_weight"] * lp
    if comb in ("weighted_mse_grad", "all") and cfg["gradient_loss_weight"] > 0:
        lg = gradient_loss(pred, target, w, cfg["gradient_method"])
        parts["gradient"] = float(lg.detach())
        total = total + cfg["gradient_loss_weight"] * lg
    else:
        parts["gradient"] = 0.0
    if comb in ("weighted_mse_fft", "all") and cfg["spectral_loss_weight"] > 0:
        ls = spectral_loss(pred, target, w, cfg["spectral_focus_high_freq"],
                           cfg["high_freq_weight_factor"])
        parts["spectral"] = float(ls.detach())
        total = total + cfg["spectral_loss_weight"] * ls
    else:
        parts["spectral"] = 0.0
    parts["total"] = float(total.detach())
    return total, parts


def preset(comb: str, pixel_w=1.0, grad_w=0.3, spec_w=0.2, scale=1.0, factor=2.0,
           pixel_type="mse", use_tid=True, method="sobel") -> Dict:
    """A loss config dict with the keys MultiLossFunction reads (:350-374)."""
    return {
        "loss_combination": comb, "pixel_loss_weight": pixel_w, "gradient_loss_weight": grad_w,
        "spectral_loss_weight": spec_w, "pixel_loss_type": pixel_type, "use_tid_weights": use_tid,
        "tid_weight_scale": scale, "gradient_method": method, "spectral_focus_high_freq": True,
        "high_freq_weight_factor": factor,
    }
