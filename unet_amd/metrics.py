"""Validation metrics without leaving the device.

The reference's validation loop copies every batch of predictions and targets to the host, keeps
them all and calls ``np.corrcoef`` at the end (enhanced_training_multi_loss.py:883-889);
``predict.py:188-200`` computes MSE / MAE / Pearson r the same way per test set.  Here each batch adds
eight double-precision sums to a device-resident state and ``compute()`` does the one copy.
"""
import math
from typing import Dict

import torch

from . import _lib
from ._lib import call, ptr, stream


class StreamingMetrics:
    """``update(pred, target)`` per batch (float32 CUDA tensors of equal shape), then ``compute()`` ->
    ``{'correlation', 'mse', 'mae', 'count'}`` over everything seen since the last ``reset()``."""

    def __init__(self, device=None):
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("StreamingMetrics runs on the GPU only (no CPU fallback)")
        self.state = torch.zeros(8, dtype=torch.float64, device=self.device)

    def reset(self):
        self.state.zero_()

    def update(self, pred: torch.Tensor, target: torch.Tensor):
        if pred.shape != target.shape:
            raise ValueError(f"shape mismatch {tuple(pred.shape)} vs {tuple(target.shape)}")
        if not (pred.is_cuda and target.is_cuda):
            raise RuntimeError("StreamingMetrics takes CUDA tensors")
        p = pred.detach().contiguous().float()
        t = target.detach().contiguous().float()
        n = p.numel()
        if n == 0:
            return
        nblk = _lib.load().unetk_metrics_num_partials(n)
        partials = torch.empty(nblk * 8, dtype=torch.float64, device=self.device)
        call("unetk_metrics_update", ptr(p), ptr(t), n, ptr(partials), ptr(self.state), stream())

    def compute(self) -> Dict[str, float]:
        n, sp, st, spp, stt, spt, sabs, ssq = self.state.tolist()
        if n == 0:
            return {"correlation": float("nan"), "mse": float("nan"), "mae": float("nan"), "count": 0}
        cov = spt - sp * st / n
        vp, vt = spp - sp * sp / n, stt - st * st / n
        den = math.sqrt(vp * vt) if vp > 0 and vt > 0 else 0.0
        return {"correlation": cov / den if den > 0 else float("nan"), "mse": ssq / n, "mae": sabs / n,
                "count": int(n)}
