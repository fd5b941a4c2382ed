"""TEST INFRASTRUCTURE ONLY: the reference's validation metrics, restated with the library calls it makes.

correlation: np.corrcoef(predictions.flatten(), targets.flatten())[0, 1] over the concatenated
validation set (enhanced_training_multi_loss.py:883-889); predict.py:196-200 uses
scipy.stats.pearsonr on the same flattened arrays.  MSE / MAE: F.mse_loss / F.l1_loss
(predict.py:188-194).  "Pinned" by call site: the reference's code for these IS the library call.
"""
from typing import Dict, List

import numpy as np


def validation_metrics(preds: List[np.ndarray], targets: List[np.ndarray]) -> Dict[str, float]:
    p = np.concatenate([a.reshape(a.shape[0], -1) for a in preds], axis=0).flatten()
    t = np.concatenate([a.reshape(a.shape[0], -1) for a in targets], axis=0).flatten()
    d = p.astype(np.float64) - t.astype(np.float64)
    return {"correlation": float(np.corrcoef(p, t)[0, 1]), "mse": float(np.mean(d * d)),
            "mae": float(np.mean(np.abs(d))), "count": int(p.size)}
