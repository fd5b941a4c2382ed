"""CPU oracle (test infrastructure): sliding-window tiled prediction with Gaussian blend.

Restates reference large_area_predictor.py:71-81 and :133-219 in numpy.  Features
are a dict of 2-D float32 arrays (lat, lon); the model is any callable mapping a
float32 (n, 4, P, P) array to (n, 1, P, P).
"""
from typing import Callable, Dict, List, Tuple

import numpy as np

FEATURE_VARS = ("GA", "DOV_EW", "DOV_NS", "VGG")   # large_area_predictor.py:161


def gaussian_weight_kernel(patch_size: int) -> np.ndarray:
    """exp(-((x-(c-.5))^2+(y-(c-.5))^2) / (2 sigma^2)), c=P/2, sigma=c/2, float64 -> float32.

    reference large_area_predictor.py:71-81
    """
    c = patch_size / 2.0
    sigma = c * 0.5
    if sigma <= 0:
        sigma = 1.0
    idx = np.arange(patch_size, dtype=np.int64)
    y = idx.reshape(-1, 1)
    x = idx.reshape(1, -1)
    k = np.exp(-0.5 * ((x - (c - 0.5)) ** 2 + (y - (c - 0.5)) ** 2) / sigma ** 2)
    return k.astype(np.float32)


def tile_starts(size: int, patch: int, stride: int) -> List[int]:
    """range(0, size - patch + 1, stride)  (large_area_predictor.py:151-152)"""
    return list(range(0, size - patch + 1, stride))


def collect_tiles(features: Dict[str, np.ndarray], patch: int, stride: int
                  ) -> Tuple[List[Tuple[int, int]], List[np.ndarray]]:
    """Tile positions (lat-major) and NaN-cleaned (4,P,P) inputs.

    A tile is dropped when any of the four variables is missing or entirely NaN in
    its window; remaining NaNs become 0 (large_area_predictor.py:157-182).
    """
    lat, lon = next(iter(features.values())).shape
    pos, tiles = [], []
    for i in tile_starts(lat, patch, stride):
        for j in tile_starts(lon, patch, stride):
            chans, ok = [], True
            for var in FEATURE_VARS:
                if var not in features:
                    ok = False
                    break
                win = features[var][i:i + patch, j:j + patch]
                if np.all(np.isnan(win)):
                    ok = False
                    break
                chans.append(win)
            if ok and len(chans) == 4:
                pos.append((i, j))
                tiles.append(np.nan_to_num(np.stack(chans, 0), 0))
    return pos, tiles


def predict_large_area(model_fn: Callable[[np.ndarray], np.ndarray], features: Dict[str, np.ndarray],
                       patch: int, stride: int, batch: int = 16) -> np.ndarray:
    """reference large_area_predictor.py:133-219"""
    lat, lon = next(iter(features.values())).shape
    acc = np.zeros((lat, lon), np.float32)
    cnt = np.zeros((lat, lon), np.float32)
    wk = gaussian_weight_kernel(patch)
    pos, tiles = collect_tiles(features, patch, stride)
    for s in range(0, len(tiles), batch):
        x = np.array(tiles[s:s + batch]).astype(np.float32)
        y = np.asarray(model_fn(x))
        for k, (i, j) in enumerate(pos[s:s + batch]):
            p = np.nan_to_num(y[k, 0], 0)
            acc[i:i + patch, j:j + patch] += p * wk
            cnt[i:i + patch, j:j + patch] += wk
    safe = np.where(cnt > 1e-9, cnt, 1.0)
    return np.divide(acc, safe)


def blend_only(preds: np.ndarray, pos: List[Tuple[int, int]], lat: int, lon: int, patch: int
               ) -> np.ndarray:
    """The accumulation + normalisation half alone, given per-tile predictions (n,P,P)."""
    acc = np.zeros((lat, lon), np.float32)
    cnt = np.zeros((lat, lon), np.float32)
    wk = gaussian_weight_kernel(patch)
    for k, (i, j) in enumerate(pos):
        p = np.nan_to_num(preds[k], 0)
        acc[i:i + patch, j:j + patch] += p * wk
        cnt[i:i + patch, j:j + patch] += wk
    return np.divide(acc, np.where(cnt > 1e-9, cnt, 1.0))
