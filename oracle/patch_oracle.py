"""TEST INFRASTRUCTURE ONLY: CPU restatement of the reference's training-patch extraction.

EnhancedDataPreparation.extract_patches (enhanced_training_multi_loss.py:633-720) and the
sample preparation of EnhancedBathymetryDataset.__getitem__ / EnhancedDataAugmentation
(:425-450, :470-513).  Pinned by tests/golden/patch.npz (frozen from the imported reference).
"""
from typing import Dict, List, Optional, Tuple

import numpy as np

FEATURE_VARS = ("GA", "DOV_EW", "DOV_NS", "VGG")     # :682


def candidate_starts(size: int, patch: int, stride: int) -> range:
    """:648-649: a margin of patch//8 cells is kept free on every side."""
    margin = patch // 8
    return range(margin, size - patch - margin + 1, stride)


def patch_passes(label: np.ndarray, weights: np.ndarray, feats: List[np.ndarray], min_valid: float) -> bool:
    """:666-701, in the reference's order of tests."""
    if np.sum(~np.isnan(label)) / label.size < min_valid:
        return False
    if np.nanstd(label) < 1e-6:
        return False
    if np.sum(weights > 0) / weights.size < min_valid * 0.5:
        return False
    for f in feats:
        if np.sum(~np.isnan(f)) / f.size < min_valid:
            return False
    return True


def extract_patches(features: Dict[str, np.ndarray], labels: np.ndarray, weights: np.ndarray, patch: int,
                    stride: int, min_valid: float = 0.25) -> List[Dict]:
    lat, lon = labels.shape
    out = []
    if any(v not in features for v in FEATURE_VARS):
        return out
    for i in candidate_starts(lat, patch, stride):
        for j in candidate_starts(lon, patch, stride):
            lab = labels[i:i + patch, j:j + patch]
            wts = weights[i:i + patch, j:j + patch]
            fs = [features[v][i:i + patch, j:j + patch] for v in FEATURE_VARS]
            if not patch_passes(lab, wts, fs, min_valid):
                continue
            out.append({"features": np.stack(fs, axis=0), "label": lab, "tid_weights": wts,
                        "lat_start": i, "lon_start": j})
    return out


def dataset_item(p: Dict, flip: Optional[str] = None) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """__getitem__ (:470-513): float32, NaN/inf -> 0, label and weights get a channel axis;
    flip None | 'h' (torch.flip dims=[2], the last axis) | 'v' (dims=[1]) as the augmentation draws it."""
    f = np.nan_to_num(p["features"].astype(np.float32), nan=0.0, posinf=0.0, neginf=0.0)
    lab = np.nan_to_num(p["label"].astype(np.float32)[None], nan=0.0, posinf=0.0, neginf=0.0)
    w = np.nan_to_num(p["tid_weights"].astype(np.float32)[None], nan=0.0, posinf=0.0, neginf=0.0)
    if flip == "h":
        f, lab, w = f[:, :, ::-1], lab[:, :, ::-1], w[:, :, ::-1]
    elif flip == "v":
        f, lab, w = f[:, ::-1], lab[:, ::-1], w[:, ::-1]
    return np.ascontiguousarray(f), np.ascontiguousarray(lab), np.ascontiguousarray(w)


def synthetic_fields(lat: int, lon: int, seed: int):
    """Fields with the features of the real grids that the quality rules react to: NaN land
    regions in the labels, NaN speckle and a dead block in the features, zero-weight areas and a
    constant (zero-variance) label plateau."""
    g = np.random.Generator(np.random.PCG64(seed))
    feats = {v: g.normal(0, 1, (lat, lon)).astype(np.float32) for v in FEATURE_VARS}
    labels = g.normal(0, 1, (lat, lon)).astype(np.float32)
    weights = g.choice(np.array([0.0, 1.0, 5.0], np.float32), size=(lat, lon), p=[0.2, 0.7, 0.1])
    labels[: lat // 3, : lon // 2] = np.nan                       # land
    labels[lat // 2:, lon // 2:] = 0.75                            # plateau: std == 0
    labels[g.random((lat, lon)) < 0.02] = np.nan
    weights[:, : lon // 5] = 0.0                                   # unweighted strip
    weights[g.random((lat, lon)) < 0.01] = np.nan
    feats["DOV_NS"][lat // 3: lat // 3 + lat // 4, lon // 2: lon // 2 + lon // 3] = np.nan
    feats["VGG"][g.random((lat, lon)) < 0.05] = np.nan
    feats["GA"][0, 0] = np.inf
    return feats, labels, weights
