"""Training-patch extraction on the GPU.

Drop-in for ``EnhancedDataPreparation.extract_patches`` (enhanced_training_multi_loss.py:633-720):
same constructor argument (a config with PATCH_SIZE, STRIDE, MIN_VALID_RATIO, OUTPUT_DIR), same
duck-typed inputs (``features_ds[var].values`` / ``.isel(...)``, ``labels_da.lat`` ...), same list of
``{'features', 'label', 'tid_weights', 'lat_start', 'lon_start'}`` dicts in the same order.  The
quality rules run as one reduction kernel over all candidate windows and the accepted windows are
gathered in one launch; ``device_batches`` hands them over as device tensors, already cleaned of
NaN/inf and optionally flipped, which is what the dataset's ``__getitem__`` (:470-513) produces.
File I/O (``load_data``) is not part of this package.
"""
import os
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np
import torch

from ._lib import call, ptr, stream

FEATURE_VARS = ("GA", "DOV_EW", "DOV_NS", "VGG")     # enhanced_training_multi_loss.py:682


def _values(da) -> np.ndarray:
    v = getattr(da, "values", None)
    if v is not None and not callable(v):
        return np.asarray(v, dtype=np.float32)
    return np.asarray(da.isel(lat=slice(0, len(da.lat)), lon=slice(0, len(da.lon))).values, dtype=np.float32)


class EnhancedDataPreparation:
    def __init__(self, config):
        self.config = config
        os.makedirs(config.OUTPUT_DIR, exist_ok=True)            # :609
        self.device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None

    def load_data(self):
        raise NotImplementedError("NetCDF input is the caller's business; pass the opened arrays to extract_patches")

    # ------------------------------------------------------------------ device side
    def to_device(self, features_ds, labels_da, tid_weights_da) -> Optional[Tuple[torch.Tensor, ...]]:
        if self.device is None:
            raise RuntimeError("patch extraction runs on the GPU only (no CPU fallback)")
        if any(v not in features_ds for v in FEATURE_VARS):
            return None                                            # :703-706: no patch is ever valid
        lat, lon = len(labels_da.lat), len(labels_da.lon)
        feats = np.stack([self._window(features_ds[v], lat, lon) for v in FEATURE_VARS])
        d = self.device
        return (torch.from_numpy(feats).to(d), torch.from_numpy(_values(labels_da)).to(d),
                torch.from_numpy(_values(tid_weights_da)).to(d))

    @staticmethod
    def _window(var, lat, lon) -> np.ndarray:
        v = getattr(var, "values", None)
        if v is not None and not callable(v):
            return np.asarray(v, dtype=np.float32)[:lat, :lon]
        return np.asarray(var.isel(lat=slice(0, lat), lon=slice(0, lon)).values, dtype=np.float32)

    def scan(self, feats: torch.Tensor, labels: torch.Tensor, weights: torch.Tensor) -> torch.Tensor:
        """(n, 2) int32 device tensor of the (lat_start, lon_start) that pass, in the reference's order."""
        P, S = int(self.config.PATCH_SIZE), int(self.config.STRIDE)
        nvar, lat, lon = feats.shape
        margin = P // 8
        rows = range(margin, lat - P - margin + 1, S)
        cols = range(margin, lon - P - margin + 1, S)
        if len(rows) == 0 or len(cols) == 0:
            return torch.zeros((0, 2), dtype=torch.int32, device=feats.device)
        flags = torch.empty(len(rows) * len(cols), dtype=torch.int32, device=feats.device)
        call("unetk_patch_qc", ptr(feats), nvar, ptr(labels), ptr(weights), lat, lon, P, S,
             float(self.config.MIN_VALID_RATIO), ptr(flags), stream())
        idx = torch.nonzero(flags, as_tuple=False).view(-1)
        pos = torch.stack([margin + (idx // len(cols)) * S, margin + (idx % len(cols)) * S], 1)
        return pos.to(torch.int32).contiguous()

    def gather(self, feats, labels, weights, pos: torch.Tensor, sanitize: bool = True,
               flips: Optional[torch.Tensor] = None):
        """Windows at ``pos`` as (n,4,P,P), (n,1,P,P), (n,1,P,P) float32 device tensors."""
        P = int(self.config.PATCH_SIZE)
        nvar, lat, lon = feats.shape
        n = pos.shape[0]
        d = feats.device
        out_f = torch.empty((n, nvar, P, P), dtype=torch.float32, device=d)
        out_l = torch.empty((n, 1, P, P), dtype=torch.float32, device=d)
        out_w = torch.empty((n, 1, P, P), dtype=torch.float32, device=d)
        if flips is not None:
            flips = flips.to(device=d, dtype=torch.int32).contiguous()
        call("unetk_patch_gather", ptr(feats), nvar, ptr(labels), ptr(weights), lat, lon, ptr(pos), ptr(flips), n, P,
             1 if sanitize else 0, ptr(out_f), ptr(out_l), ptr(out_w), stream())
        return out_f, out_l, out_w

    def device_batches(self, features_ds, labels_da, tid_weights_da, batch: int,
                       flips: Optional[torch.Tensor] = None) -> Iterator[Tuple[torch.Tensor, ...]]:
        """Batches of cleaned (features, label, tid_weights) straight from the grids on the device."""
        fields = self.to_device(features_ds, labels_da, tid_weights_da)
        if fields is None:
            return
        pos = self.scan(*fields)
        for s in range(0, pos.shape[0], batch):
            yield self.gather(*fields, pos[s:s + batch], True, None if flips is None else flips[s:s + batch])

    # ------------------------------------------------------------------ reference API
    def extract_patches(self, features_ds, labels_da, tid_weights_da) -> List[Dict]:
        fields = self.to_device(features_ds, labels_da, tid_weights_da)
        if fields is None:
            return []
        pos = self.scan(*fields)
        patches: List[Dict] = []
        chunk = max(1, (256 << 20) // (6 * 4 * int(self.config.PATCH_SIZE) ** 2))     # ~256 MiB per copy
        for s in range(0, pos.shape[0], chunk):
            p = pos[s:s + chunk]
            f, l, w = (t.cpu().numpy() for t in self.gather(*fields, p, sanitize=False))
            ph = p.cpu().numpy()
            for k in range(ph.shape[0]):
                patches.append({"features": f[k], "label": l[k, 0], "tid_weights": w[k, 0],
                                "lat_start": int(ph[k, 0]), "lon_start": int(ph[k, 1])})
        return patches
