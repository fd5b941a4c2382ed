"""GPU tests: training-patch extraction against the CPU oracle (pinned by tests/golden/patch.npz)."""
import tempfile

import numpy as np
import pytest
import torch

from oracle import patch_oracle
from tests.test_oracle_golden import PATCH_CASES, _DA, _DSet

pytestmark = pytest.mark.gpu


def _cfg(P, S, mv):
    return type("C", (), {"OUTPUT_DIR": tempfile.mkdtemp(), "PATCH_SIZE": P, "STRIDE": S, "MIN_VALID_RATIO": mv})


@pytest.mark.parametrize("case", PATCH_CASES, ids=[c[0] for c in PATCH_CASES])
def test_extract_patches_bit_exact(golden, case):
    from unet_amd.data import EnhancedDataPreparation
    name, lat, lon, P, S, mv, seed = case
    feats, labels, weights = patch_oracle.synthetic_fields(lat, lon, seed)
    got = EnhancedDataPreparation(_cfg(P, S, mv)).extract_patches(_DSet(feats), _DA(labels), _DA(weights))
    want = patch_oracle.extract_patches(feats, labels, weights, P, S, mv)
    g = golden("patch")
    assert np.array_equal(np.array([(p["lat_start"], p["lon_start"]) for p in got], np.int32), g[name + ".pos"])
    assert len(got) == len(want)
    for a, b in zip(got, want):
        for key in ("features", "label", "tid_weights"):
            assert a[key].dtype == np.float32 and np.array_equal(a[key], b[key], equal_nan=True)


def test_device_batches_clean_and_flip():
    from unet_amd.data import EnhancedDataPreparation
    name, lat, lon, P, S, mv, seed = PATCH_CASES[1]
    feats, labels, weights = patch_oracle.synthetic_fields(lat, lon, seed)
    want = patch_oracle.extract_patches(feats, labels, weights, P, S, mv)
    prep = EnhancedDataPreparation(_cfg(P, S, mv))
    flips = torch.tensor([k % 3 for k in range(len(want))])
    k = 0
    for f, l, w in prep.device_batches(_DSet(feats), _DA(labels), _DA(weights), batch=7, flips=flips):
        for b in range(f.shape[0]):
            fo, lo, wo = patch_oracle.dataset_item(want[k], {0: None, 1: "h", 2: "v"}[k % 3])
            assert np.array_equal(f[b].cpu().numpy(), fo) and np.array_equal(l[b].cpu().numpy(), lo)
            assert np.array_equal(w[b].cpu().numpy(), wo)
            k += 1
    assert k == len(want)


def test_edge_cases():
    from unet_amd.data import EnhancedDataPreparation
    feats, labels, weights = patch_oracle.synthetic_fields(40, 40, 9)
    prep = EnhancedDataPreparation(_cfg(64, 16, 0.25))               # grid smaller than a patch + margins
    assert prep.extract_patches(_DSet(feats), _DA(labels), _DA(weights)) == []
    missing = {k: v for k, v in feats.items() if k != "VGG"}          # a feature variable is absent
    assert EnhancedDataPreparation(_cfg(8, 4, 0.25)).extract_patches(_DSet(missing), _DA(labels), _DA(weights)) == []
    nan = np.full_like(labels, np.nan)                                # nothing valid anywhere
    assert EnhancedDataPreparation(_cfg(8, 4, 0.25)).extract_patches(_DSet(feats), _DA(nan), _DA(weights)) == []
