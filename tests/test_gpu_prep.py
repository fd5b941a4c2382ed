"""GPU parity: the Gaussian low-pass of the data preparation (data_preparation.py:392-414) against the CPU
oracle and the fixture frozen from the reference.  float64: agreement to rounding (1e-12 of the field's
scale; only the summation order differs), NaN cells identical."""
import numpy as np
import pytest
import torch

from oracle import filter_oracle
from tests.test_oracle_golden import FILTER_CASES

pytestmark = pytest.mark.gpu
TOL = 1e-12


@pytest.mark.parametrize("case", FILTER_CASES, ids=[c[0] for c in FILTER_CASES])
def test_filter_matches_golden(golden, case):
    from unet_amd import prep
    name, lat, lon, s_lat, s_lon, nan_frac, seed = case
    x = filter_oracle.synthetic_field(lat, lon, nan_frac, seed)
    got = prep.apply_gaussian_filter(x, s_lat, s_lon)
    gold = golden("filter")[name]
    assert got.dtype == np.float64 and got.shape == x.shape
    assert np.array_equal(np.isnan(got), np.isnan(gold))
    assert np.nanmax(np.abs(got - gold)) <= TOL * np.nanmax(np.abs(gold))


@pytest.mark.parametrize("shape,sig", [((300, 517), (7.0, 11.5)), ((16, 16), (0.6, 0.6)), ((1, 64), (3.0, 3.0)),
                                        ((530, 40), (64.0, 1.5))])
def test_filter_matches_oracle(shape, sig):
    """Ragged sizes (tile edges in both passes), a one-row grid, and the largest radius the kernel holds."""
    from unet_amd import prep
    x = filter_oracle.synthetic_field(shape[0], shape[1], 0.02, 11)
    got = prep.apply_gaussian_filter(x, *sig)
    want = filter_oracle.apply_gaussian_filter(x, *sig)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.nanmax(np.abs(got - want)) <= TOL * np.nanmax(np.abs(want))


def test_filter_properties_at_full_size():
    """8192 x 8192 with the 135 km sigma of a 1-arc-minute grid: a constant field stays constant (weights sum
    to 1 at every border), the filter is linear, and NaNs come back exactly where they were."""
    from unet_amd import prep
    n, sig = 8192, 31.0
    dev = torch.device("cuda")
    c = torch.full((n, n), -4000.0, dtype=torch.float64, device=dev)
    out = prep.gaussian_filter_device(c, sig, sig)
    assert float((out + 4000.0).abs().max()) < 1e-9
    g = torch.Generator(device=dev).manual_seed(5)
    a = torch.randn((n, n), dtype=torch.float64, device=dev, generator=g)
    b = torch.randn((n, n), dtype=torch.float64, device=dev, generator=g)
    fa, fb = prep.gaussian_filter_device(a, sig, sig, nan_aware=False), prep.gaussian_filter_device(b, sig, sig, nan_aware=False)
    fab = prep.gaussian_filter_device(2.0 * a - 3.0 * b, sig, sig, nan_aware=False)
    assert float((fab - (2.0 * fa - 3.0 * fb)).abs().max()) < 1e-12
    a[100:200, 4000:4100] = float("nan")
    fn = prep.gaussian_filter_device(a, sig, sig)
    assert torch.equal(torch.isnan(fn), torch.isnan(a))


def test_sigma_pixels_and_errors():
    from unet_amd import prep
    lat = np.linspace(10.0, 20.0, 601)
    lon = np.linspace(140.0, 150.0, 601)
    s_lat, s_lon = prep.calculate_sigma_pixels(135.0, lat, lon)
    assert abs(s_lat - 135.0 / 111.0 / (10.0 / 600) / 2.355) < 1e-9
    assert abs(s_lon - 135.0 / (111.0 * np.cos(np.radians(15.0))) / (10.0 / 600) / 2.355) < 1e-9
    with pytest.raises(ValueError):
        prep.apply_gaussian_filter(np.zeros((8, 8)), 100.0, 1.0)      # radius 400 > the kernel's 256
    with pytest.raises(RuntimeError):
        prep.gaussian_filter_device(torch.zeros((4, 4), dtype=torch.float64), 1.0, 1.0)
