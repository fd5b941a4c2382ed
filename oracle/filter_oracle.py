"""CPU restatement of the reference's Gaussian low-pass (TEST INFRASTRUCTURE ONLY: imported by
tests/ and the bench's CPU baseline, never by the product path).

Follows data_preparation.py:392-414 (apply_gaussian_filter) and the published algorithm of its
one dependency, scipy.ndimage.gaussian_filter (scipy 1.x: _gaussian_kernel1d + correlate1d per
axis, mode='reflect' = half-sample symmetric, truncate = 4.0).  Pinned by
tests/test_oracle_golden.py against scipy itself (present in this image) and against the
committed fixture tests/golden/filter.npz.
"""
import numpy as np


def gaussian_kernel1d(sigma, truncate=4.0):
    radius = int(truncate * float(sigma) + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return phi / phi.sum()


def reflect_index(i, n):
    """... c b a | a b c d | d c b ..."""
    i = np.asarray(i).copy()
    while True:
        lo, hi = i < 0, i >= n
        if not (lo.any() or hi.any()):
            return i
        i[lo] = -i[lo] - 1
        i[hi] = 2 * n - 1 - i[hi]


def correlate1d_reflect(x, w, axis):
    r = (len(w) - 1) // 2
    n = x.shape[axis]
    idx = reflect_index(np.arange(-r, n + r), n)
    xp = np.take(x, idx, axis=axis)
    out = np.zeros_like(x)
    for t in range(2 * r + 1):
        out += w[t] * np.take(xp, np.arange(t, t + n), axis=axis)
    return out


def gaussian_filter(x, sigma_lat, sigma_lon):
    x = np.asarray(x, dtype=float)
    y = correlate1d_reflect(x, gaussian_kernel1d(sigma_lat)[::-1], 0)
    return correlate1d_reflect(y, gaussian_kernel1d(sigma_lon)[::-1], 1)


def apply_gaussian_filter(data, sigma_lat, sigma_lon):
    """data_preparation.py:394-404"""
    v = np.asarray(data).copy().astype(float)
    nan_mask = np.isnan(v)
    v[nan_mask] = 0
    f = gaussian_filter(v, sigma_lat, sigma_lon)
    f[nan_mask] = np.nan
    return f


def synthetic_field(lat, lon, nan_frac, seed):
    """Seeded test grid: depths in metres with a share of NaN cells."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((lat, lon)) * 300.0 - 4000.0
    x[rng.random((lat, lon)) < nan_frac] = np.nan
    return x
